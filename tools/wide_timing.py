#!/usr/bin/env python3
"""Step time of the K-lanes-per-LP kernel (csrc/hs_kernels_wide.hpp) against one lane per LP, on the GPU box:

    gpurun -- 'python tools/wide_timing.py'                      # kernel ms (median of 10 reset + run steps), sizes x K
    gpurun -- 'HS_HIP_LIB=happy_simulator_amd/lib/instr/libhs_widecyc.so python tools/wide_timing.py --cycles'

--cycles needs a library built with -DHS_WIDE_CYC (python -c "from happy_simulator_amd import _native as N;
N.build(defines=('HS_WIDE_CYC',), lib_path='happy_simulator_amd/lib/instr/libhs_widecyc.so')"): s_memtime cycles each of the three
role wavefronts of workgroup 0 spends working, next to the loop's total (what is left is barrier wait).
Debug flags: bits 24..27 force K = 1 << (value - 1); 1 << 22 keeps hs_station_run; 1 << 19 / 1 << 18 switch the log appends /
the service-time sum off (timing experiments: results are then wrong)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happy_simulator_amd import _native as N  # noqa: E402
from happy_simulator_amd.engine import StationArrays, StationEngine  # noqa: E402

END = 60_000_000_000
FORCE = {4: 3 << 24, 8: 4 << 24, 16: 5 << 24, 64: 7 << 24, 65: 8 << 24}   # 64 / 65: a wavefront per LP, 16 / 8 LPs per workgroup


def step_ms(n, flags, reps=10):
    st = StationArrays.uniform(n, rate=8.0, mean=0.1)
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=END, seed=42) as eng:
        eng.set_debug_flags(flags)
        eng.bench_runs(END, 3)
        k, _ = eng.bench_runs(END, reps)
        return float(np.median(k))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,4096,8192,16384,32768,65536")
    ap.add_argument("--cycles", action="store_true")
    ap.add_argument("--wave-span", action="store_true", help="library built with -DHS_WAVE_CYC -DHS_WAVE_SPAN: first / last start and end of the workgroups")
    ap.add_argument("--wave-cycles", action="store_true", help="library built with -DHS_WAVE_CYC: cycles of wavefront 1 of workgroup 0 of hs_station_wave")
    a = ap.parse_args()
    for n in (int(x) for x in a.sizes.split(",")):
        if a.wave_span:
            for K in (64, 65):
                st = StationArrays.uniform(n, rate=8.0, mean=0.1)
                with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=END, seed=42) as eng:
                    eng.set_debug_flags(FORCE[K])
                    eng.run_until(END)
                    out = (C.c_ulonglong * 4)()
                    eng._lib.hs_debug_async_counters(eng._h, out)
                    m = (1 << 64) - 1
                    s0, s1, e0, e1 = m - out[0], out[1], m - out[2], out[3]
                    print(f"n_lp {n} K {K}: workgroup starts span {s1 - s0} ticks, first end {e0 - s0}, last end {e1 - s0} (ticks after the first start)", flush=True)
            continue
        if a.wave_cycles:
            for K in (64, 65):
                st = StationArrays.uniform(n, rate=8.0, mean=0.1)
                with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=END, seed=42) as eng:
                    eng.set_debug_flags(FORCE[K])
                    eng.run_until(END)
                    out = (C.c_ulonglong * 4)()
                    eng._lib.hs_debug_async_counters(eng._h, out)
                    print(f"n_lp {n} K {K}: cycles of wavefront 1 of workgroup 0: compute {out[0]}, barrier wait {out[1]}, T + writes {out[2]}, "
                          f"before the loop {out[3] // 1000000}, after it {out[3] % 1000000}", flush=True)
            continue
        if a.cycles:
            for K in (4, 8):
                st = StationArrays.uniform(n, rate=8.0, mean=0.1)
                with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=END, seed=42) as eng:
                    eng.set_debug_flags(FORCE[K])
                    eng.run_until(END)
                    out = (C.c_ulonglong * 4)()
                    eng._lib.hs_debug_async_counters(eng._h, out)
                    print(f"n_lp {n} K {K}: work cycles of workgroup 0 [values, chain + sum, Lindley]: {list(out)[:3]}, loop total {out[3]}")
            continue
        row = {"one lane per LP": round(step_ms(n, 1 << 22), 4), "automatic": round(step_ms(n, 0), 4)}
        for K in (4, 8, 16, 64, 65):
            row[f"K={K}"] = round(step_ms(n, FORCE[K]), 4)
        row["K=64 no logs"] = round(step_ms(n, FORCE[64] | (1 << 19)), 4)
        row["K=64 no logs, no sum"] = round(step_ms(n, FORCE[64] | (1 << 19) | (1 << 18)), 4)
        print(f"n_lp {n}: kernel ms {row}", flush=True)


if __name__ == "__main__":
    main()
