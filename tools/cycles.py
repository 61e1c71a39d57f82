#!/usr/bin/env python3
"""Where the headline kernel's request-order loop spends its cycles (scratch tool).

Builds an instrumented copy of the library (-DHS_CYCLES: s_memtime around the wave-level refill and around the serial
request step) into scratch/, runs the grid workload once and prints cycles per iteration.  The shipped library never
contains the instrumentation.

    python tools/cycles.py --build        # here (cross-compile)
    gpurun -- 'HS_HIP_LIB=scratch/libhs_hip_cycles.so python tools/cycles.py'
"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scratch", "libhs_hip_cycles.so")

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--n-lp", type=int, default=65536)
ap.add_argument("--end-s", type=float, default=60.0)
ap.add_argument("--define", action="append", default=[])
ap.add_argument("--ring", action="store_true", help="the asynchronous network engine on the 65 536-station ring")
a = ap.parse_args()
from happy_simulator_amd import _native as N
if a.build:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    N.build(defines=("HS_CYCLES", *a.define), lib_path=OUT)
    print("built", OUT)
    sys.exit(0)
if a.ring:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    spec = dict(name="ring_full", topology="ring", n=a.n_lp, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01,
                end_s=a.end_s, seed=42)
    eng, p = H.ring_engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary()
        out = (C.c_ulonglong * 4)()
        N.lib().hs_debug_async_counters(eng._h, out)
        waves = (a.n_lp + 63) // 64
        print(json.dumps(dict(events=s.events_processed, kernel_ms=float(s.kernel_ms), waves=waves,
                              cycles_per_wave=dict(refills=out[0] / waves, receive_and_bound_scan=out[1] / waves,
                                                   groups=out[2] / waves, publish=out[3] / waves))))
    sys.exit(0)
from happy_simulator_amd.engine import StationArrays, StationEngine
end_ns = int(a.end_s * 1e9)
eng = StationEngine(StationArrays.uniform(a.n_lp), mode=N.MODE_SINGLE, horizon_ns=end_ns, seed=42)
with eng:
    eng.run_until(end_ns)
    s = eng.summary()
    out = (C.c_ulonglong * 4)()
    N.lib().hs_debug_async_counters(eng._h, out)
    k, tot = eng.bench_runs(end_ns, 5)
    it = out[2] / max(out[3], 1)
    print(json.dumps(dict(events=s.events_processed, kernel_ms=float(k.mean()), waves=out[3], iterations_per_wave=it,
                          refill_cycles_per_iteration=out[0] / max(out[2], 1),
                          step_cycles_per_iteration=out[1] / max(out[2], 1),
                          loop_cycles_per_wave=(out[0] + out[1]) / max(out[3], 1))))
