#!/usr/bin/env python3
"""Config 3 at full size on one MI355X: 65 536-station ring, timing + accounting identities (scratch tool)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--end-s", type=float, default=60.0)
ap.add_argument("--lat-min", type=float, default=0.001)
ap.add_argument("--jitter", type=float, default=0.01)
ap.add_argument("--repeats", type=int, default=2)
ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
spec = dict(name="ring_full", topology="ring", n=a.n, ext_rate=4.0, mean=0.1, lat_min=a.lat_min, jitter_mean=a.jitter,
            end_s=a.end_s, seed=42)
eng, p = H.ring_engine_for_spec(spec, flags=a.flags)
with eng:
    t0 = time.perf_counter(); eng.run_until(p["end_ns"]); t1 = time.perf_counter()
    s = eng.summary()
    print(json.dumps(dict(first_run_wall_s=t1 - t0, events=s.events_processed, launches=s.launches, window_ns=s.window_ns,
                          kernel_ms=s.kernel_ms, by_kind=s.events_by_kind.tolist(), final=s.final_time_ns)))
    k, tot = eng.bench_runs(p["end_ns"], a.repeats)
    s = eng.summary()
    print(json.dumps(dict(bench_ms=[float(x) for x in k], events=s.events_processed, ev_per_s=float(s.events_processed / (k.mean() * 1e-3)),
                          us_per_window=float(k.mean()) * 1e3 / s.launches)))
    import ctypes as C
    from happy_simulator_amd import _native as N
    out = (C.c_ulonglong * 4)()
    if N.lib().hs_debug_async_counters(eng._h, out) == 0 and out[3]:
        print(json.dumps(dict(async_wave_iterations_avg=out[0] / out[3], async_wave_iterations_max=out[1],
                              groups_per_lp=out[2] / a.n, waves=out[3])))
