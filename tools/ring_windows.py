#!/usr/bin/env python3
"""Windows over the 65 536-station ring: W windows of end_s / W against ONE run to end_s -- wall time of each, both
ways of driving windows (continue from the last state: round 6; repeat from the start: debug flag 1 << 24, rounds 4-5), and that the
final states agree.  Measurement tool."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--end-s", type=float, default=1.0)
ap.add_argument("--windows", type=int, default=1000)
ap.add_argument("--repeat-windows", type=int, default=100, help="windows of the by-repetition run (it is quadratic)")
ap.add_argument("--engine-flags", type=int, default=0)
a = ap.parse_args()
spec = dict(name="ring_windows", topology="ring", n=a.n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=a.end_s, seed=42)


def state(eng):
    s = eng.summary()
    return (s.events_processed, tuple(int(x) for x in s.events_by_kind), s.final_time_ns, {k: v.tobytes() for k, v in eng.lp_stats().items()},
            {k: v.tobytes() for k, v in eng.net_stats().items()}, [x.tobytes() for x in eng.read_sinks()])


out = {}
eng, p = H.ring_engine_for_spec(spec, flags=a.engine_flags)
with eng:
    eng.run_until(p["end_ns"])                    # (warm: module load, first launch)
    eng.reset()
    t0 = time.perf_counter(); eng.run_until(p["end_ns"]); out["one_run_ms"] = (time.perf_counter() - t0) * 1e3
    out["events"] = eng.summary().events_processed
    want = state(eng)
for label, extra, W in (("continued", 0, a.windows), ("repeated", 1 << 24, a.repeat_windows)):
    eng, p = H.ring_engine_for_spec(spec, flags=a.engine_flags | extra)
    with eng:
        eng.run_until(p["end_ns"] // W)
        eng.reset()
        paths = {}
        t0 = time.perf_counter()
        for k in range(W):
            eng.run_until(p["end_ns"] * (k + 1) // W)
            paths[eng.window_path()] = paths.get(eng.window_path(), 0) + 1
        out[label] = dict(windows=W, total_ms=(time.perf_counter() - t0) * 1e3, paths=paths, equal_to_one_run=state(eng) == want)
        out[label]["ms_per_window"] = out[label]["total_ms"] / W
print(json.dumps(out))
