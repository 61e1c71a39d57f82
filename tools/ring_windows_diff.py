#!/usr/bin/env python3
"""Debug (scratch): first window end at which a windowed drive differs from one run, and the stations around the difference."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--first-s", type=float, default=1.0)
ap.add_argument("--step-s", type=float, default=0.05)
ap.add_argument("--end-s", type=float, default=2.0)
ap.add_argument("--engine-flags", type=int, default=0)
a = ap.parse_args()
spec = dict(name="ring_windows", topology="ring", n=a.n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=a.end_s, seed=42)


def state(eng):
    s = eng.summary()
    d = dict(events=np.array([s.events_processed]), final=np.array([s.final_time_ns]), kinds=np.asarray(s.events_by_kind))
    d.update({"lp_" + k: v for k, v in eng.lp_stats().items()})
    d.update({"net_" + k: v for k, v in eng.net_stats().items()})
    return d


eng, p = H.ring_engine_for_spec(spec, flags=a.engine_flags)
ref, _ = H.ring_engine_for_spec(spec, flags=a.engine_flags)
ends = [H.ns_from_seconds(a.first_s)]
while ends[-1] < p["end_ns"]:
    ends.append(min(p["end_ns"], ends[-1] + H.ns_from_seconds(a.step_s)))
with eng, ref:
    for k, e in enumerate(ends):
        eng.run_until(e)
        ref.reset(); ref.run_until(e)
        x, y = state(eng), state(ref)
        bad = sorted({int(i) for key in x if key.startswith(("lp_", "net_")) for i in np.flatnonzero(x[key] != y[key])})
        if bad or x["events"][0] != y["events"][0]:
            out = dict(flags=a.engine_flags, window=k, end_ns=e, path=eng.window_path(), stations=bad[:12], events=[int(x["events"][0]), int(y["events"][0])],
                       final=[int(x["final"][0]), int(y["final"][0])], kinds_minus_ref=[int(v) for v in (x["kinds"].astype(np.int64) - y["kinds"].astype(np.int64))])
            for i in bad[:4]:
                out[f"lp{i}"] = {key: [str(x[key][i]), str(y[key][i])] for key in x if key.startswith(("lp_", "net_")) and x[key][i] != y[key][i]}
            print(json.dumps(out))
            break
    else:
        print(json.dumps(dict(flags=a.engine_flags, windows=len(ends), all_equal=True)))
