#!/usr/bin/env python3
"""Debug (scratch): one multi-source ring case driven by the sweep's random window ends, under several flags, against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import random_specs as RS
from oracle import hs_oracle as O

k = int(sys.argv[1]); windows = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = RS.multi_source_ring_spec(k)
g, nodes = H.oracle_ring_graph(spec)
p = H.ring_params(spec)
r = O.run(g, p["end_ns"], seed=spec["seed"], schedule=[(nodes[c]["srv"], t) for c, t in p["schedule"]])
rng = np.random.default_rng(spec["seed"] + 977)
ends = [int(e) for e in np.unique(rng.integers(1, p["end_ns"], windows))]
print("ends", ends, "end", p["end_ns"])
for flags in (0, 1 << 24, 16, 16 | (1 << 24), 1 << 16):
    for use_windows in (True, False):
        eng, _ = H.ring_engine_for_spec(spec, flags=flags)
        with eng:
            paths = []
            if use_windows:
                for e in ends:
                    eng.run_until(e); paths.append((eng.window_path(), eng.prologue_path()))
            eng.run_until(p["end_ns"]); paths.append((eng.window_path(), eng.prologue_path()))
            bad = []
            for i in range(spec["n"]):
                if "prb" in nodes[i]:
                    t, v = r.sinks[nodes[i]["prb"]]
                    pt, pv = eng.read_probe(i)
                    if not (np.array_equal(pt, t) and np.array_equal(pv, v)):
                        bad.append((i, pt.tolist(), pv.tolist(), np.asarray(t).tolist(), np.asarray(v).tolist()))
            s = eng.summary()
            print("flags", flags, "windows" if use_windows else "one run", "paths", paths, "events", s.events_processed, r.events_processed, "probe diffs", bad)
