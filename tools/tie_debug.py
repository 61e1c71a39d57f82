#!/usr/bin/env python3
"""Debug (scratch): one tie-storm case on the station engine under several debug flags against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import random_specs as RS
import test_gpu_random as TR

k = int(sys.argv[1])
for flags in (0, 1, 1 << 16, 1 << 17, 1 | (1 << 16)):
    spec = RS.tie_spec(k)
    spec["trace"] = False
    runs = H.run_oracle_for_spec(spec)
    eng, p = H.engine_for_spec(spec, flags=flags)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary()
        st = eng.lp_stats()
        r = runs[0][2]
        print("flags", flags, "prologue_path", eng.prologue_path(), "events", s.events_processed, "oracle", r.events_processed, "final", s.final_time_ns, r.final_time_ns)
        print("  kinds eng ", list(map(int, s.events_by_kind)))
        print("  kinds orcl", list(map(int, r.events_by_kind)))
        nodes = runs[0][1]
        for key in ("generated", "accepted", "dropped", "completed", "queue_depth", "active"):
            print("  ", key, st[key].tolist())
        try:
            TR._compare_engine_to_oracle(spec, eng, p, runs, check_kinds=True)
            print("  == oracle")
        except AssertionError as e:
            print("  DIFF", str(e).replace("\n", " | ")[:300])
