"""Debug helper (GPU): asynchronous vs windowed network engine on one ring golden."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import helpers as H

name = sys.argv[1]
gold = H.Golden(name)
res = {}
for flags in (0, 16):
    eng, p = H.ring_engine_for_spec(gold.spec, flags=flags)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary(); st = eng.lp_stats(); ns = eng.net_stats()
        c, t, cr = eng.read_sinks()
    res[flags] = (s, st, ns, c, t, cr)
    print("flags", flags, "events", s.events_processed, "final", s.final_time_ns, "launches", s.launches)
    print("  kinds", s.events_by_kind)
a, b = res[0], res[16]
for k in a[1]:
    if not np.array_equal(a[1][k], b[1][k]):
        print("lp_stats differ", k, a[1][k], b[1][k])
for k in a[2]:
    if not np.array_equal(a[2][k], b[2][k]):
        print("net_stats differ", k, a[2][k], b[2][k])
print("sink counts", a[3], b[3])
print("gold total", gold.meta["total_events"], gold.meta["final_ns"])
