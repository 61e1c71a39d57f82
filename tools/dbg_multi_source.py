"""ON THE GPU BOX: engine vs oracle for one multi_source_spec case, per LP, on the lazy and on the forced-prologue path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import helpers as H, random_specs as RS
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22522
spec = RS.multi_source_spec(k); spec["trace"] = False
runs = H.run_oracle_for_spec(spec)
for flags in (0, 1 << 16, 1):
    eng, p = H.engine_for_spec(spec)
    if flags: eng.set_debug_flags(flags)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary(); st = eng.lp_stats()
        print("flags", flags, "prologue path", eng.prologue_path(), "events", s.events_processed, "final", s.final_time_ns)
        for key in ("generated", "accepted", "dropped", "completed", "rejected", "sink_received", "queue_depth", "active", "events"):
            print("  eng", key, list(st[key]))
        print("  by kind", list(s.events_by_kind))
for chain_ids, nodes, r in runs:
    print("oracle events", r.events_processed, "final", r.final_time_ns, "by kind", list(r.events_by_kind))
    srv = [nodes[c][1] for c in chain_ids]          # nodes[c] = (source, server, sink)
    for key, arr in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed), ("rejected", r.rejected), ("depth", r.depth), ("active", r.active)):
        print("  orc", key, list(arr[srv]))
