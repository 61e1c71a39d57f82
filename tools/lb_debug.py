"""Debug helper (GPU): engine vs oracle per-backend counters for one load-balancer spec of tests/test_gpu_lb.py."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import helpers as H
from oracle import hs_oracle as O
import test_gpu_lb as T

name = sys.argv[1]
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spec = [s for s in T.TIES + T.SWEEP if s["name"] == name][0]
g, p = H.oracle_lb_graph_ext(spec)
S, B = p["S"], p["B"]
tr_cap = 200000
r = O.run(g, p["end_ns"], seed=spec["seed"], trace_cap=tr_cap)
eng, _ = H.lb_engine_for_spec(spec, flags=flags)
with eng:
    eng.run(p["end_ns"])
    s = eng.summary(); st = eng.stats()
print("engine kinds", s.events_by_kind)
print("oracle kinds", r.events_by_kind)
be = slice(S + 1, S + 1 + B)
for k, ok in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed), ("queue_depth", r.depth), ("active", r.active)):
    print(k, st[k], ok[be])
# per-backend notify counts from the oracle trace
t, k, nd, ix = r.trace
for b in range(B):
    m = (nd == S + 1 + b)
    print("backend", b, "oracle kinds", np.bincount(k[m], minlength=13))
# the oracle's notifies: print context around each notify
idx = np.where(k == 2)[0]
for i in idx:
    lo = max(0, i - 8)
    print("--- notify at", t[i], "node", nd[i])
    for j in range(lo, min(len(t), i + 6)):
        print("   ", t[j], O.EV_NAMES[k[j]], nd[j], ix[j])
