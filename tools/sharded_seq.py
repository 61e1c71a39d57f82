#!/usr/bin/env python3
"""Run a sequence of sharded-vs-single-engine comparisons in ONE process and say which differ (tests/test_gpu_sharded.py runs it in a
fresh process: what an engine reads must not depend on the engines the process created and destroyed before it).
usage: sharded_seq.py "spec:world:proto,spec:world:proto,..."   proto = d (device rounds) | c (collective rounds) | w (windows)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sharded as T

specs = {s["name"]: s for s in T.SPECS}
proto = {"d": True, "c": "collective", "w": False}
res = []
for item in sys.argv[1].split(","):
    name, world, pr = item.split(":")
    try:
        one = T._single(specs[name])
        summ, stats, netst, sinks = T._sharded(specs[name], int(world), rounds=proto[pr])
        bad = []
        if summ.events_processed != one["events"]: bad.append(f"events {summ.events_processed} vs {one['events']}")
        for k in stats:
            if k in one["stats"] and not np.array_equal(stats[k], one["stats"][k]):
                bad.append(f"{k}: {np.flatnonzero(stats[k] != one['stats'][k])[:5].tolist()}")
        for k in netst:
            if not np.array_equal(netst[k], one["net"][k]):
                ix = np.flatnonzero(netst[k] != one["net"][k])
                bad.append(f"net {k}: n={len(ix)} first={ix[:3].tolist()} last={ix[-3:].tolist()} got={netst[k][ix[:3]].tolist()} want={one['net'][k][ix[:3]].tolist()}")
        res.append((item, "ok" if not bad else "; ".join(bad)[:400]))
    except Exception as e:  # noqa
        res.append((item, f"EXC {type(e).__name__}: {str(e)[:200]}"))
for r in res:
    if r[1] != "ok":
        print(r)
print("n =", len(res), "bad =", sum(1 for r in res if r[1] != "ok"))
