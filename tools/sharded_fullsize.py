#!/usr/bin/env python3
"""Config 3 on ONE MI355X with virtual shards (LocalComm): the 65 536-station ring cut into N segments that take turns on
the device -- exchange counts and wall time of the two shard protocols (scratch tool; real multi-GPU runs: bench.py
--workload ring --gpus N)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from happy_simulator_amd.sharded import LocalComm, ShardedNetwork

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--end-s", type=float, default=60.0)
ap.add_argument("--shards", type=int, default=4)
ap.add_argument("--round-iters", type=int, nargs="+", default=[16, 32, 64])
ap.add_argument("--windows", action="store_true")
ap.add_argument("--exchange", default="device", choices=("device", "collective"))
a = ap.parse_args()
spec = dict(name="ring_full", topology="ring", n=a.n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=a.end_s, seed=42)
st, net, cap, p = H.ring_arrays(spec)
for it in ([0] if a.windows else a.round_iters):
    sn = ShardedNetwork.on_gpu(st, net, LocalComm(a.shards), horizon_ns=p["end_ns"], seed=42, log_capacity=cap,
                               sync_every=64 if a.windows else 4, rounds=not a.windows, round_iters=max(it, 1), exchange=a.exchange)
    with sn:
        sn.run_until(p["end_ns"])                       # warm-up
        t0 = time.perf_counter(); s = sn.run_until(p["end_ns"]); wall = time.perf_counter() - t0
        print(json.dumps(dict(protocol="windows" if a.windows else ("live" if sn.live else "rounds"), round_iters=it, shards=a.shards,
                              exchanges=s.windows, events=s.events_processed, wall_s=wall)))
