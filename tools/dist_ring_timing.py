#!/usr/bin/env python3
"""Where a sharded ring run spends its wall time when the ranks are PROCESSES (torch.distributed): per-phase host timings of the
asynchronous rounds with the device-side exchange against the collective exchange.  On a one-GPU box:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29577 tools/dist_ring_timing.py --same-device

(scratch tool: `bench.py --workload ring --gpus N` / `--fake-ranks N` is the measurement of record)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--end-s", type=float, default=60.0)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--sync-every", type=int, default=4)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import helpers as H
    from happy_simulator_amd.sharded import DistComm, ShardedNetwork

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    spec = dict(name="ring_full", topology="ring", n=a.n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=a.end_s, seed=42)
    st, net, cap, p = H.ring_arrays(spec)
    for exchange in ("device", "collective"):
        sn = ShardedNetwork.on_gpu(st, net, DistComm(), horizon_ns=p["end_ns"], seed=42, device=local, log_capacity=cap,
                                   sync_every=a.sync_every, rounds=True, exchange=exchange)
        with sn:
            sn.run_until(p["end_ns"])
            # instrument the calls of the (single) local shard
            sh = sn.shards[0]
            acc = {}

            def timed(name, fn, sync=False):
                def w(*x, **k):
                    t0 = time.perf_counter()
                    r = fn(*x, **k)
                    if sync:
                        torch.cuda.synchronize()
                    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
                    return r
                return w
            for nm in ("round", "push", "inject_ipc", "inject_async", "round_done", "final", "begin"):
                setattr(sh, nm, timed(nm, getattr(sh, nm)))
            comm = sn.comm
            for nm in ("flag_barrier", "exchange", "allreduce_max", "allgather_rows", "reduce_host"):
                setattr(comm, nm, timed("comm." + nm, getattr(comm, nm)))
            dist.barrier()
            t0 = time.perf_counter()
            s = sn.run_until(p["end_ns"])
            wall = time.perf_counter() - t0
            if rank == 0:
                print(json.dumps(dict(exchange=exchange, world=world, exchanges=s.windows, wall_ms=round(wall * 1e3, 2),
                                      exchange_ms=round(s.exchange_seconds * 1e3, 2),
                                      host_ms={k: round(v * 1e3, 2) for k, v in sorted(acc.items())})), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
