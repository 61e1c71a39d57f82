"""One rank of a sharded ring run over torch.distributed (launched by tests/test_gpu_dist.py, one process per GPU).
Prints one JSON line on rank 0: the sharded summary and whether it equals the single-engine run of the same network."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    import torch.distributed as dist

    import helpers as H
    from happy_simulator_amd.sharded import DistComm, ShardedNetwork

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    spec = dict(name="dist_ring", topology="ring", n=int(sys.argv[1]), ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01,
                end_s=float(sys.argv[2]), seed=11)
    out = {}
    for rounds in (True, False):
        st, net, cap, p = H.ring_arrays(spec)
        with ShardedNetwork.on_gpu(st, net, DistComm(), horizon_ns=p["end_ns"], seed=spec["seed"], device=local,
                                   log_capacity=cap, rounds=rounds) as sn:
            s = sn.run_until(p["end_ns"])
            stats, counts, t, cr, ns = sn.collect(spec["n"], spec["n"])
            lo, hi = sn.shards[0].lo, sn.shards[0].hi
            out["rounds" if rounds else "windows"] = dict(
                events=int(s.events_processed), final=int(s.final_time_ns), exchanges=int(s.windows), world=int(s.world),
                completed_local=int(stats["completed"][lo:hi].sum()), sinks_local=int(counts[lo:hi].sum()))
    if rank == 0:
        eng, p = H.ring_engine_for_spec(spec)            # the same network on one engine
        with eng:
            eng.run_until(p["end_ns"])
            s1 = eng.summary()
            out["single"] = dict(events=int(s1.events_processed), final=int(s1.final_time_ns),
                                 completed_local=int(eng.lp_stats()["completed"][lo:hi].sum()))
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
