"""One rank of a sharded network run over torch.distributed (launched by tests/test_gpu_dist.py, one process per rank).

    dist_ring_worker.py CASE N END_S [--backend nccl|gloo] [--same-device]

`--backend nccl` (default): one rank per GPU, RCCL moves the device tensors.  `--backend gloo --same-device`: every rank uses
device 0 and DistComm stages the exchange tensors through host memory -- two REAL GpuShards in two processes on ONE GPU, the
slicing (`shard_arrays`), both exchange protocols, the cross-rank election and the reductions are the multi-GPU run's.

CASE: `ring` = the uniform ring of bench.py; `mixed` = a ring with several Sources per Server (listed extras-first), probes,
a time-varying profile and Requests injected with schedule() -- the paths `shard_arrays` filters and re-bases; `lockstep` = only
constant Sources, two of them in different shards and listed in reverse station order, so that the one event beyond end_time is a
tie on (time, creation time, lineage) that only the NETWORK-WIDE construction rank decides.

Prints one JSON line on rank 0: per protocol the sharded totals and whether every per-station statistic equals the single-engine
run of the same network."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

STAT_KEYS = ("generated", "accepted", "dropped", "completed", "sink_received", "queue_depth", "events")


def case_spec(case, n, end_s):
    if case == "ring":
        return dict(name="dist_ring", topology="ring", n=n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=end_s,
                    seed=11)
    if case == "mixed":
        prof = [None] * n
        for i in range(5, n, 31):
            prof[i] = ["ramp", end_s * (0.5 + 0.1 * (i % 5)), 3.0 + (i % 4), 6.0 + (i % 17)]
        more = [None if prof[i] is not None else ([["constant", 5.0]] if i % 11 == 3 else
                                                   [["poisson", 2.0], ["constant", 4.0]] if i % 29 == 7 else None) for i in range(n)]
        return dict(name="dist_mixed", topology="ring", n=n, ext_rate=[0.0 if i % 13 == 12 else 4.0 for i in range(n)], mean=0.1,
                    lat_min=0.001, jitter_mean=0.006, end_s=end_s, seed=17, profile=prof, more_sources=more,
                    sources_order="extras_first",
                    probes=[["depth", end_s / 12.0] if i % 17 == 0 else ["requests_completed", end_s / 7.5] if i % 23 == 5 else None for i in range(n)],
                    schedule=[[i, end_s * (0.08 + 0.12 * k) + 1e-9 * (i % 7)] for i in range(3, n, 19) for k in range(3)] +
                             [[n - 1, 0.5 * end_s], [n - 1, 0.5 * end_s]])
    if case == "lockstep":
        # stations 1 (first shard) and n - 2 (last shard) tick in lock-step; "extras_first" lists station n - 2's Source first
        more = [[["constant", 4.0]] if i in (1, n - 2) else None for i in range(n)]
        return dict(name="dist_lockstep", topology="ring", n=n, ext_rate=0.0, mean=0.001, lat_min=0.001, jitter_mean=0.002,
                    end_s=end_s, seed=23, more_sources=more, sources_order="extras_first")
    raise SystemExit(f"unknown case {case}")


PROTOCOLS = (("live", True, "live"), ("rounds", True, "device"), ("rounds_collective", True, "collective"), ("windows", False, "collective"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("n", type=int)
    ap.add_argument("end_s", type=float)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--windows-end-s", type=float, default=None,
                    help="horizon of the WINDOW protocol's run (default END_S): one exchange per smallest link latency, and with several "
                         "processes on one GPU every exchange costs process switches on the device")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import helpers as H
    from happy_simulator_amd.sharded import DistComm, ShardedNetwork

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        os.environ["HS_RANKS_PER_DEVICE"] = str(world)         # (the LIVE exchange: every rank's launch must be resident on the ONE device)
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    specs = {True: case_spec(args.case, args.n, args.end_s),
             False: case_spec(args.case, args.n, args.windows_end_s if args.windows_end_s is not None else args.end_s)}
    n = specs[True]["n"]
    out = {"backend": args.backend, "same_device": bool(args.same_device), "case": args.case}
    mine = {}
    # the three exchange paths: asynchronous rounds with the DEVICE-SIDE exchange (peers' buffers mapped over IPC, one word
    # all-reduced per round: the default), the same rounds over collectives (all-to-all + all-reduce), the window protocol
    for name, rounds, exchange in PROTOCOLS:
        spec = specs[rounds]
        st, net, cap, p = H.ring_arrays(spec)
        with ShardedNetwork.on_gpu(st, net, DistComm(), horizon_ns=p["end_ns"], seed=spec["seed"], device=local,
                                   log_capacity=cap, rounds=rounds, exchange=exchange) as sn:
            assert sn.device_exchange == (rounds and exchange == "device") and sn.live == (exchange == "live"), getattr(sn.comm, "peer_errors", None)
            s = sn.run_until(p["end_ns"])
            stats, counts, t, cr, ns = sn.collect(n, net.n_links)
            lo, hi = sn.shards[0].lo, sn.shards[0].hi
            # every rank holds its own stations' rows (zeros elsewhere): summed over the ranks they are the network's
            rows = np.stack([stats[k].astype(np.int64) for k in STAT_KEYS] + [counts, ns["routed"]])
            sink_digest = np.zeros(2, np.int64)
            sink_digest[0] = int(t.sum() % (1 << 61))
            sink_digest[1] = int(cr.sum() % (1 << 61))
            tt = torch.from_numpy(np.concatenate([rows.ravel(), sink_digest]))
            if args.backend == "nccl":                # RCCL reduces device tensors
                tt = tt.cuda()
            dist.all_reduce(tt)
            tt = tt.cpu()
            probes = {}
            for i, prs in enumerate(p["probe_list"]):
                if lo <= i < hi:
                    for j in range(len(prs)):
                        pt, pv = sn.read_probe(i, j)
                        probes[f"{i}.{j}"] = [int(pt.sum()), int(pv.sum()), len(pt)]
            out[name] = dict(events=int(s.events_processed), final=int(s.final_time_ns), exchanges=int(s.windows),
                             world=int(s.world), by_kind=[int(x) for x in s.events_by_kind], owns=[lo, hi])
            mine[name] = (tt.numpy().copy(), probes)
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: v[1] for k, v in mine.items()})
    if rank == 0:
        for name, rounds, _ in PROTOCOLS:
            spec = specs[rounds]
            eng, p = H.ring_engine_for_spec(spec)            # the same network on one engine
            with eng:
                eng.run_until(p["end_ns"])
                s1 = eng.summary()
                stats = eng.lp_stats()
                counts, t, cr = eng.read_sinks()
                rows = np.stack([stats[k].astype(np.int64) for k in STAT_KEYS] + [counts, eng.net_stats()["routed"]])
                want = np.concatenate([rows.ravel(), [int(t.sum() % (1 << 61)) , int(cr.sum() % (1 << 61))]])
                want_probes = {}
                for i, prs in enumerate(p["probe_list"]):
                    for j in range(len(prs)):
                        pt, pv = eng.read_probe(i, j)
                        want_probes[f"{i}.{j}"] = [int(pt.sum()), int(pv.sum()), len(pt)]
                out["single" if rounds else "single_windows"] = dict(
                    events=int(s1.events_processed), final=int(s1.final_time_ns), by_kind=[int(x) for x in s1.events_by_kind],
                    generated=[int(x) for x in stats["generated"]][:64])
            got = mine[name][0]
            # the sink digests are sums of per-rank sums mod 2^61: compare mod 2^61
            ok_rows = bool(np.array_equal(got[:-2], want[:-2]))
            ok_sinks = all(int(got[-2 + q]) % (1 << 61) == int(want[-2 + q]) % (1 << 61) for q in range(2))
            got_probes = {}
            for g in gathered:
                got_probes.update(g[name])
            out[name]["stats_equal"] = ok_rows
            out[name]["sinks_equal"] = ok_sinks
            out[name]["probes_equal"] = got_probes == want_probes
            out[name]["n_probes"] = len(want_probes)
            if not ok_rows:
                bad = np.nonzero(got[:-2] != want[:-2])[0][:8]
                out[name]["first_bad"] = [[int(b // n), int(b % n), int(got[b]), int(want[b])] for b in bad]
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
