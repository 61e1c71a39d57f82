"""A station network partitioned over several engines -- one shard per GPU, one process per GPU.

This is the MI355X form of the reference's coordinated parallel run
(`ParallelSimulation._run_coordinated` -> `WindowedCoordinator.run`, happysimulator/parallel/simulation.py:197-223,
parallel/coordinator.py:75-172): EXECUTE every partition to the window end, EXCHANGE the cross-partition events,
ADVANCE.  Differences by design:

* a partition is a contiguous block of station LPs resident in one GPU's HBM (`shard_bounds`), not a thread;
* the exchange is one all-to-all of fixed-size outbox rows (RCCL over xGMI) instead of Python lists drained on
  the main thread, and the receiver injects the messages on the device (`hs_engine_shard_inject`);
* window ends follow the global virtual time: `wend = min(end, max(prev + 1, GVT) + W - 1)` with
  `GVT = all_reduce(min)` of every rank's earliest pending work and `W` = the smallest link latency (the
  reference's `min(link.min_latency)`, parallel/simulation.py:82-87) -- idle stretches are skipped in one step;
* nothing runs past the end of a window, so the reference's windowed "time travel" drops
  cannot happen: the result is bit-identical to the single-heap run, for any number of shards.

The per-window protocol only ENQUEUES device work (engine launches and collectives share torch's current
stream); the host synchronises once every `sync_every` windows to learn how far the GVT has advanced.

`LocalComm` runs several shards inside one process on one GPU (virtual shards) with tensor copies in place of
RCCL -- same protocol, same kernels -- so the sharding logic is testable on a single-GPU box.

**Asynchronous rounds (the default, `rounds=True`).**  The window protocol pays one exchange per smallest link
latency (60 000 exchanges for 60 s of a ring with 1 ms links).  With rounds, every shard runs the ASYNCHRONOUS engine
(`hs_net_async`: per-link lower bounds, no windows inside the shard) for a few iterations, then the ranks exchange the
boundary messages (the same all-to-all) and all-reduce (MAX) one vector: the lower bound of every cross-shard link plus a
"still working" flag.  A cross link then looks to its destination like any other link of the asynchronous engine -- a
queue refilled and a bound raised between launches -- and exchange rounds follow the boundary stations' lookahead (their
next possible completion + the link floor: tens of ms) instead of the link floor alone: 24 exchanges (rounds of 64 iterations) instead of
59 968 for the 65 536-station ring on 4 shards, the same bits.  `rounds=False` keeps the window protocol.

**Device-side exchange (round 5, the default of the rounds).**  Instead of the all-to-all and the all-reduce of the bounds, every
rank maps its peers' exchange buffers (`hipIpcGetMemHandle` / `hipIpcOpenMemHandle`; handles all-gathered once) and, after a
round, WRITES its outbox rows and link bounds straight into them (`hs_engine_shard_push`: xGMI peer-to-peer stores across GPUs,
plain device memory when ranks share one).  One barrier per round -- the all-reduce of the one "still working" word, which is all
RCCL is left with on this path, as `north_star` asks -- then `hs_engine_shard_inject_ipc` takes what the peers pushed.  No
whole rows, no host staging under gloo, nothing to ranks that are not neighbours beyond an empty row header.
`exchange="collective"` keeps the all-to-all path (and is what ranks that are virtual shards of ONE process may also use).
"""
from __future__ import annotations

import ctypes as C
import time as _time
from dataclasses import dataclass

import numpy as np

MSG_WORDS = 5      # int64 words per message in an outbox / inbox row (csrc/hs_netstation.hpp kMsgWords): arrival, send time, created_at,
                   # destination << 32 | link, lineage
CAND_WORDS = 8     # a rank's candidate for the one event beyond end_time (ShardCtl::cand_out)

from . import _native as N
from .engine import NetworkArrays, StationArrays, StationEngine

INF_NS = np.iinfo(np.int64).max


def shard_bounds(n_stations: int, world: int) -> np.ndarray:
    """Contiguous block partition: rank r owns stations [lo[r], lo[r+1])."""
    base, rem = divmod(n_stations, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


# ------------------------------------------------------------------------------------------------------------
# communicators: how rows / scalars travel between shards
# ------------------------------------------------------------------------------------------------------------
class LocalComm:
    """All `world` shards live in this process (virtual shards on one device): copies instead of collectives."""

    def __init__(self, world: int):
        self.world = world
        self.local_ranks = list(range(world))

    def exchange(self, outboxes, inboxes):
        import torch

        stacked = torch.stack(outboxes)                  # [src, dst, row]
        for j, inbox in enumerate(inboxes):
            inbox.copy_(stacked[:, j, :])                # row i of shard j's inbox = what shard i sent to j

    def allreduce_min(self, scalars):
        import torch

        m = torch.stack([s.reshape(()) for s in scalars]).min()
        for s in scalars:
            s.fill_(m)

    def allreduce_max(self, vectors):
        import torch

        m = torch.stack(vectors).max(dim=0).values
        for v in vectors:
            v.copy_(m)

    def allgather_rows(self, rows):
        import torch

        return torch.stack(rows).cpu().numpy()

    def gather_rows(self, arrays):
        """Host arrays of every shard (different lengths), concatenated -- the same on every rank."""
        return np.concatenate([np.asarray(a) for a in arrays], axis=0)

    def reduce_host(self, dicts):
        return _combine(dicts)

    def connect_peers(self, shards):
        """Device-side exchange between virtual shards of one process: the buffers' addresses instead of IPC handles."""
        for s in shards:
            s.ipc_export()
        ptrs = [s.ipc_buffers() for s in sorted(shards, key=lambda x: x.rank)]
        for s in shards:
            s.peers_local([p[0] for p in ptrs], [p[1] for p in ptrs])

    def flag_barrier(self, flags):
        self.allreduce_max(flags)

    # (no connect_live: the LIVE exchange needs every shard's launch RUNNING at the same time; launches of one process on several
    #  streams may share a hardware queue and then run one after the other -- measured: the second never started, the first gave up
    #  waiting.  Virtual shards of one process keep the asynchronous rounds; one shard per process (DistComm) goes live.)
    def barrier(self):
        pass


class DistComm:
    """One shard per process: torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" for tests).

    RCCL moves device tensors directly.  gloo has no all-to-all on device tensors (and RCCL refuses two ranks on one GPU), so under
    gloo every device tensor is staged through host memory: copy out (which waits for the engine launches enqueued on torch's current
    stream), collective on the host copy, copy back.  That is the mode in which two REAL shards run as two processes on ONE GPU
    (`tests/test_gpu_dist.py`, `bench.py --fake-ranks`): the protocol, the slicing and the collectives' semantics are the multi-GPU
    run's, only the transport differs."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self._group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]
        self.stage_host = dist.get_backend(group) != "nccl"
        # how many ranks share ONE device (1 on a real multi-GPU node; `bench.py --fake-ranks R` / tests/test_gpu_dist.py: R) -- the
        # LIVE exchange needs every rank's launch resident at the same time
        import os as _os
        self.ranks_per_device = int(_os.environ.get("HS_RANKS_PER_DEVICE", "1"))

    def _staged(self, t):
        return t.cpu() if (self.stage_host and t.is_cuda) else t

    def exchange(self, outboxes, inboxes):
        src, dst = self._staged(outboxes[0]), self._staged(inboxes[0])
        self._dist.all_to_all_single(dst, src, group=self._group)
        if dst is not inboxes[0]:
            inboxes[0].copy_(dst)

    def _all_reduce(self, t, op):
        h = self._staged(t)
        self._dist.all_reduce(h, op=op, group=self._group)
        if h is not t:
            t.copy_(h)

    def allreduce_min(self, scalars):
        self._all_reduce(scalars[0], self._dist.ReduceOp.MIN)

    def allreduce_max(self, vectors):
        self._all_reduce(vectors[0], self._dist.ReduceOp.MAX)

    def connect_peers(self, shards):
        """Device-side exchange: all-gather the IPC handles of every rank's exchange buffers, map the peers' (once)."""
        (s,) = shards
        # every step that can fail on ONE rank (export: uncached allocation / IPC unsupported; attach: ranks on different
        # nodes, peer access disabled) is followed by an agreement of all ranks, so that either every rank uses the device-side
        # exchange or every rank falls back to the collective path -- never one rank raising while its peers wait in a collective.
        try:
            mine, err = s.ipc_export(), None
        except Exception as e:                     # noqa: BLE001 -- whatever it is, the peers must hear about it
            mine, err = None, f"rank {self.rank}: export: {e}"
        every = [None] * self.world
        self._dist.all_gather_object(every, (mine, err), group=self._group)
        errs = [e for _, e in every if e]
        if not errs:
            try:
                s.ipc_attach(b"".join(h for h, _ in every))
            except Exception as e:                 # noqa: BLE001
                err = f"rank {self.rank}: attach: {e}"
            oks = [None] * self.world
            self._dist.all_gather_object(oks, err, group=self._group)
            errs = [e for e in oks if e]
        if errs:
            self.peer_errors = errs
            return False
        return True

    def connect_live(self, shards, peer_links):
        """LIVE exchange: all-gather the IPC handles of every rank's link queues, map the peers' (once).  Like connect_peers, every step
        that can fail on one rank is followed by an agreement of all ranks."""
        (s,) = shards
        try:
            mine, err = s.live_export(), None
        except Exception as e:                     # noqa: BLE001
            mine, err = None, f"rank {self.rank}: export: {e}"
        every = [None] * self.world
        self._dist.all_gather_object(every, (mine, err), group=self._group)
        errs = [e for _, e in every if e]
        if not errs:
            try:
                s.live_attach(b"".join(h for h, _ in every), peer_links[s.rank])
            except Exception as e:                 # noqa: BLE001
                err = f"rank {self.rank}: attach: {e}"
            oks = [None] * self.world
            self._dist.all_gather_object(oks, err, group=self._group)
            errs = [e for e in oks if e]
        if errs:
            self.peer_errors = errs
            return False
        return True

    def barrier(self):
        self._dist.barrier(group=self._group)

    def flag_barrier(self, flags):
        """The one collective of a round on the device-side exchange path: all-reduce(MAX) of the "still working" word.  It is also the
        barrier between the ranks' pushes and their injects: stream-ordered under RCCL (every rank's push precedes its share of the
        all-reduce on its stream); under gloo the staging copy waits for the stream first."""
        self._all_reduce(flags[0], self._dist.ReduceOp.MAX)

    def allgather_rows(self, rows):
        import torch

        row = self._staged(rows[0])
        out = [torch.empty_like(row) for _ in range(self.world)]
        self._dist.all_gather(out, row, group=self._group)
        return torch.stack(out).cpu().numpy()

    def gather_rows(self, arrays):
        every = [None] * self.world
        self._dist.all_gather_object(every, np.concatenate([np.asarray(a) for a in arrays], axis=0), group=self._group)
        return np.concatenate(every, axis=0)

    def reduce_host(self, dicts):
        import torch

        local = _combine(dicts)
        dev = "cpu" if self.stage_host else "cuda"
        out = {}
        for k in sorted(local):
            v = local[k]
            if isinstance(v, np.ndarray):
                t = torch.as_tensor(v, dtype=torch.int64, device=dev).clone()
            else:
                t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX if k.startswith("max_") else self._dist.ReduceOp.SUM,
                                  group=self._group)
            out[k] = t.cpu().numpy() if isinstance(v, np.ndarray) else int(t.item())
        return out


def _combine(dicts):
    out = {}
    for d in dicts:
        for k, v in d.items():
            if k not in out:
                out[k] = v.copy() if isinstance(v, np.ndarray) else v
            elif k.startswith("max_"):
                out[k] = max(out[k], v)
            else:
                out[k] = out[k] + v
    return out


# ------------------------------------------------------------------------------------------------------------
# splitting a network description into shards
# ------------------------------------------------------------------------------------------------------------
def shard_arrays(stations: StationArrays, net: NetworkArrays, lo: int, hi: int):
    """The slice [lo, hi) of a network as (StationArrays, NetworkArrays) for one engine: local stations, every link
    that starts or ends in the slice (network-wide endpoints + ids), router / link references re-indexed."""
    n = stations.n
    sl = slice(lo, hi)
    base = stations.stream_base if stations.stream_base is not None else np.arange(n, dtype=np.uint64)
    st = StationArrays(
        n=hi - lo, src_kind=stations.src_kind[sl], src_rate=stations.src_rate[sl],
        src_stop_after_ns=stations.src_stop_after_ns[sl], concurrency=stations.concurrency[sl],
        svc_kind=stations.svc_kind[sl], svc_mean_s=stations.svc_mean_s[sl], queue_cap=stations.queue_cap[sl],
        egress=stations.egress[sl], seed=None if stations.seed is None else stations.seed[sl],
        stream_base=np.asarray(base, np.uint64)[sl],
        # (not lowered on shards yet: passed on so that the engine refuses them instead of losing them silently)
        src_profile_kind=None if stations.src_profile_kind is None else stations.src_profile_kind[sl],
        src_profile_params=None if stations.src_profile_params is None else stations.src_profile_params[sl],
        probe_metric=None if stations.probe_metric is None else stations.probe_metric[sl],
        probe_interval_s=None if stations.probe_interval_s is None else stations.probe_interval_s[sl],
        src_more_kind=None if stations.src_more_kind is None else np.asarray(stations.src_more_kind)[:, sl],
        src_more_rate=None if stations.src_more_kind is None else np.asarray(stations.src_more_rate)[:, sl],
        src_more_stop_after_ns=(None if stations.src_more_stop_after_ns is None
                                else np.asarray(stations.src_more_stop_after_ns)[:, sl]),
        probe_metric_more=None if stations.probe_metric_more is None else np.asarray(stations.probe_metric_more)[:, sl],
        probe_interval_more=None if stations.probe_interval_more is None else np.asarray(stations.probe_interval_more)[:, sl],
        sched_off=None if stations.sched_off is None else (np.asarray(stations.sched_off)[lo:hi + 1] - int(stations.sched_off[lo])),
        sched_time_ns=None if stations.sched_off is None else np.asarray(stations.sched_time_ns)[
            int(stations.sched_off[lo]):int(stations.sched_off[hi])])
    # the reference's construction order (`sources=[...]`, `probes=[...]`, the Events handed to schedule()): the shard keeps the
    # relative order of what it owns -- the election's last key (csrc/hs_kernels.hpp cand_rank) and, on one engine, the prologue
    if stations.sched_off is not None and getattr(stations, "sched_rank", None) is not None:
        st.sched_rank = np.asarray(stations.sched_rank, np.int64)[int(stations.sched_off[lo]):int(stations.sched_off[hi])]
    for order_name, slot_name in (("source_order", "source_slot_order"), ("probe_order", "probe_slot_order")):
        order = getattr(stations, order_name, None)
        if order is None:
            continue
        order = np.asarray(order, np.int64)
        keep = (order >= lo) & (order < hi)
        setattr(st, order_name, (order[keep] - lo).astype(np.int32))
        slots = getattr(stations, slot_name, None)
        if slots is not None:
            setattr(st, slot_name, np.asarray(slots, np.uint8)[keep])
    src, dst = np.asarray(net.link_src), np.asarray(net.link_dst)
    touch = ((src >= lo) & (src < hi)) | ((dst >= lo) & (dst < hi))
    gids = np.nonzero(touch)[0].astype(np.int64)
    local_of = np.full(net.n_links, -1, np.int64)
    local_of[gids] = np.arange(len(gids))

    def remap(a):
        a = np.asarray(a, np.int64)[sl]
        return np.where(a >= 0, local_of[np.clip(a, 0, max(net.n_links - 1, 0))], a).astype(np.int32)

    lbase = net.link_stream_base if net.link_stream_base is not None else np.asarray(base, np.uint64)[src]
    rbase = net.router_stream_base if net.router_stream_base is not None else np.asarray(base, np.uint64)
    sub = NetworkArrays(
        egress_kind=net.egress_kind[sl], router_target0=remap(net.router_target0),
        router_target1=remap(net.router_target1), link_of=remap(net.link_of),
        link_src=src[gids].astype(np.int32), link_dst=dst[gids].astype(np.int32),
        link_lat_min_s=np.asarray(net.link_lat_min_s)[gids], link_jitter_kind=np.asarray(net.link_jitter_kind)[gids],
        link_jitter_mean_s=np.asarray(net.link_jitter_mean_s)[gids],
        router_stream_base=np.asarray(rbase, np.uint64)[sl], link_stream_base=np.asarray(lbase, np.uint64)[gids],
        link_loss_rate=None if net.link_loss_rate is None else np.asarray(net.link_loss_rate, np.float64)[gids],
        router_n_targets=None if net.router_n_targets is None else np.asarray(net.router_n_targets)[sl],
        router_target2=None if net.router_target2 is None else remap(net.router_target2),
        router_target3=None if net.router_target3 is None else remap(net.router_target3),
        # (a loss table belongs to the shard that owns the link's SOURCE station: that is where packets enter the link)
        link_drop_capacity=None if net.link_drop_capacity is None else np.where(
            (src[gids] >= lo) & (src[gids] < hi), np.asarray(net.link_drop_capacity, np.int64)[gids], 0),
        bag_capacity=net.bag_capacity, n_global_lp=n, link_gid=gids, n_global_links=net.n_links)
    return st, sub


MAX_XSRC, MAX_PROBES = 3, 4     # csrc/hs_station.hpp kMaxXSrc / kMaxProbes


class ElectionRanks:
    """The last key of the election of the one event beyond `end_time`, NETWORK-WIDE: the position of the candidate's entity in the
    reference's construction order (`sources=[...]`, then sourceless stations, then `probes=[...]`) -- the table
    `hs_engine_set_stations` builds for one engine (csrc/hs_engine.hip, `tie_rank`; csrc/hs_station.hpp `cand_rank`), built here
    from the unsharded description.  A shard's engine only knows the relative order of its OWN entities (its `source_order` /
    `probe_order` are filtered and re-based, `shard_arrays`), so a rank it reports cannot be compared with another shard's: the
    host looks the winner's rank up by (station, kind) instead (`hs_shard.cand_dev` words 3 and 7)."""

    def __init__(self, stations: StationArrays):
        n = self.n = int(stations.n)
        # no order given: the engine ranks by LP position (cand_rank's closed form, `tie_rank == nullptr`)
        self.lp_order = stations.source_order is None and stations.probe_order is None
        kinds = np.zeros((1 + MAX_XSRC, n), np.uint8)
        kinds[0] = np.asarray(stations.src_kind)
        if stations.src_more_kind is not None:
            kinds[1:] = np.asarray(stations.src_more_kind)
        has = kinds != N.SRC_NONE
        if stations.source_order is not None:
            so = np.asarray(stations.source_order, np.int64)
            ss = (np.zeros(len(so), np.int64) if stations.source_slot_order is None
                  else np.asarray(stations.source_slot_order, np.int64))
        else:                                                  # LP-major, slot-minor
            lp, slot = np.nonzero(has.T)
            so, ss = lp.astype(np.int64), slot.astype(np.int64)
        self.tick = np.full((1 + MAX_XSRC, n), -1, np.int64)   # a tick: its own Source's position
        self.tick[ss, so] = np.arange(len(so))
        self.first = np.full(n, -1, np.int64)                  # anything else of the station: its first-listed Source
        for q in range(len(so) - 1, -1, -1):
            self.first[so[q]] = q
        none = self.first < 0
        self.first[none] = len(so) + np.nonzero(none)[0]       # sourceless stations after every Source
        pm = np.full((MAX_PROBES, n), N.PROBE_NONE, np.uint8)
        if stations.probe_metric is not None:
            pm[0] = np.asarray(stations.probe_metric)
        if stations.probe_metric_more is not None:
            pm[1:] = np.asarray(stations.probe_metric_more)
        if stations.probe_order is not None:
            po = np.asarray(stations.probe_order, np.int64)
            ps = (np.zeros(len(po), np.int64) if stations.probe_slot_order is None
                  else np.asarray(stations.probe_slot_order, np.int64))
        else:
            lp, slot = np.nonzero((pm != N.PROBE_NONE).T)
            po, ps = lp.astype(np.int64), slot.astype(np.int64)
        self.probe = np.full((MAX_PROBES, n), -1, np.int64)    # Probes behind all of them, each by its own position
        self.probe[ps, po] = len(so) + n + np.arange(len(po))

    def rank(self, station: int, kind: int) -> int:
        if self.lp_order:
            if kind >= 8:
                return self.n * (MAX_XSRC + 1) + station * MAX_PROBES + (kind - 8)
            return station * (MAX_XSRC + 1) + (kind - 2 if kind >= 2 else 0)
        if kind >= 8:
            return int(self.probe[kind - 8, station])
        if kind >= 2:
            return int(self.tick[kind - 2, station])
        return int(self.first[station])


# ------------------------------------------------------------------------------------------------------------
# one shard on the GPU
# ------------------------------------------------------------------------------------------------------------
class GpuShard:
    """One engine + its exchange tensors (torch owns the device memory so torch.distributed can move it)."""

    def __init__(self, stations: StationArrays, net: NetworkArrays, rank: int, bounds: np.ndarray, *, horizon_ns: int,
                 start_ns: int = 0, seed: int = 42, device: int = 0, msg_capacity: int = 256, log_capacity: int = 0):
        import torch

        self.rank, self.world = rank, len(bounds) - 1
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.bounds = np.ascontiguousarray(bounds, np.int64)
        self.gids = np.asarray(net.link_gid, np.int64)
        self.msg_capacity = msg_capacity
        dev = torch.device("cuda", device)
        row = 1 + MSG_WORDS * msg_capacity
        self.outbox = torch.zeros((self.world, row), dtype=torch.int64, device=dev)
        self.inbox = torch.zeros((self.world, row), dtype=torch.int64, device=dev)
        self.gvt = torch.zeros(2, dtype=torch.int64, device=dev)
        self.cand = torch.zeros(CAND_WORDS, dtype=torch.int64, device=dev)
        self.engine = StationEngine(stations, mode=N.MODE_SINGLE, horizon_ns=horizon_ns, start_ns=start_ns, seed=seed,
                                    lp_base=self.lo, device=device, log_capacity=log_capacity, network=net)
        self.engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.local_window_ns = self.engine.summary().window_ns
        self.has_more_sources = stations.src_more_kind is not None
        self._attached = False

    def attach(self, window_ns: int):
        sh = N.Shard(self.rank, self.world, self.bounds.ctypes.data, self.outbox.data_ptr(), self.inbox.data_ptr(),
                     self.msg_capacity, 0, int(window_ns), self.gvt.data_ptr(), self.cand.data_ptr())
        e = self.engine
        e._check(e._lib.hs_engine_shard_attach(e._h, C.byref(sh)))
        self._attached = True

    # -- the per-window protocol (enqueue only) ------------------------------------------------------------
    def begin(self, end_ns):
        e = self.engine
        e._check(e._lib.hs_engine_shard_begin(e._h, int(end_ns)))

    def window(self, k):
        e = self.engine
        e._check(e._lib.hs_engine_shard_window(e._h, k))

    def inject(self, k):
        e = self.engine
        e._check(e._lib.hs_engine_shard_inject(e._h, k))

    def gvt_slot(self, k):
        return self.gvt[(k & 1):(k & 1) + 1]

    def progress(self, k_last) -> int:
        e = self.engine
        w = C.c_int64(0)
        e._check(e._lib.hs_engine_shard_progress(e._h, k_last, C.byref(w)))
        return int(w.value)

    def final(self, k):
        e = self.engine
        e._check(e._lib.hs_engine_shard_final(e._h, k))

    # -- asynchronous exchange rounds (enqueue only, except round_done) ----------------------------------------
    def async_setup(self, cross_gid: np.ndarray, max_iters: int):
        import torch

        self.cross_gid = np.ascontiguousarray(cross_gid, np.int64)
        self.xbounds = torch.zeros(len(self.cross_gid) + 1, dtype=torch.int64, device=self.outbox.device)
        e = self.engine
        e._check(e._lib.hs_engine_shard_async_setup(e._h, len(self.cross_gid), self.cross_gid.ctypes.data,
                                                    self.xbounds.data_ptr(), int(max_iters)))

    def round(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_round(e._h))

    def inject_async(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_inject_async(e._h))

    # -- device-side exchange (hs_engine_shard_ipc_*) ----------------------------------------------------------
    # -- LIVE exchange: one launch per run, the ranks' kernels talk through each other's link queues (hs_engine_shard_live_*) -----
    def live_export(self) -> bytes:
        e = self.engine
        buf = (C.c_char * (3 * N.IPC_HANDLE_BYTES))()
        e._check(e._lib.hs_engine_shard_live_export(e._h, buf))
        return bytes(buf)

    def live_attach(self, all_handles: bytes, peer_link: np.ndarray):
        e = self.engine
        if len(all_handles) != 3 * N.IPC_HANDLE_BYTES * self.world:
            raise ValueError("expected three handles per rank")
        pl = np.ascontiguousarray(peer_link, np.int32)
        e._check(e._lib.hs_engine_shard_live_attach(e._h, all_handles, pl.ctypes.data))

    def live_run(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_live_run(e._h))

    def live_wait(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_live_wait(e._h))

    def ipc_export(self) -> bytes:
        e = self.engine
        buf = (C.c_char * (2 * N.IPC_HANDLE_BYTES))()
        e._check(e._lib.hs_engine_shard_ipc_export(e._h, buf))
        return bytes(buf)

    def ipc_attach(self, all_handles: bytes):
        e = self.engine
        if len(all_handles) != 2 * N.IPC_HANDLE_BYTES * self.world:
            raise ValueError("expected two handles per rank")
        e._check(e._lib.hs_engine_shard_ipc_attach(e._h, all_handles))

    def ipc_buffers(self):
        e = self.engine
        a, b = C.c_void_p(), C.c_void_p()
        e._check(e._lib.hs_engine_shard_ipc_buffers(e._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def peers_local(self, inbox_ptrs, bounds_ptrs):
        e = self.engine
        pi = (C.c_void_p * self.world)(*inbox_ptrs)
        pb = (C.c_void_p * self.world)(*bounds_ptrs)
        e._check(e._lib.hs_engine_shard_peers_local(e._h, pi, pb))

    def push(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_push(e._h))

    def inject_ipc(self):
        e = self.engine
        e._check(e._lib.hs_engine_shard_inject_ipc(e._h))

    def flag_word(self):
        return self.xbounds[-1:]

    def round_done(self) -> bool:
        e = self.engine
        flag = C.c_int32(0)
        e._check(e._lib.hs_engine_shard_async_done(e._h, C.byref(flag)))
        return flag.value == 0

    def overshoot(self, lp_local):
        e = self.engine
        e._check(e._lib.hs_engine_shard_overshoot(e._h, int(lp_local)))

    def totals(self) -> dict:
        s = self.engine.summary()
        return {"events": s.events_processed, "by_kind": s.events_by_kind.copy(), "completed": s.requests_completed,
                "sink_records": s.sink_records, "max_final_ns": s.final_time_ns, "launches": s.launches}

    def close(self):
        self.engine.close()


@dataclass
class ShardedSummary:
    events_processed: int
    events_by_kind: np.ndarray
    requests_completed: int
    sink_records: int
    final_time_ns: int
    windows: int
    window_ns: int
    world: int
    run_seconds: float = 0.0        # host wall clock of run_until
    exchange_seconds: float = 0.0   # ... of which in EXCHANGE / GVT / inject calls (coordinator.py:105-109's barrier time)


class ShardedNetwork:
    """Drives the shards this process owns through the window protocol.  `shards` are the local shard objects
    (GpuShard, or any object with the same methods), `comm` moves rows and scalars between all shards."""

    def __init__(self, shards: list, comm, *, window_ns: int, sync_every: int = 64, rounds: bool = False,
                 ranks: "ElectionRanks | None" = None, device_exchange: bool = False, live: bool = False):
        self.shards = shards
        self.ranks = ranks               # network-wide construction ranks for the election across shards (None: trust word 6)
        self.comm = comm
        self.window_ns = int(window_ns)
        self.sync_every = max(1, int(sync_every))
        self.windows = 0
        self.rounds = bool(rounds)       # asynchronous exchange rounds instead of windows (GpuShard.async_setup done)
        self.device_exchange = bool(device_exchange)   # ... with the ranks pushing into each other's buffers (comm.connect_peers done)
        self.live = bool(live)           # ONE launch per rank and run, the kernels exchange while they run (comm.connect_live done)

    @classmethod
    def on_gpu(cls, stations: StationArrays, net: NetworkArrays, comm, *, horizon_ns: int, start_ns: int = 0,
               seed: int = 42, device: int = 0, msg_capacity: int = 256, log_capacity: int = 0,
               sync_every: int | None = None, bounds: np.ndarray | None = None, rounds: bool = True,
               round_iters: int = 64, exchange: str = "live"):
        """Partition `stations` / `net` (network-wide descriptions, identical on every rank) over comm.world shards
        and build the shards this process owns on `device`.  `bounds` (world + 1 station offsets) overrides the
        balanced block partition, e.g. with the user's own SimulationPartition sizes."""
        import torch

        bounds = shard_bounds(stations.n, comm.world) if bounds is None else np.asarray(bounds, np.int64)
        if len(bounds) != comm.world + 1 or bounds[0] != 0 or bounds[-1] != stations.n or (np.diff(bounds) <= 0).any():
            raise ValueError("bounds must be world + 1 increasing station offsets covering every station")
        shards = []
        for r in comm.local_ranks:
            st, sub = shard_arrays(stations, net, int(bounds[r]), int(bounds[r + 1]))
            shards.append(GpuShard(st, sub, r, bounds, horizon_ns=horizon_ns, start_ns=start_ns, seed=seed,
                                   device=device, msg_capacity=msg_capacity, log_capacity=log_capacity))
        # the lookahead is a property of the whole network: min over all shards' links
        dev = shards[0].gvt.device
        w = [torch.tensor([s.local_window_ns], dtype=torch.int64, device=dev) for s in shards]
        comm.allreduce_min(w)
        window_ns = int(w[0].item())
        for s in shards:
            s.attach(window_ns)
        if rounds:
            # the network's cross-shard links (the same list on every rank): their bounds travel between the rounds
            src, dst = np.asarray(net.link_src, np.int64), np.asarray(net.link_dst, np.int64)
            rank_of = lambda x: np.searchsorted(bounds, x, side="right") - 1          # noqa: E731
            cross = np.nonzero(rank_of(src) != rank_of(dst))[0].astype(np.int64)
            for s in shards:
                s.async_setup(cross, round_iters)
        if exchange not in ("device", "collective", "live"):
            raise ValueError("exchange must be 'live', 'device' or 'collective'")
        live = False
        if rounds and exchange == "live" and hasattr(comm, "connect_live"):
            # LIVE exchange (round 6): ONE launch per rank and run; the kernels exchange messages and bounds through each other's link
            # queues while they run.  Every link that leaves a shard needs its index in the destination rank's link table (the table
            # of a shard = the links that touch it, in network order: shard_arrays).  The launches wait for one another, so what this
            # process puts on its device must be resident together: one workgroup of 256 stations per CU.
            src, dst = np.asarray(net.link_src, np.int64), np.asarray(net.link_dst, np.int64)
            rank_of = lambda x: np.searchsorted(bounds, x, side="right") - 1          # noqa: E731
            rs, rd = rank_of(src), rank_of(dst)
            tables = {}
            for r in range(comm.world):
                lo, hi = int(bounds[r]), int(bounds[r + 1])
                tables[r] = np.nonzero(((src >= lo) & (src < hi)) | ((dst >= lo) & (dst < hi)))[0]
            peer_links = {}
            for s in shards:
                pl = np.full(len(s.gids), -1, np.int32)
                for l, g in enumerate(s.gids):
                    if rs[g] == s.rank and rd[g] != s.rank:
                        pl[l] = int(np.searchsorted(tables[int(rd[g])], g))
                peer_links[s.rank] = pl
            blocks = sum(-(-(s.hi - s.lo) // 256) for s in shards)
            cus = torch.cuda.get_device_properties(device).multi_processor_count
            fits = blocks <= cus // max(1, getattr(comm, "ranks_per_device", 1))
            live = fits and comm.connect_live(shards, peer_links) is not False
        device_exchange = rounds and not live and exchange in ("device", "live") and hasattr(comm, "connect_peers")
        if device_exchange and comm.connect_peers(shards) is False:
            device_exchange = False          # (agreed by all ranks: the collective exchange path, comm.peer_errors says why)
        if sync_every is None:           # exchanges between host synchronisations: a run is ~25 rounds or ~60 000 windows
            sync_every = 4 if rounds else 64
        return cls(shards, comm, window_ns=window_ns, sync_every=sync_every, rounds=rounds, ranks=ElectionRanks(stations),
                   device_exchange=device_exchange, live=live)

    def _run_rounds(self, end_ns: int) -> int:
        """Asynchronous rounds: every shard runs the asynchronous engine for a few iterations, then messages (all-to-all)
        and the cross links' lower bounds (all-reduce MAX) are exchanged.  Returns the number of rounds."""
        sh, comm = self.shards, self.comm
        r = 0
        while True:
            for _ in range(self.sync_every):
                for s in sh:
                    s.round()                                          # EXECUTE (one cooperative launch per shard)
                t0 = _time.perf_counter()
                if self.device_exchange:
                    for s in sh:
                        s.push()                                       # EXCHANGE: straight into the peers' buffers ...
                    comm.flag_barrier([s.flag_word() for s in sh])     # ... one word all-reduced = the round's barrier
                    for s in sh:
                        s.inject_ipc()
                else:
                    comm.exchange([s.outbox for s in sh], [s.inbox for s in sh])   # EXCHANGE messages ...
                    comm.allreduce_max([s.xbounds for s in sh])         # ... and bounds (+ the "still working" flag)
                    for s in sh:
                        s.inject_async()
                self._exchange_s += _time.perf_counter() - t0
                r += 1
            done = self._all_ranks([lambda s=s: s.round_done() for s in sh])   # the only host synchronisation
            if all(done):
                break
        return r

    def _all_ranks(self, calls):
        """Run the shards' host-synchronising calls and make an engine error COLLECTIVE: an error flag is all-reduced (MAX)
        after them, so that a rank whose shard overflowed does not leave the others waiting in the next collective -- every
        rank raises together (the failing one its own EngineError, the others a note that a peer failed)."""
        import torch

        out, exc = [], None
        for c in calls:
            try:
                out.append(c())
            except N.EngineError as e:
                exc = exc or e
                out.append(False)
        dev = self.shards[0].gvt.device
        flags = [torch.tensor([1 if exc is not None else 0], dtype=torch.int64, device=dev) for _ in self.shards]
        self.comm.allreduce_max(flags)
        if exc is not None:
            raise exc
        if int(flags[0].item()) != 0:
            raise N.EngineError(N.HS_E_STATE, "another rank's shard reported an engine error (see that rank's message)")
        return out

    def run_until(self, end_ns: int) -> ShardedSummary:
        sh, comm = self.shards, self.comm
        wall0 = _time.perf_counter()
        self._exchange_s = 0.0
        for s in sh:
            s.begin(end_ns)
        k = 0
        if self.live:
            comm.barrier()                                             # every rank has reset its queues: now the peers may write
            for s in sh:
                s.live_run()                                           # EXECUTE + EXCHANGE: one launch per rank, they talk while they run
            self._all_ranks([lambda s=s: s.live_wait() for s in sh])
            comm.barrier()                                             # (nobody resets while a peer still writes)
            k = 1
            for s in sh:
                s.final(0)
        elif self.rounds:
            k = self._run_rounds(end_ns)
            for s in sh:
                s.final(0)
        else:
            while True:
                for _ in range(self.sync_every):
                    for s in sh:
                        s.window(k)                                    # EXECUTE
                    t0 = _time.perf_counter()
                    comm.exchange([s.outbox for s in sh], [s.inbox for s in sh])   # EXCHANGE
                    for s in sh:
                        s.inject(k)
                    comm.allreduce_min([s.gvt_slot(k) for s in sh])    # GVT
                    self._exchange_s += _time.perf_counter() - t0
                    k += 1
                wends = self._all_ranks([lambda s=s: s.progress(k - 1) for s in sh])   # the only host synchronisation
                if min(wends) >= end_ns:
                    break
            for s in sh:
                s.final(k)
        self.windows = k
        # [world, CAND_WORDS]: valid, t, t_created, station, steps from its group's root, that root's creation time, the construction
        # rank among the shard's own entities, what the candidate is
        cands = comm.allgather_rows([s.cand for s in sh])
        valid = cands[cands[:, 0] != 0]
        winner_t = None
        if len(valid):
            # the election's key (csrc/hs_kernels.hpp cand_less): time, creation time, lineage, construction rank -- the last one
            # network-wide: a shard's word 6 only orders that shard's entities (ElectionRanks)
            rank = valid[:, 6] if self.ranks is None else np.array(
                [self.ranks.rank(int(c[3]), int(c[7])) for c in valid], np.int64)
            order = np.lexsort((valid[:, 3], rank, valid[:, 5], valid[:, 4], valid[:, 2], valid[:, 1]))
            win = valid[order[0]]
            # Did the election rest on the construction rank of a stand-in (csrc/hs_kernels.hpp hs_net_window, round 5)?  Inside the
            # winner's shard: bit 1 of its first word.  Across shards: another rank's candidate shares the winner's whole lineage key
            # and one of the two is a departure / a message / an injected Request (word 7 < 2), which ranks with its station's
            # first-listed Source instead of the Source its lineage goes back to.  Every rank sees the same rows: all raise together.
            same = [c for c in valid[order[1:]] if (c[1], c[2], c[4], c[5]) == (win[1], win[2], win[4], win[5])]
            # (a LOSING shard's best candidate that tied in-shard with a stand-in -- its bit 1 -- and shares the winner's key
            #  means that stand-in ties the winner too)
            if (int(win[0]) & 2) or any(int(c[7]) < 2 or int(win[7]) < 2 or (int(c[0]) & 2) for c in same):
                raise N.EngineError(N.HS_E_UNSUPPORTED,
                                    "the one event beyond end_time is a lock-step tie between two stations that only the reference's "
                                    "sort-index ledger decides (one of them a departure, a message or an injected Request): refused "
                                    "instead of guessing")
            t, station = int(win[1]), int(win[3])
            winner_t = t
            for s in sh:
                if s.lo <= station < s.hi:
                    s.overshoot(station - s.lo)                        # the one event beyond end_time
        tot = comm.reduce_host([s.totals() for s in sh])
        final = winner_t if winner_t is not None else int(tot["max_final_ns"])
        return ShardedSummary(events_processed=int(tot["events"]), events_by_kind=np.asarray(tot["by_kind"]),
                              requests_completed=int(tot["completed"]), sink_records=int(tot["sink_records"]),
                              final_time_ns=final, windows=k, window_ns=self.window_ns, world=comm.world,
                              run_seconds=_time.perf_counter() - wall0, exchange_seconds=self._exchange_s)

    def collect(self, n_stations: int, n_links: int):
        """Whole-network results from this process's shards, as one engine would report them: (lp_stats dict, sink counts,
        sink t, sink created_at, net_stats dict).  Per-link counters are summed over the two ends' shards (each end keeps
        the counters it owns, the other end reports 0)."""
        stats, counts, ts, crs = {}, np.zeros(n_stations, np.int64), [], []
        net = {"routed": np.zeros(n_stations, np.int64), "link_entered": np.zeros(n_links, np.int64),
               "link_packets_sent": np.zeros(n_links, np.int64), "link_packets_dropped": np.zeros(n_links, np.int64)}
        for s in sorted(self.shards, key=lambda x: x.lo):
            for k, v in s.engine.lp_stats().items():
                stats.setdefault(k, np.zeros(n_stations, v.dtype))[s.lo:s.hi] = v
            if getattr(s, "has_more_sources", False):      # several Sources per Server: Source.generated_count of slots 1..3
                more = stats.setdefault("generated_more", [np.zeros(n_stations, np.int64) for _ in range(3)])
                for slot in range(3):
                    more[slot][s.lo:s.hi] = s.engine.source_generated(1 + slot)
            c, t, cr = s.engine.read_sinks()
            counts[s.lo:s.hi] = c
            ts.append(t)
            crs.append(cr)
            ns = s.engine.net_stats()
            net["routed"][s.lo:s.hi] = ns["routed"]
            for k in ("link_entered", "link_packets_sent", "link_packets_dropped"):
                net[k][s.gids] += ns[k]
        return stats, counts, np.concatenate(ts), np.concatenate(crs), net

    def read_probe(self, station: int, slot: int = 0):
        """Samples of the Probe in `slot` of network-wide station `station` (owned by one of this process's shards)."""
        for s in self.shards:
            if s.lo <= station < s.hi:
                return s.engine.read_probe(station - s.lo, slot)
        raise IndexError(f"station {station} is not on this process's shards")

    def close(self):
        for s in self.shards:
            s.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
