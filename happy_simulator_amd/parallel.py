"""Mirror of happysimulator/parallel for the hot path, re-designed for one GPU per process.

* `ParallelRunner.run_replicas / run_sweep` (parallel/runner.py:82-142): the reference forks one process per
  replica and seeds `random` with `base_seed + i`.  Here all replicas are lowered into ONE engine launch --
  one LP per lane, `HS_MODE_REPLICAS`, per-LP Philox key `base_seed + i` -- so 4 096 replicas cost one kernel.
* `ParallelSimulation` without links (parallel/simulation.py:170-195): every partition is an independent
  Simulation; same batching.  With links (parallel/simulation.py:197-223): one shard of the windowed network
  engine per partition (happy_simulator_amd/sharded.py) -- a GPU per partition under torch.distributed.
* Across GPUs (one process per GPU, torch.distributed): `shard_range` block-partitions replicas / LPs over
  ranks and `reduce_summaries` combines the per-rank totals (SUM of events, MAX of final time).  There is no
  data-path collective because the units are independent.
"""
from __future__ import annotations

import time as _time
from dataclasses import dataclass, field
from typing import Any, Callable

import numpy as np

from . import _native as N
from .core.temporal import Instant
from .engine import StationArrays, StationEngine
from .entities import Probe
from .graph_engine import GeneralGraph, GraphEngine
from .lowering import UnsupportedTopology, write_back, write_back_probes
from .simulation import Simulation, entity_summaries
from .summary import SimulationSummary


@dataclass
class RunConfig:
    name: str
    build_fn: Callable
    seed: int | None = None


@dataclass
class ParallelResult:
    name: str
    summary: SimulationSummary
    artifacts: dict[str, Any] = field(default_factory=dict)


def shard_range(n_units: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block partition of `n_units` independent units over ranks: [lo, hi)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(n_units, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_summaries(local: dict, group=None) -> dict:
    """All-reduce a dict of per-rank totals: keys starting with `max_` use MAX, everything else SUM.
    With no initialised process group this is the identity (single process)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    out = {}
    for k in sorted(local):
        t = torch.tensor([local[k]], dtype=torch.int64 if isinstance(local[k], (int, np.integer)) else torch.float64,
                         device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if k.startswith("max_") else dist.ReduceOp.SUM, group=group)
        out[k] = t.item()
    return out


def _concat(arrs: list[StationArrays]) -> StationArrays:
    """One StationArrays for several Simulations' stations.  EVERY field LoweredGraph.arrays() / Simulation._schedule_arrays
    can produce is carried -- time-varying profiles, probes, scheduled Requests -- with the field's default where a
    Simulation does not use it (dropping them would silently run a ramp source at its peak rate, lose probe samples, ignore
    schedule() calls)."""
    keys = ("src_kind", "src_rate", "src_stop_after_ns", "concurrency", "svc_kind", "svc_mean_s", "queue_cap", "egress")
    n = sum(a.n for a in arrs)
    out = StationArrays(n=n, **{k: np.concatenate([getattr(a, k) for a in arrs]) for k in keys})
    if any(a.src_profile_kind is not None for a in arrs):
        out.src_profile_kind = np.concatenate([a.src_profile_kind if a.src_profile_kind is not None
                                               else np.zeros(a.n, np.uint8) for a in arrs])
        out.src_profile_params = np.concatenate([a.src_profile_params if a.src_profile_params is not None
                                                 else np.zeros((a.n, 4), np.float64) for a in arrs])
    if any(a.probe_metric is not None for a in arrs):
        out.probe_metric = np.concatenate([a.probe_metric if a.probe_metric is not None
                                           else np.full(a.n, N.PROBE_NONE, np.uint8) for a in arrs])
        out.probe_interval_s = np.concatenate([a.probe_interval_s if a.probe_interval_s is not None
                                               else np.ones(a.n, np.float64) for a in arrs])
    if any(a.probe_metric_more is not None for a in arrs):
        out.probe_metric_more = np.concatenate([a.probe_metric_more if a.probe_metric_more is not None
                                                else np.full((3, a.n), N.PROBE_NONE, np.uint8) for a in arrs], axis=1)
        out.probe_interval_more = np.concatenate([a.probe_interval_more if a.probe_interval_more is not None
                                                  else np.ones((3, a.n), np.float64) for a in arrs], axis=1)
    if any(a.src_more_kind is not None for a in arrs):
        out.src_more_kind = np.concatenate([a.src_more_kind if a.src_more_kind is not None
                                            else np.full((3, a.n), N.SRC_NONE, np.uint8) for a in arrs], axis=1)
        out.src_more_rate = np.concatenate([a.src_more_rate if a.src_more_kind is not None
                                            else np.ones((3, a.n), np.float64) for a in arrs], axis=1)
        out.src_more_stop_after_ns = np.concatenate([a.src_more_stop_after_ns if a.src_more_kind is not None
                                                     else np.full((3, a.n), -1, np.int64) for a in arrs], axis=1)
    if any(a.sched_off is not None for a in arrs):
        offs, times, ranks, base = [np.zeros(1, np.int64)], [], [], 0
        for a in arrs:
            o = np.asarray(a.sched_off, np.int64) if a.sched_off is not None else np.zeros(a.n + 1, np.int64)
            t = np.asarray(a.sched_time_ns, np.int64) if a.sched_off is not None else np.zeros(0, np.int64)
            r = (np.asarray(a.sched_rank, np.int64) if getattr(a, "sched_rank", None) is not None
                 else np.arange(len(t), dtype=np.int64))
            offs.append(o[1:] + base)
            times.append(t)
            ranks.append(r)                               # every Simulation numbers its own Events (one prologue per LP)
            base += len(t)
        out.sched_off = np.concatenate(offs)
        out.sched_time_ns = np.concatenate(times)
        out.sched_rank = np.concatenate(ranks)
    return out


def write_back_probes_sharded(g, sn) -> None:
    """Probe samples of the stations this process's shards own."""
    for i, stn in enumerate(g.stations):
        if stn.probes and any(s.lo <= i < s.hi for s in sn.shards):
            for slot, pr in enumerate(stn.probes):
                t, v = sn.read_probe(i, slot)
                pr.data_sink._set(t, v, Probe.value_map(pr.metric, stn.server))


class _LpOffset:
    """read_probe() of one Simulation's stations inside a batched engine."""

    def __init__(self, eng, off):
        self._eng, self._off = eng, off

    def read_probe(self, i, slot=0):
        return self._eng.read_probe(self._off + i, slot)


def _run_independent(sims: list[Simulation], seeds: list[int], device: int = 0,
                     stream_bases: list[int] | None = None) -> list[SimulationSummary]:
    """Run independent Simulations as one engine launch (each Simulation must lower to a single station).
    `stream_bases`: entity stream numbering per Simulation (default 0: every replica numbers its entities from 0)."""
    graphs = [s.lowered() for s in sims]
    one_station = [i for i, g in enumerate(graphs) if hasattr(g, "stations") and len(g.stations) == 1]
    if len(one_station) != len(sims):
        # Simulations that are more than one station -- what parallel/runner.py:82-142 hands to a worker process each.  General
        # graphs (graph_engine.GeneralGraph) run SIDE BY SIDE on the device, one workgroup and one heap each
        # (hs_graph_run_many); a station network or a load-balancer pipeline fills the device on its own and takes its turn.
        if stream_bases is not None:
            raise UnsupportedTopology("stream_bases number the entities of one-station replicas only")
        out: list = [None] * len(sims)
        for i, s in enumerate(sims):
            s._seed, s._device = int(seeds[i]), device
        if one_station:
            for i, summ in zip(one_station, _run_independent([sims[i] for i in one_station], [seeds[i] for i in one_station], device)):
                out[i] = summ
        general = [i for i, g in enumerate(graphs) if isinstance(g, GeneralGraph)]
        for i, summ in zip(general, _run_general_batch([sims[i] for i in general], [graphs[i] for i in general])):
            out[i] = summ
        for i, s in enumerate(sims):
            if out[i] is None:
                out[i] = s.run()
        return out
    ends = {s._end_time.nanoseconds for s in sims}
    starts = {s._start_time.nanoseconds for s in sims}
    if len(ends) != 1 or len(starts) != 1:
        raise UnsupportedTopology("batched replicas must share start_time and end_time")
    if Instant.Infinity.nanoseconds in ends:
        raise UnsupportedTopology("auto-terminating runs are not lowered; pass end_time/duration")
    end_ns, start_ns = ends.pop(), starts.pop()
    per_sim, cancelled = [], []
    for s, g in zip(sims, graphs):
        a = g.arrays()
        cancelled.append(s._schedule_arrays(g, a))         # Simulation.schedule() calls of this replica
        per_sim.append(a)
    st = _concat(per_sim)
    if st.src_more_kind is not None:     # several Sources per Server: every replica's `sources=[...]` order (its own prologue)
        order, slots = [], []
        for i, (s, g) in enumerate(zip(sims, graphs)):
            where = {id(g.stations[0].source): 0, **{id(x): 1 + k for k, x in enumerate(g.stations[0].more_sources)}}
            for src in s._sources:
                order.append(i)
                slots.append(where[id(src)])
        st.source_order, st.source_slot_order = np.asarray(order, np.int32), np.asarray(slots, np.uint8)
    st.seed = np.asarray(seeds, np.uint64)
    st.stream_base = (np.zeros(st.n, np.uint64) if stream_bases is None else np.asarray(stream_bases, np.uint64))
    wall0 = _time.monotonic()
    with StationEngine(st, mode=N.MODE_REPLICAS, horizon_ns=end_ns, start_ns=start_ns, device=device) as eng:
        eng.run_until(end_ns)
        stats = eng.lp_stats()
        more = [eng.source_generated(1 + k) for k in range(3)] if st.src_more_kind is not None else None
        counts, t_ns, created_ns = eng.read_sinks()
        for i, (s, g) in enumerate(zip(sims, graphs)):
            if s._probes:
                write_back_probes(g, _LpOffset(eng, i))
    wall = _time.monotonic() - wall0
    out = []
    off = 0
    for i, (s, g) in enumerate(zip(sims, graphs)):
        c = int(counts[i])
        sl = {k: v[i:i + 1] for k, v in stats.items()}
        if more is not None:
            sl["generated_more"] = [m[i:i + 1] for m in more]
        write_back(g, sl, counts[i:i + 1], t_ns[off:off + c], created_ns[off:off + c])
        off += c
        s._events_processed = int(stats["events"][i])
        s._current_time = Instant(int(stats["final_time_ns"][i]))
        final_ns = int(stats["final_time_ns"][i])            # a cancelled Event is counted when the loop pops it
        s._events_cancelled = sum(1 for t in cancelled[i] if final_ns <= end_ns or t <= final_ns)
        s._summary = s._build_summary(wall / len(sims))
        out.append(s._summary)
    return out


def _run_general_batch(sims: list[Simulation], graphs: list) -> list[SimulationSummary]:
    """Independent general graphs with one end each: engines created one by one, run to their ends in ONE batch per distinct end
    (hs_graph_run_many), results bound like Simulation._run_general binds them."""
    if not sims:
        return []
    wall0 = _time.monotonic()
    engines, plans = [], []
    try:
        for s, g in zip(sims, graphs):
            auto = s._end_time == Instant.Infinity
            if auto and (s._sources or s._probes):
                raise UnsupportedTopology("end_time = Infinity with Sources or Probes never terminates (their ticks are primary events, "
                                          "in the reference too); pass end_time/duration")
            end_ns, start_ns, sched, cancelled_ns = s._general_prepare(g, auto)
            s._refuse_long_run(1)
            # the record log of a replica starts near what its Sources can produce (it grows on demand): thousands of replicas
            # should not hold 1.3 MB of log each
            a = g.arrays
            horizon_s = 0.0 if auto else max(end_ns - start_ns, 0) / 1e9
            want = 2.0 * float(a.src_rate[a.kind == N.NODE_SOURCE].sum()) * horizon_s + 2.0 * len(sched) + 256.0
            n_probe = int((a.kind == N.NODE_PROBE).sum())
            if n_probe:
                want += float((horizon_s / a.probe_interval_s[a.kind == N.NODE_PROBE]).sum()) + 2.0 * n_probe
            engines.append(s._general_engine(g, start_ns, sched, record_capacity=int(min(max(want, 1024.0), 65536.0))))
            plans.append((end_ns, cancelled_ns))
        for end_ns in sorted({p[0] for p in plans}):
            GraphEngine.run_many([e for e, p in zip(engines, plans) if p[0] == end_ns], end_ns)
        wall = (_time.monotonic() - wall0) / len(sims)
        return [s._general_finish(g, e, p[0], p[1], wall) for s, g, e, p in zip(sims, graphs, engines, plans)]
    finally:
        for e in engines:
            e.close()


class ParallelRunner:
    def __init__(self, max_workers: int | None = None, device: int = 0):
        self._max_workers = max_workers        # accepted for API compatibility; lanes replace worker processes
        self._device = device

    def run_sweep(self, configs: list[RunConfig]) -> list[ParallelResult]:
        if not configs:
            return []
        sims = [cfg.build_fn() for cfg in configs]
        seeds = [cfg.seed if cfg.seed is not None else s._seed for cfg, s in zip(configs, sims)]
        summaries = _run_independent(sims, seeds, self._device)
        return [ParallelResult(name=c.name, summary=s) for c, s in zip(configs, summaries)]

    def run_replicas(self, build_fn: Callable, n_replicas: int, base_seed: int = 42) -> list[ParallelResult]:
        return self.run_sweep([RunConfig(name=f"replica_{i}", build_fn=build_fn, seed=base_seed + i)
                               for i in range(n_replicas)])


@dataclass
class SimulationPartition:
    name: str
    entities: list = field(default_factory=list)
    sources: list = field(default_factory=list)
    probes: list = field(default_factory=list)


@dataclass
class PartitionLink:
    source_partition: str
    dest_partition: str
    min_latency: float
    latency: Any = None
    packet_loss: float = 0.0

    def __post_init__(self):
        if self.min_latency <= 0:
            raise ValueError(f"PartitionLink min_latency must be > 0, got {self.min_latency}")  # parallel/link.py:41-45
        if not (0.0 <= self.packet_loss < 1.0):                                                  # parallel/link.py:46-49
            raise ValueError(f"PartitionLink packet_loss must be in [0, 1), got {self.packet_loss}")
        if self.source_partition == self.dest_partition:                                         # parallel/link.py:50-53
            raise ValueError(f"PartitionLink source and dest must differ, got '{self.source_partition}'")

    @staticmethod
    def bidirectional(partition_a: str, partition_b: str, min_latency: float, latency: Any = None,
                      packet_loss: float = 0.0) -> "tuple[PartitionLink, PartitionLink]":
        """A pair of links (A -> B and B -> A) with identical parameters (parallel/link.py:55-79)."""
        return (PartitionLink(partition_a, partition_b, min_latency, latency, packet_loss),
                PartitionLink(partition_b, partition_a, min_latency, latency, packet_loss))


@dataclass
class ParallelSimulationSummary:
    """happysimulator/parallel/summary.py:12-86 -- same fields, `to_dict()` keys and `__str__` layout.

    How the timing fields read on this engine (the formulas are the reference's, parallel/simulation.py:256-263):
    `partition_wall_times[name]` is the wall clock the partition spent executing on the device.  Independent partitions
    all advance inside ONE launch, so each one's time is that launch's (a partition run alone takes as long: the launch is
    bound by its slowest lane, not by the number of lanes); linked partitions that are virtual shards of one GPU take turns
    on it and split the execution time; with one process per GPU each process reports its own partition.  `speedup` is
    their sum over the whole run's wall clock (lowering, engine set-up and write-back included), `parallelism_efficiency`
    that per partition; `barrier_overhead_seconds` is the host time inside the exchange / GVT / inject calls of a linked
    run (coordinator.py:105-109), `coordination_efficiency` = 1 - overhead / wall clock."""

    duration_s: float
    total_events_processed: int
    events_per_second: float = 0.0
    wall_clock_seconds: float = 0.0
    partitions: dict = field(default_factory=dict)
    entities: dict = field(default_factory=dict)
    partition_wall_times: dict = field(default_factory=dict)
    speedup: float = 1.0
    parallelism_efficiency: float = 1.0
    total_windows: int = 0
    total_cross_partition_events: int = 0
    window_size_s: float = 0.0
    barrier_overhead_seconds: float = 0.0
    coordination_efficiency: float = 1.0
    engine_exchanges: int = 0       # (not a reference field, not in to_dict()) exchange rounds the engine's shards actually ran

    def to_dict(self) -> dict:
        return {"duration_s": self.duration_s, "total_events_processed": self.total_events_processed,
                "events_per_second": self.events_per_second, "wall_clock_seconds": self.wall_clock_seconds,
                "partitions": {k: v.to_dict() for k, v in self.partitions.items()},
                "entities": {k: v.to_dict() for k, v in self.entities.items()},
                "partition_wall_times": dict(self.partition_wall_times),
                "speedup": self.speedup, "parallelism_efficiency": self.parallelism_efficiency,
                "total_windows": self.total_windows,
                "total_cross_partition_events": self.total_cross_partition_events,
                "window_size_s": self.window_size_s,
                "barrier_overhead_seconds": self.barrier_overhead_seconds,
                "coordination_efficiency": self.coordination_efficiency}

    def __str__(self) -> str:
        lines = ["Parallel Simulation Summary",
                 f"  Duration: {self.duration_s:.2f}s (sim) / {self.wall_clock_seconds:.3f}s (wall)",
                 f"  Events processed: {self.total_events_processed}",
                 f"  Events/sec (sim): {self.events_per_second:.1f}",
                 f"  Partitions: {len(self.partitions)}",
                 f"  Speedup: {self.speedup:.2f}x",
                 f"  Efficiency: {self.parallelism_efficiency:.1%}"]
        if self.total_windows > 0:
            lines += [f"  Windows: {self.total_windows} (size={self.window_size_s:.4f}s)",
                      f"  Cross-partition events: {self.total_cross_partition_events}",
                      f"  Barrier overhead: {self.barrier_overhead_seconds:.3f}s",
                      f"  Coordination efficiency: {self.coordination_efficiency:.1%}"]
        return "\n".join(lines)


def _trim(a: np.ndarray) -> np.ndarray:
    """A loss table without its trailing zeros (a kept packet needs no bit)."""
    nz = np.nonzero(a)[0]
    return a[:int(nz[-1]) + 1] if len(nz) else a[:0]


def reference_window_count(start: Instant, end: Instant, window_s: float) -> int:
    """`total_windows` as the reference's coordinator counts them (parallel/coordinator.py:87-114): from the start, the window end is
    `Instant.from_seconds(current.to_seconds() + W)` clamped to the end, until the clock reaches the end -- binary64 accumulation
    included (12 s of 0.05 s windows are 241, not 240).  The engine's own exchange cadence follows the lookahead of the boundary
    stations instead (a few dozen rounds, `engine_exchanges`); the reference's early exit when every heap is empty does not arise
    while a Source ticks.  Beyond 5 million windows the count is the quotient without the walk's drift (approximate, reported only)."""
    # (the same arithmetic on plain ints / floats: Instant.to_seconds() is ns / 1e9, Instant.from_seconds(x) is int(x * 1e9) --
    #  one Instant per window made a 60 s run with 1 us windows spend tens of seconds here, for a number that is only reported)
    cur, end_ns, end_s = start.nanoseconds, end.nanoseconds, end.to_seconds()
    if (end_ns - cur) / 1e9 / window_s > 5_000_000:
        # beyond a few million windows the walk itself would dominate the run: the count without the binary64 drift of the walk
        # (the reference's own loop would take minutes of pure Python per partition there)
        # (approximate there: the reference's walk accumulates binary64 drift window by window).  ADVICE r5: a window too small to move
        # the nanosecond clock still raises as the walk would -- one step at the start and one just before the end.
        import math
        for c in (cur, end_ns - 1):
            w = c / 1e9 + window_s
            if not (c < int((w if w < end_s else end_s) * 1e9)):
                raise ValueError(f"window_size {window_s}s does not advance the clock at {c / 1e9}s")
        return math.ceil((end_ns - cur) / 1e9 / window_s)
    n = 0
    while cur < end_ns:
        w = cur / 1e9 + window_s
        nxt = int((w if w < end_s else end_s) * 1e9)
        if not (cur < nxt):          # a window too small to move the nanosecond clock: the reference would spin
            raise ValueError(f"window_size {window_s}s does not advance the clock at {cur / 1e9}s")
        cur, n = nxt, n + 1
    return n


def _parallel_summary(part_summaries: dict, part_wall: dict, wall: float, *, duration_s: float, total_events: int,
                      n_partitions: int, windows: int = 0, cross: int = 0, window_s: float = 0.0,
                      barrier_s: float = 0.0) -> ParallelSimulationSummary:
    """ParallelSimulation._build_summary (parallel/simulation.py:225-284)."""
    entities = {}
    for ps in part_summaries.values():
        entities.update(ps.entities)
    speedup = sum(part_wall.values()) / wall if wall > 0 else 1.0
    return ParallelSimulationSummary(
        duration_s=duration_s, total_events_processed=total_events,
        events_per_second=total_events / duration_s if duration_s > 0 else 0.0, wall_clock_seconds=wall,
        partitions=part_summaries, entities=entities, partition_wall_times=part_wall, speedup=speedup,
        parallelism_efficiency=speedup / n_partitions if n_partitions > 0 else 1.0, total_windows=windows,
        total_cross_partition_events=cross, window_size_s=window_s, barrier_overhead_seconds=barrier_s,
        coordination_efficiency=1.0 - barrier_s / wall if wall > 0 else 1.0)


class ParallelSimulation:
    """`ParallelSimulation(partitions, links=..., seed=...)` (happysimulator/parallel/simulation.py:31-284).

    * no links: every partition is an independent Simulation; all of them run as ONE engine launch.
    * links: the partitions form one network whose stations are connected by `NetworkLink`s.  Each partition becomes
      one shard of the windowed network engine (happy_simulator_amd/sharded.py): a GPU per partition when
      torch.distributed is initialised with world_size == len(partitions) (RCCL exchange + GVT all-reduce), otherwise
      virtual shards on one GPU.  The lookahead is the smallest `NetworkLink` base latency, which must be at least the
      `PartitionLink.min_latency` declared for the partition pair it crosses.  Unlike the reference's coordinator --
      which overshoots each window by one event and then drops late cross-partition events as "time travel"
       -- the result is exactly the single-heap `Simulation.run()` of the same entities."""

    def __init__(self, partitions: list[SimulationPartition], *, start_time: Instant | None = None,
                 end_time: Instant | None = None, duration: float | None = None, max_workers: int | None = None,
                 links: list[PartitionLink] | None = None, window_size: float | None = None, seed: int = 42,
                 device: int = 0, log_capacity: int | None = None, bag_capacity: int | None = None,
                 msg_capacity: int | None = None):
        if duration is not None and end_time is not None:
            raise ValueError("Cannot specify both 'duration' and 'end_time'")
        self._links = list(links or [])
        self._window_size = window_size
        self._validate(partitions, self._links, window_size)
        self._partitions = partitions
        self._seed = seed
        self._device = device
        self._log_capacity, self._bag_capacity, self._msg_capacity = log_capacity, bag_capacity, msg_capacity
        self._start = start_time if start_time is not None else Instant.Epoch
        if duration is not None:
            self._end = self._start + duration
        elif end_time is not None:
            self._end = end_time
        else:
            self._end = Instant.Infinity
        if not self._links:
            self._sims = [Simulation(start_time=start_time, end_time=end_time, duration=duration, sources=p.sources,
                                     entities=p.entities, probes=p.probes, seed=seed) for p in partitions]
        else:
            self._lower_linked()

    # parallel/validation.py:19-110
    @staticmethod
    def _validate(partitions, links, window_size):
        names = [p.name for p in partitions]
        seen = set()
        for nm in names:
            if nm in seen:
                raise ValueError(f"Duplicate partition name: '{nm}'")
            seen.add(nm)
        owner: dict[int, str] = {}
        for p in partitions:
            for e in p.entities:
                if id(e) in owner:
                    raise ValueError(f"Entity '{getattr(e, 'name', e)}' is in partitions '{owner[id(e)]}' and '{p.name}'")
                owner[id(e)] = p.name
        for p in partitions:
            for src in p.sources:
                tgt = getattr(getattr(src, "_event_provider", None), "_target", None)
                if tgt is not None and owner.get(id(tgt), p.name) != p.name:
                    raise ValueError(f"Source '{src.name}' in partition '{p.name}' targets entity '{tgt.name}' in "
                                     f"partition '{owner[id(tgt)]}'")
        for lk in links:
            if lk.source_partition not in seen:
                raise ValueError(f"PartitionLink references unknown source partition '{lk.source_partition}'")
            if lk.dest_partition not in seen:
                raise ValueError(f"PartitionLink references unknown dest partition '{lk.dest_partition}'")
            if lk.latency is not None:
                # parallel/coordinator.py:207-210: the coordinator calls `link.latency.sample()`, a method none of the reference's
                # LatencyDistributions has (AttributeError there: tests/test_oracle_live_reference.py) -- no library behaviour to mirror.
                raise UnsupportedTopology("PartitionLink(latency=...) is not lowered: the reference's coordinator calls `.sample()` on it, "
                                          "which its own LatencyDistributions do not have; put the latency on the hop "
                                          "(NetworkLink(latency=..., jitter=...))")
        # PartitionLink(packet_loss=p) (parallel/coordinator.py:68,203-205): ONE `random.Random(seed)` per coordinator, drawn once per
        # cross-partition event of a lossy link, in exchange order = (window, source partition in dict order, outbox order).  While
        # every lossy PartitionLink leaves the SAME partition that order is the partition's own processing order of the sending
        # events (pinned on the live class: tests/golden parallel_linked_loss*), which the host replays over the run's sends
        # (`_replay_partition_losses`).  With several lossy source partitions the order depends on which window the reference's
        # coordinator processed each sending event in -- its one-event overshoot per window included (core/simulation.py:472) -- a
        # property of its window walk over EVERY event of the partition, not of the model: refused by name.
        lossy_from = {lk.source_partition for lk in links if lk.packet_loss > 0.0}
        if len(lossy_from) > 1:
            raise UnsupportedTopology("PartitionLink(packet_loss=...) on links out of several partitions (" + ", ".join(sorted(lossy_from)) +
                                      "): the reference's one loss stream then interleaves by its windows' event-by-event overshoot, "
                                      "which is not lowered; keep the lossy links to one source partition or put the loss on the hops "
                                      "(NetworkLink(packet_loss_rate=...))")
        if window_size is not None and links:
            m = min(lk.min_latency for lk in links)
            if window_size > m:
                raise ValueError(f"window_size ({window_size}s) must be <= min(link.min_latency) ({m}s)")

    def _lower_linked(self):
        from .lowering import lower

        if self._end == Instant.Infinity:
            raise UnsupportedTopology("auto-terminating runs are not lowered; pass end_time/duration")
        parts = self._partitions
        owner = {id(e): k for k, p in enumerate(parts) for e in list(p.entities) + list(p.sources)}
        g = lower([s for p in parts for s in p.sources], [e for p in parts for e in p.entities])
        part_of = []
        for st in g.stations:
            ref = st.server if st.server is not None else (st.sink if st.sink is not None else st.source)
            if id(ref) not in owner:
                raise UnsupportedTopology(f"'{ref.name}' is not listed in any partition's entities")
            part_of.append(owner[id(ref)])
        if any(a > b for a, b in zip(part_of, part_of[1:])):
            raise UnsupportedTopology("list every Server in its partition's `entities` so that partitions are contiguous")
        sizes = np.bincount(np.asarray(part_of, np.int64), minlength=len(parts))
        if (sizes == 0).any():
            raise UnsupportedTopology("a partition without a Server / Source station cannot be a shard")
        declared = {(lk.source_partition, lk.dest_partition): lk.min_latency for lk in self._links}
        loss_of = {(lk.source_partition, lk.dest_partition): lk.packet_loss for lk in self._links}
        self._cross_links = []
        self._lossy_links: dict[int, float] = {}          # graph link -> PartitionLink.packet_loss of the pair it crosses
        for l, (lk, s, d) in enumerate(g.links):
            ps, pd = part_of[s], part_of[d]
            if ps == pd:
                continue
            key = (parts[ps].name, parts[pd].name)
            if key not in declared:                                    # parallel/validation.py:188-200
                raise ValueError(f"Entity in partition '{key[0]}' references entity '{lk.egress.name}' in partition "
                                 f"'{key[1]}' but no PartitionLink exists from '{key[0]}' to '{key[1]}'")
            if lk.latency.mean < declared[key]:                        # parallel/coordinator.py:213-219 (RuntimeError there)
                raise ValueError(f"link '{lk.name}' can deliver after {lk.latency.mean}s, less than the PartitionLink "
                                 f"min_latency {declared[key]}s")
            self._cross_links.append(l)
            p_loss = loss_of.get(key, 0.0)
            if p_loss > 0.0:
                if lk.packet_loss_rate > 0.0:
                    raise UnsupportedTopology(f"link '{lk.name}' has a packet_loss_rate of its own and crosses a PartitionLink with "
                                              "packet_loss: two loss decisions per packet are not lowered")
                self._lossy_links[l] = p_loss
        if any(p.probes for p in parts):
            from .lowering import attach_probes

            attach_probes(g, [pr for p in parts for pr in p.probes])
        # a collector behind stations of SEVERAL partitions (the reference allows it when it is not registered in any
        # partition): its records are merged from all of them after the run -- which needs them in one process
        seen: dict[int, int] = {}
        self._sink_spans = False
        for i, st in enumerate(g.stations):
            if st.sink is not None and seen.setdefault(id(st.sink), part_of[i]) != part_of[i]:
                self._sink_spans = True
        self._graph = g
        self._bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    def run(self) -> ParallelSimulationSummary:
        if self._links:
            return self._run_linked()
        wall0 = _time.monotonic()
        # every partition is its own Simulation (own heap, own overshoot); all share the run's Philox key and
        # number their entities globally (partition i = stream base i), so identical partitions still draw
        # independent streams
        n = len(self._sims)
        summaries = _run_independent(self._sims, [self._seed] * n, self._device, stream_bases=list(range(n)))
        dur = max(s.duration_s for s in summaries)
        tot = sum(s.total_events_processed for s in summaries)
        launch = sum(s.wall_clock_seconds for s in summaries)      # _run_independent books launch / n on every partition
        return _parallel_summary({p.name: s for p, s in zip(self._partitions, summaries)},
                                 {p.name: launch for p in self._partitions}, _time.monotonic() - wall0,
                                 duration_s=dur, total_events=tot, n_partitions=n)

    def _replay_partition_losses(self, sn, comm, summ, end_ns):
        """`PartitionLink.packet_loss` by host replay (parallel/coordinator.py:68,203-205).  The k-th cross-partition event of a lossy
        link, in the sending partition's processing order, is lost iff the k-th `random.Random(seed).random()` is below the link's
        `packet_loss`.  Which send is the k-th depends on the run, so: run, read every shard's send log (hs_engine_read_send_log),
        order it by send time, draw, hand the decisions back as one bit per packet of each link (hs_engine_set_link_drops), run
        again -- until a run's sends reproduce the decisions it ran with.  A pipeline settles after the second run (nothing upstream
        of the lossy link depends on what it loses); a cycle back into the sending partition settles decision by decision."""
        import random

        p_of = self._lossy_links
        used: dict[int, np.ndarray] = {l: np.zeros(0, bool) for l in p_of}
        for _attempt in range(64):
            logs = comm.gather_rows([s.engine.send_log() for s in sn.shards])          # [k, 3]: send ns, link, packet number
            order = np.lexsort((logs[:, 2], logs[:, 1], logs[:, 0]))
            logs = logs[order]
            # two sends of DIFFERENT links on one nanosecond: their order is the reference's sort-index order of two events of
            # different stations, which the send log does not carry (lock-step constants only) -- refused by name
            same_ns = (logs[1:, 0] == logs[:-1, 0]) & (logs[1:, 1] != logs[:-1, 1])
            if same_ns.any():
                raise UnsupportedTopology("two hops of a lossy PartitionLink sent on the same nanosecond: the order of their loss draws "
                                          "is the reference's sort-index order, which is not lowered (lock-step constant services)")
            rng = random.Random(self._seed)                                # the coordinator's generator (ParallelSimulation(seed=))
            want = {l: np.zeros(int((logs[:, 1] == l).sum()), bool) for l in p_of}
            for _t, l, e in logs.tolist():
                want[l][e] = rng.random() < p_of[l]
            # (trailing survivors need no bits: packets beyond the table's written part are kept)
            if all(np.array_equal(_trim(want[l]), _trim(used[l])) for l in p_of):
                return summ
            for s in sn.shards:
                for l in p_of:
                    if s.lo <= self._graph.links[l][1] < s.hi:
                        s.engine.set_link_drops(int(np.nonzero(s.gids == l)[0][0]), want[l])
            used = want
            summ = sn.run_until(end_ns)
        raise N.EngineError(N.HS_E_UNSUPPORTED, "PartitionLink.packet_loss: the loss decisions did not settle in 64 runs")

    def _run_linked(self) -> ParallelSimulationSummary:
        import torch.distributed as dist

        from .sharded import DistComm, LocalComm, ShardedNetwork
        from .summary import SimulationSummary as _SS

        wall0 = _time.monotonic()
        g, parts = self._graph, self._partitions
        world = len(parts)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() == world and world > 1:
            comm = DistComm()                     # one partition per process / GPU
            if self._sink_spans:
                raise UnsupportedTopology("a Sink / Counter fed from several partitions cannot be merged across processes yet: "
                                          "give every partition its own collector (or run the partitions on one GPU)")
        else:
            comm = LocalComm(world)               # virtual shards on this GPU
        end_ns, start_ns = self._end.nanoseconds, self._start.nanoseconds
        st, net = g.arrays(), g.network_arrays(self._bag_capacity or 0)
        cap = self._log_capacity or g.log_capacity((end_ns - start_ns) / 1e9)
        if self._lossy_links:            # a loss table per lossy cross link, as long as a record log (no station forwards more)
            net.link_drop_capacity = np.zeros(net.n_links, np.int64)
            net.link_drop_capacity[list(self._lossy_links)] = cap
        with ShardedNetwork.on_gpu(st, net, comm, horizon_ns=end_ns, start_ns=start_ns, seed=self._seed,
                                   device=self._device, log_capacity=cap, bounds=self._bounds,
                                   **({"msg_capacity": self._msg_capacity} if self._msg_capacity else {})) as sn:
            summ = sn.run_until(end_ns)
            if self._lossy_links:
                summ = self._replay_partition_losses(sn, comm, summ, end_ns)
            part_summaries = {}
            cross_local = 0
            # ONE write-back over all of this process's shards: a collector fed by stations of several shards gets its
            # records merged in completion order (hs_merge_sink_records) instead of being overwritten shard by shard
            full, fc, t_all, cr_all, fnet = sn.collect(st.n, net.n_links)
            local = sorted(sn.shards, key=lambda x: x.lo)
            if len(local) == world:
                write_back(g, full, fc, t_all, cr_all, None, device=self._device)
            else:                                  # one shard per process: its own stations only
                write_back(g, full, fc, t_all, cr_all, None, lo=local[0].lo, hi=local[0].hi, device=self._device)
            if any(p.probes for p in parts):
                write_back_probes_sharded(g, sn)
            for s in sn.shards:
                # packets_sent is counted where the link ENDS, losses where it STARTS: each end's shard owns its counters
                for l, (lk, src, dst) in enumerate(g.links):
                    if s.lo <= dst < s.hi:
                        lk.packets_sent = int(fnet["link_packets_sent"][l])
                    if s.lo <= src < s.hi:
                        lk._entered = int(fnet["link_entered"][l])
                        # (what a lossy PartitionLink took is the coordinator's doing, not the hop's: NetworkLink.packets_dropped stays 0)
                        lk.packets_dropped = 0 if l in self._lossy_links else int(fnet["link_packets_dropped"][l])
                for i in range(s.lo, s.hi):
                    if g.stations[i].router is not None:
                        g.stations[i].router.stats_routed = int(fnet["routed"][i])
                cross_local += int(sum(fnet["link_entered"][l] - fnet["link_packets_dropped"][l] for l in self._cross_links
                                       if s.lo <= g.links[l][1] < s.hi))
                tot = s.totals()
                dur = (tot["max_final_ns"] - start_ns) / 1e9
                part_summaries[parts[s.rank].name] = _SS(
                    duration_s=dur, total_events_processed=tot["events"],
                    events_per_second=tot["events"] / dur if dur > 0 else 0.0)
            for s in sn.shards:              # every shard has written back: the objects hold the results now
                part_summaries[parts[s.rank].name].entities = entity_summaries(parts[s.rank].entities)
            cross = comm.reduce_host([{"cross": cross_local}])["cross"]
        dur = (summ.final_time_ns - start_ns) / 1e9
        # this process's shards take turns on its GPU: they split the time run_until spent outside the exchanges
        busy = max(summ.run_seconds - summ.exchange_seconds, 0.0) / max(len(part_summaries), 1)
        # total_windows / window_size_s read as the reference's (parallel/simulation.py:82-87, coordinator.py:87-114): W = the
        # caller's window_size or the smallest PartitionLink.min_latency; the engine's exchange rounds are reported beside them
        window_s = self._window_size if self._window_size is not None else min(lk.min_latency for lk in self._links)
        out = _parallel_summary(part_summaries, {name: busy for name in part_summaries}, _time.monotonic() - wall0,
                                duration_s=dur, total_events=summ.events_processed, n_partitions=world,
                                windows=reference_window_count(self._start, self._end, window_s), cross=int(cross),
                                window_s=window_s, barrier_s=summ.exchange_seconds)
        out.engine_exchanges = int(summ.windows)
        return out
