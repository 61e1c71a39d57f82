"""Mirror of happysimulator/parallel for the hot path, re-designed for one GPU per process.

* `ParallelRunner.run_replicas / run_sweep` (parallel/runner.py:82-142): the reference forks one process per
  replica and seeds `random` with `base_seed + i`.  Here all replicas are lowered into ONE engine launch --
  one LP per lane, `HS_MODE_REPLICAS`, per-LP Philox key `base_seed + i` -- so 4 096 replicas cost one kernel.
* `ParallelSimulation` without links (parallel/simulation.py:170-195): every partition is an independent
  Simulation; same batching.  Linked partitions (windows + GVT) are the next scope row.
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
from .lowering import UnsupportedTopology, write_back
from .simulation import Simulation
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
    keys = ("src_kind", "src_rate", "src_stop_after_ns", "concurrency", "svc_kind", "svc_mean_s", "queue_cap", "egress")
    n = sum(a.n for a in arrs)
    return StationArrays(n=n, **{k: np.concatenate([getattr(a, k) for a in arrs]) for k in keys})


def _run_independent(sims: list[Simulation], seeds: list[int], device: int = 0,
                     stream_bases: list[int] | None = None) -> list[SimulationSummary]:
    """Run independent Simulations as one engine launch (each Simulation must lower to a single station).
    `stream_bases`: entity stream numbering per Simulation (default 0: every replica numbers its entities from 0)."""
    graphs = [s.lowered() for s in sims]
    for s, g in zip(sims, graphs):
        if len(g.stations) != 1:
            raise UnsupportedTopology(
                "batched replicas need one station per Simulation; run multi-station Simulations with .run()")
    ends = {s._end_time.nanoseconds for s in sims}
    starts = {s._start_time.nanoseconds for s in sims}
    if len(ends) != 1 or len(starts) != 1:
        raise UnsupportedTopology("batched replicas must share start_time and end_time")
    if Instant.Infinity.nanoseconds in ends:
        raise UnsupportedTopology("auto-terminating runs are not lowered; pass end_time/duration")
    end_ns, start_ns = ends.pop(), starts.pop()
    st = _concat([g.arrays() for g in graphs])
    st.seed = np.asarray(seeds, np.uint64)
    st.stream_base = (np.zeros(st.n, np.uint64) if stream_bases is None else np.asarray(stream_bases, np.uint64))
    wall0 = _time.monotonic()
    with StationEngine(st, mode=N.MODE_REPLICAS, horizon_ns=end_ns, start_ns=start_ns, device=device) as eng:
        eng.run_until(end_ns)
        stats = eng.lp_stats()
        counts, t_ns, created_ns = eng.read_sinks()
    wall = _time.monotonic() - wall0
    out = []
    off = 0
    for i, (s, g) in enumerate(zip(sims, graphs)):
        c = int(counts[i])
        sl = {k: v[i:i + 1] for k, v in stats.items()}
        write_back(g, sl, counts[i:i + 1], t_ns[off:off + c], created_ns[off:off + c])
        off += c
        s._events_processed = int(stats["events"][i])
        s._current_time = Instant(int(stats["final_time_ns"][i]))
        s._summary = s._build_summary(wall / len(sims))
        out.append(s._summary)
    return out


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


@dataclass
class ParallelSimulationSummary:
    duration_s: float
    total_events_processed: int
    partitions: dict
    wall_clock_seconds: float
    total_windows: int = 0
    total_cross_partition_events: int = 0


class ParallelSimulation:
    """Independent partitions only (no links): each partition runs as its own Simulation, all in one launch."""

    def __init__(self, partitions: list[SimulationPartition], *, start_time: Instant | None = None,
                 end_time: Instant | None = None, duration: float | None = None, max_workers: int | None = None,
                 links: list[PartitionLink] | None = None, window_size: float | None = None, seed: int = 42,
                 device: int = 0):
        if duration is not None and end_time is not None:
            raise ValueError("Cannot specify both 'duration' and 'end_time'")
        if links:
            raise UnsupportedTopology("linked partitions (windowed coordination) are the next scope row")
        names = [p.name for p in partitions]
        if len(set(names)) != len(names):
            raise ValueError("partition names must be unique")
        self._partitions = partitions
        self._sims = [Simulation(start_time=start_time, end_time=end_time, duration=duration, sources=p.sources,
                                 entities=p.entities, seed=seed) for p in partitions]
        self._seed = seed
        self._device = device

    def run(self) -> ParallelSimulationSummary:
        wall0 = _time.monotonic()
        # every partition is its own Simulation (own heap, own overshoot); all share the run's Philox key and
        # number their entities globally (partition i = stream base i), so identical partitions still draw
        # independent streams
        n = len(self._sims)
        summaries = _run_independent(self._sims, [self._seed] * n, self._device, stream_bases=list(range(n)))
        return ParallelSimulationSummary(
            duration_s=max(s.duration_s for s in summaries),
            total_events_processed=sum(s.total_events_processed for s in summaries),
            partitions={p.name: s for p, s in zip(self._partitions, summaries)},
            wall_clock_seconds=_time.monotonic() - wall0)
