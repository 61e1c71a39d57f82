"""`Simulation` -- host-side mirror of happysimulator/core/simulation.py:66-288 whose `run()` executes on the
MI355X engine instead of the Python heap loop.  Same constructor, same `SimulationSummary`."""
from __future__ import annotations

import time as _time

import warnings

import numpy as np

from . import _native as N
from .core.event import Event
from .core.temporal import Instant
from .engine import StationEngine
from .graph_engine import (DEFAULT_MAX_EVENTS, MAX_PARTS, GeneralGraph, GraphEngine, PartRun, keyless_hazard, lower_general, split_parts,
                           write_back_general)
from .entities import Entity, Server
from .lowering import (LazyRecords, LbGraph, LoweredGraph, UnsupportedTopology, attach_lb_probes, attach_probes, find_load_balancer, lower,
                       plain_probe_arrays, write_back_plain_probes,
                       lower_lb, write_back, write_back_lb, write_back_plain, write_back_probes, write_back_shared_sink_probes)
from .lowering import _plain_chains as plain_chains
from .summary import EntitySummary, LazyEntities, QueueStats, SimulationSummary

_DEFAULT_SEED = 42


LAZY_PROBES_MIN = 256      # plain-chain runs with more Probes than this leave the samples on the device until a Data is read


def seed(value: int) -> None:
    """Default Philox key for Simulations created afterwards (the reference's idiom is `random.seed(42)`)."""
    global _DEFAULT_SEED
    _DEFAULT_SEED = int(value)


class Simulation:
    def __init__(self, start_time: Instant | None = None, end_time: Instant | None = None, sources=None,
                 entities=None, probes=None, trace_recorder=None, fault_schedule=None, duration: float | None = None,
                 *, seed: int | None = None, device: int = 0, log_capacity: int | None = None,
                 bag_capacity: int | None = None, msg_capacity: int | None = None, max_graph_events: int | None = None):
        if duration is not None and end_time is not None:
            raise ValueError("Cannot specify both 'duration' and 'end_time'")        # core/simulation.py:79-80
        self._start_time = start_time if start_time is not None else Instant.Epoch
        if duration is not None:
            self._end_time = self._start_time + duration
        elif end_time is not None:
            self._end_time = end_time
        else:
            self._end_time = Instant.Infinity
        self._sources = list(sources or [])
        self._entities = list(entities or [])
        self._probes = list(probes or [])
        if trace_recorder is not None:
            raise UnsupportedTopology("trace recorders force the reference's slow loop; profile with rocprofv3 instead")
        if fault_schedule is not None:
            raise UnsupportedTopology("fault schedules are not lowered")
        self._seed = _DEFAULT_SEED if seed is None else int(seed)
        self._device = device
        # engine capacities (records per station log / in-flight messages per station / messages per exchange row); None =
        # derived from the rates.  An HS_E_OVERFLOW names the one to raise.
        self._log_capacity, self._bag_capacity, self._msg_capacity = log_capacity, bag_capacity, msg_capacity
        # a graph outside the station shape runs on the single-heap loop (graph_engine.py): events it may cost before it is refused
        self._max_graph_events = DEFAULT_MAX_EVENTS if max_graph_events is None else int(max_graph_events)
        self._summary: SimulationSummary | None = None
        self._graph: LoweredGraph | None = None
        self._events_processed = 0
        self._events_cancelled = 0
        self._current_time = self._start_time
        self._engine_summary = None
        self._scheduled: list[Event] = []

    def schedule(self, events) -> None:
        """Inject Events from outside the event loop, before run() (core/simulation.py:195-206).  Lowered: Requests for a
        Server of this Simulation; they enter the Server's queue at their time like a Source's payload would, with
        context["created_at"] = their own time.  An Event built before run() precedes run-time events of the same ns."""
        if self._summary is not None:
            raise UnsupportedTopology("schedule() after run(): the engine run is over (build a new Simulation)")
        for ev in (events if isinstance(events, list) else [events]):
            if not isinstance(ev, Event):
                raise TypeError(f"schedule() takes Event objects, got {type(ev).__name__}")
            self._scheduled.append(ev)

    def _schedule_arrays(self, g: LoweredGraph, arrays) -> list[int]:
        """Scheduled Events -> per-station ascending time lists (stable in call order).  Returns the times of the
        cancelled ones."""
        if not self._scheduled:
            return []
        station_of = {id(st.server): i for i, st in enumerate(g.stations) if st.server is not None}
        per: list[list[tuple[int, int]]] = [[] for _ in g.stations]
        cancelled: list[int] = []
        start_ns = self._start_time.nanoseconds
        for rank, ev in enumerate(self._scheduled):     # rank: the Event's position among everything the caller constructed
            if ev.cancelled:                       # lazy deletion: skipped when popped, counted (simulation.py:475-477)
                cancelled.append(ev.time.nanoseconds)
                continue
            i = station_of.get(id(ev.target))
            if i is None:
                raise UnsupportedTopology(
                    f"scheduled event {ev!r}: only Requests for a Server of this Simulation are lowered")
            if ev.on_complete:
                raise UnsupportedTopology(f"scheduled event {ev!r}: completion hooks are host Python (not lowered)")
            if ev.context.get("created_at") != ev.time:
                raise UnsupportedTopology(f"scheduled event {ev!r}: a custom created_at is not lowered")
            if ev.time.nanoseconds < start_ns:     # "time travel": the loop skips it without counting (simulation.py:480-489)
                warnings.warn(f"Time travel detected: {ev!r} lies before the simulation start; skipping event", stacklevel=3)
                continue
            per[i].append((ev.time.nanoseconds, rank))
        # Same-nanosecond order against run-time events is the reference's: the engine replays the first constructions of
        # the run in exact heap order (csrc/hs_exact.hpp) from the construction ranks handed over here.
        off = np.zeros(len(g.stations) + 1, np.int64)
        off[1:] = np.cumsum([len(p) for p in per])
        flat = [tr for p in per for tr in sorted(p)]                         # ascending time, ties in construction order
        arrays.sched_off = off
        arrays.sched_time_ns = np.array([t for t, _ in flat], np.int64)
        arrays.sched_rank = np.array([r for _, r in flat], np.int64)
        return cancelled

    @property
    def summary(self) -> SimulationSummary | None:
        return self._summary

    def lowered(self) -> "LoweredGraph | LbGraph | GeneralGraph":
        if self._graph is None:
            plain = plain_chains(self._sources, self._entities)      # (n plain chains: no LoadBalancer among them, one pass less)
            if plain is not None:
                self._graph = LoweredGraph(plain)
                lb = None
            else:
                lb = None
                try:
                    lb = find_load_balancer(self._sources, self._entities)
                    self._graph = lower_lb(self._sources, self._entities, lb) if lb is not None else lower(self._sources,
                                                                                                            self._entities)
                except UnsupportedTopology as station_shape:
                    # not the shape the station engines take: the single-heap loop (csrc/hs_graph.hip) runs what the same entity
                    # classes can be wired into otherwise -- exactly, at ~2.4 us per event and heap (several LoadBalancers, a LoadBalancer
                    # behind Servers ... included)
                    try:
                        self._graph = lower_general(self._sources, self._entities, self._probes)
                    except UnsupportedTopology as general:
                        raise UnsupportedTopology(f"{station_shape}; and not on the single-heap path either: {general}") from None
                    self._station_refusal = str(station_shape)
                    return self._graph
            if self._probes:
                try:
                    if lb is not None:
                        attach_lb_probes(self._graph, self._probes)
                    elif plain is not None and not self._scheduled:
                        self._plain_probes_pending = True     # (_run puts them into the arrays without a Station per chain)
                    else:
                        attach_probes(self._graph, self._probes)
                except UnsupportedTopology as station_shape:
                    # (more than four probes on a station, a further Source of a Server sampled, a probe on the LoadBalancer's
                    # Source with stop_after ...: the single heap samples anything)
                    try:
                        self._graph = lower_general(self._sources, self._entities, self._probes)
                    except UnsupportedTopology:
                        raise station_shape from None
                    self._station_refusal = str(station_shape)
        return self._graph

    def _run_lb(self, g: LbGraph, wall0: float) -> SimulationSummary:
        """Sources -> LoadBalancer(ConsistentHash) -> Servers -> Sink(s): the pipeline engine (csrc/hs_lb.hip)."""
        from .lb_engine import LoadBalancerEngine

        end_ns = self._end_time.nanoseconds
        src, be = g.engine_arrays()
        from .entities import ConsistentHash, RoundRobin

        strat = g.lb.strategy
        code = N.LB_CONSISTENT_HASH if isinstance(strat, ConsistentHash) else N.LB_ROUND_ROBIN if isinstance(strat, RoundRobin) else N.LB_RANDOM
        with LoadBalancerEngine(src, be, virtual_nodes=getattr(strat, "virtual_nodes", 1), horizon_ns=end_ns,
                                shared_sink=g.shared_sink, start_ns=self._start_time.nanoseconds, seed=self._seed,
                                device=self._device, strategy=code) as eng:
            if g.probes:
                eng.set_probes(*g.probe_arrays())
            eng.run(end_ns)
            es = eng.summary()
            write_back_lb(g, eng.stats(), eng)
        self._engine_summary = es
        self._events_processed = es.events_processed
        self._current_time = Instant(es.final_time_ns)
        self._summary = self._build_summary(_time.monotonic() - wall0)
        return self._summary

    def __del__(self):
        try:                       # results this run has not bound to its entities yet (entities._PENDING) outlive the Simulation
            from .entities import _flush_pending
            _flush_pending()
        except Exception:          # noqa: BLE001 -- interpreter shutdown
            pass

    def run(self) -> SimulationSummary:
        auto = self._end_time == Instant.Infinity
        if auto:
            # Auto-termination (core/simulation.py:311-322): the loop ends when no PRIMARY event is pending
            # (core/event_heap.py:102-104).  A Source's ticks are primary and never stop -- with any Source the reference
            # itself never returns -- so the lowered case is a Simulation driven by schedule()d Requests only: it ends with
            # the last completion, nothing beyond it is processed.  (Probe ticks are daemon events: where the run stops
            # relative to them is decided pop by pop -- not lowered.)
            if self._sources:
                raise UnsupportedTopology("end_time = Infinity with Sources never terminates (their ticks are primary events, in "
                                          "the reference too); pass end_time/duration")
            if self._probes:
                # a Probe IS a Source (instrumentation/probe.py:81): its ticks are SourceEvents built with daemon=False
                # (load/source.py:136,171; load/source_event.py:27) -- only the probe_event samples are daemons -- so they keep
                # the primary count above zero for ever, exactly like a Source's
                raise UnsupportedTopology("end_time = Infinity with Probes never terminates either (a Probe's ticks are primary events: "
                                          "only its probe_event samples are daemons, load/source_event.py:27); pass end_time/duration")
        wall0 = _time.monotonic()
        import gc

        gc_was = gc.isenabled()
        gc.disable()        # lowering and write-back touch every entity once: generational collections over 10^5 live objects cost more than the run
        try:
            return self._run(auto, wall0)
        finally:
            if gc_was:
                gc.enable()

    def _run(self, auto: bool, wall0: float) -> SimulationSummary:
        g = self.lowered()
        if not isinstance(g, GeneralGraph):
            try:
                return self._run_station(g, auto, wall0)
            except UnsupportedTopology as station_shape:
                # refused on the way (a probe the station slots do not hold, a tick on the nanosecond of a shared Sink's record ...):
                # the single-heap loop takes it if the graph is its kind; otherwise the station engines' refusal stands
                try:
                    g = lower_general(self._sources, self._entities, self._probes)
                except UnsupportedTopology:
                    raise station_shape from None
                self._station_refusal = str(station_shape)
                self._graph = g
        try:
            return self._run_general(g, auto, wall0)
        except UnsupportedTopology as general:
            raise UnsupportedTopology(f"{self._station_refusal}; and not on the single-heap path either: {general}") from None

    def _run_station(self, g, auto: bool, wall0: float) -> SimulationSummary:
        if isinstance(g, LbGraph):
            if self._scheduled:
                raise UnsupportedTopology("schedule() is not lowered for load-balancer topologies yet")
            return self._run_lb(g, wall0)
        if auto and g.is_network:
            # the network engines run to a horizon; the single-heap loop ends with its heap, as the reference's does
            # (core/simulation.py:304-322: no Sources here, so every pending event is primary and the run drains)
            raise UnsupportedTopology("auto-terminating runs of station networks are not lowered on the network engines")
        # auto-termination: a horizon nothing reaches (the station kernel stops when no event is pending)
        end_ns = self._end_time.nanoseconds if not auto else (1 << 61)
        net = g.network_arrays(self._bag_capacity or 0) if g.is_network else None
        horizon_s = (end_ns - self._start_time.nanoseconds) / 1e9
        arrays = g.arrays()
        if getattr(self, "_plain_probes_pending", False):
            self._plain_probes_pending = False
            fast = not self._scheduled and g.plain is not None and g._stations is None      # (`arrays` is PlainChains.arrays)
            where = plain_probe_arrays(g.plain, self._probes, arrays) if fast else None
            if where is not None:
                return self._run_plain(g, arrays, end_ns, wall0, where)
            attach_probes(g, self._probes)                # (a probe the fast path does not cover: Station objects after all)
            arrays = g.arrays()
        elif g.plain is not None and not self._probes and not self._scheduled:
            return self._run_plain(g, arrays, end_ns, wall0)
        # the order in which Simulation.__init__ constructs the first SourceEvents / probe ticks (core/simulation.py:145-160)
        st_of = {id(st.source): (i, 0) for i, st in enumerate(g.stations) if st.source is not None}
        st_of.update({id(x): (i, 1 + k) for i, st in enumerate(g.stations) for k, x in enumerate(st.more_sources)})
        arrays.source_order = np.array([st_of[id(s)][0] for s in self._sources], np.int32)
        if any(st.more_sources for st in g.stations):
            arrays.source_slot_order = np.array([st_of[id(s)][1] for s in self._sources], np.uint8)
        if self._probes:
            where = {id(pr): (i, slot) for i, st in enumerate(g.stations) for slot, pr in enumerate(st.probes)}
            arrays.probe_order = np.array([where[id(p)][0] for p in self._probes], np.int32)
            arrays.probe_slot_order = np.array([where[id(p)][1] for p in self._probes], np.uint8)
        cancelled_ns = self._schedule_arrays(g, arrays)
        if net is not None and arrays.n > self._resident_stations():
            return self._run_time_shared(g, arrays, net, end_ns, horizon_s, wall0, cancelled_ns)
        with StationEngine(arrays, mode=N.MODE_SINGLE, horizon_ns=end_ns, start_ns=self._start_time.nanoseconds,
                           seed=self._seed, device=self._device, network=net,
                           log_capacity=self._log_cap(g, horizon_s, arrays, net is not None)) as eng:
            eng.run_until(end_ns)
            es = eng.summary()
            stats = eng.lp_stats()
            if arrays.src_more_kind is not None:
                stats["generated_more"] = [eng.source_generated(1 + k) for k in range(3)]
            counts, t_ns, created_ns = eng.read_sinks()
            net_stats = eng.net_stats() if net is not None else None
            if self._probes:
                write_back_probes(g, eng)
        write_back(g, stats, counts, t_ns, created_ns, net_stats, device=self._device)
        write_back_shared_sink_probes(g)
        # a cancelled event is counted when the loop pops it: everything up to the last processed event, or the whole
        # heap when the run ended with nothing left beyond end_time (core/simulation.py:472-477)
        drained = es.final_time_ns <= end_ns
        self._events_cancelled = sum(1 for t in cancelled_ns if drained or t <= es.final_time_ns)
        self._engine_summary = es
        self._events_processed = es.events_processed
        self._current_time = Instant(es.final_time_ns)
        self._summary = self._build_summary(_time.monotonic() - wall0)
        return self._summary

    def _run_general(self, g: GeneralGraph, auto: bool, wall0: float) -> SimulationSummary:
        """A graph outside the station shape (graph_engine.lower_general) on the device's single-heap loop."""
        end_ns, start_ns, sched, cancelled_ns = self._general_prepare(g, auto)
        # (one component per heap while there are at most MAX_PARTS: components that SHARE a heap are checked as one -- two Probes of one
        #  interval, two constant Sources of one rate in different components then look like a mixed timestamp group: undecided)
        parts = None if cancelled_ns else split_parts(g.arrays, MAX_PARTS)
        try:
            self._refuse_long_run(1 if parts is None else len(parts))
        except UnsupportedTopology:
            if parts is None:
                raise
            parts = None                                           # (uneven parts: the one heap's own estimate decides)
            self._refuse_long_run(1)
        if parts is not None:
            # the graph falls into parts no Request can cross: one heap each, side by side (hs_graph_run_parts) -- exact unless a
            # timestamp group turns out to be ordered by ALL of the Simulation's events; then the one heap below decides
            part_of = np.empty(g.arrays.n, np.int64)
            local = np.empty(g.arrays.n, np.int64)
            for p, (ids, _pos, _b) in enumerate(parts):
                part_of[ids] = p
                local[ids] = np.arange(len(ids))
            engines = []
            try:
                for p, (ids, _pos, b) in enumerate(parts):
                    engines.append(GraphEngine(b, seed=self._seed, start_ns=start_ns, device=self._device, max_events=self._max_graph_events,
                                               record_capacity=4096))
                for node, t in sched:                              # (call order: each part keeps the order of its own)
                    engines[int(part_of[node])].schedule(int(local[node]), t)
                run = PartRun(g.arrays, parts, engines)
                if run.run(end_ns):
                    self._graph_parts = len(parts)
                    return self._general_finish(g, run, end_ns, cancelled_ns, _time.monotonic() - wall0)
            finally:
                for e in engines:
                    e.close()
        self._graph_parts = 1
        self._refuse_long_run(1)
        with self._general_engine(g, start_ns, sched) as eng:
            eng.run_until(end_ns)
            return self._general_finish(g, eng, end_ns, cancelled_ns, _time.monotonic() - wall0)

    def _refuse_long_run(self, heaps: int = 1) -> None:
        """(after _general_prepare) refuse up front what would keep one lane busy for minutes; `heaps`: the heaps the run is spread over."""
        est = self._general_est / max(heaps, 1)
        if self._max_graph_events > 0 and est > 4.0 * self._max_graph_events:
            raise UnsupportedTopology(
                f"the single-heap path (one lane, ~2 us per event) would need ~{est:.2g} events per heap for this run "
                f"(limit {self._max_graph_events}: Simulation(max_graph_events=...))")

    def _general_prepare(self, g: GeneralGraph, auto: bool):
        """What the single-heap engine is created with: (end ns, start ns, the schedule()d Requests as (node, ns), the cancelled
        Events' times) -- or the refusal of a run it does not take."""
        end_ns = self._end_time.nanoseconds if not auto else (1 << 61)
        start_ns = self._start_time.nanoseconds
        a = g.arrays
        # ~8 reference events per Request that is served and ~2 per hop: refuse up front what would take the one lane minutes
        horizon_s = 0.0 if auto else (end_ns - start_ns) / 1e9
        self._general_est = 12.0 * float(a.src_rate[a.kind == N.NODE_SOURCE].sum()) * horizon_s
        cancelled_ns: list[int] = []
        sched: list[tuple[int, int]] = []
        for ev in self._scheduled:
            if ev.cancelled:                       # lazy deletion: skipped when popped, counted (simulation.py:475-477)
                cancelled_ns.append(ev.time.nanoseconds)
                continue
            i = g.node_of.get(id(ev.target))
            if i is None or a.kind[i] == N.NODE_SOURCE:
                raise UnsupportedTopology(f"scheduled event {ev!r}: only Requests for a Server / Sink / NetworkLink / RandomRouter of "
                                          "this Simulation are lowered")
            if ev.on_complete:
                raise UnsupportedTopology(f"scheduled event {ev!r}: completion hooks are host Python (not lowered)")
            if ev.context.get("created_at") != ev.time:
                raise UnsupportedTopology(f"scheduled event {ev!r}: a custom created_at is not lowered")
            lb_name = keyless_hazard(g, ev.target)
            if lb_name is not None:
                raise UnsupportedTopology(f"scheduled event {ev!r} carries no client id and can reach the Random LoadBalancer "
                                          f"'{lb_name}' (the reference would ask the process-wide random generator): not lowered")
            if cancelled_ns:
                # a cancelled Event keeps its place in the process-wide counter: the ones behind it would need the gap
                raise UnsupportedTopology("cancelled Events in front of live ones are not lowered on the single-heap path")
            if ev.time.nanoseconds < start_ns:     # "time travel": the loop skips it without counting (simulation.py:480-489)
                warnings.warn(f"Time travel detected: {ev!r} lies before the simulation start; skipping event", stacklevel=3)
                continue
            sched.append((i, ev.time.nanoseconds))
        return end_ns, start_ns, sched, cancelled_ns

    def _general_engine(self, g: GeneralGraph, start_ns: int, sched, record_capacity: int = 0) -> GraphEngine:
        eng = GraphEngine(g.arrays, seed=self._seed, start_ns=start_ns, device=self._device, max_events=self._max_graph_events,
                          record_capacity=record_capacity)
        try:
            for node, t in sched:
                eng.schedule(node, t)
        except BaseException:
            eng.close()
            raise
        return eng

    def _general_finish(self, g: GeneralGraph, eng: GraphEngine, end_ns: int, cancelled_ns, wall_s: float) -> SimulationSummary:
        """The run's results off the engine onto the user's objects (the engine may have run alone or in a batch)."""
        es = eng.summary()
        stats = eng.stats()
        rec = eng.records()
        write_back_general(g, stats, *rec, device=self._device)
        drained = es.final_time_ns <= end_ns
        self._events_cancelled = sum(1 for t in cancelled_ns if drained or t <= es.final_time_ns)
        self._engine_summary = es
        self._events_processed = es.events_processed
        self._current_time = Instant(es.final_time_ns)
        self._summary = self._build_summary(wall_s)
        return self._summary

    def _run_plain(self, g: LoweredGraph, arrays, end_ns: int, wall0: float, probe_where=None) -> SimulationSummary:
        """n plain Source -> Server -> [Sink] chains (lowering.PlainChains): no per-station Python objects on the way in, Python
        lists instead of numpy scalars on the way out, and the Sink records stay on the device until a Sink's lists are first read
        (LazyRecords keeps the engine until then) -- at 65 536 chains run() used to spend 0.7 s around a 0.45 ms device run."""
        arrays.source_order = g.plain.source_station
        eng = StationEngine(arrays, mode=N.MODE_SINGLE, horizon_ns=end_ns, start_ns=self._start_time.nanoseconds, seed=self._seed,
                            device=self._device, log_capacity=int(self._log_capacity or 0))
        try:
            eng.run_until(end_ns)
            es = eng.summary()
            stats = eng.lp_stats()
        except Exception:
            eng.close()
            raise
        # a few probes: their samples come down now (microseconds each) and nothing pins the engine beyond the Sink records; many
        # (a Probe on every one of 65 536 Servers): they stay on the device until a Data is read, and every unread Data keeps the
        # engine alive through the LazyRecords it holds (a Data read after the Simulation was dropped found a closed engine)
        lazy_probes = bool(probe_where) and len(probe_where) > LAZY_PROBES_MIN
        records = LazyRecords(eng, stats["sink_received"], keep_engine=lazy_probes)
        self._records = records                 # (keeps the device buffers alive as long as the Simulation, or until fetched)
        write_back_plain(g.plain, stats, records, device=self._device)
        if probe_where:
            write_back_plain_probes(self._probes, probe_where, records, lazy=lazy_probes)
        self._events_cancelled = 0
        self._engine_summary = es
        self._events_processed = es.events_processed
        self._current_time = Instant(es.final_time_ns)
        self._summary = self._build_summary(_time.monotonic() - wall0)
        return self._summary

    def _log_cap(self, g, horizon_s: float, arrays, is_net: bool) -> int:
        if self._log_capacity:
            return int(self._log_capacity)
        if not is_net:
            return 0                           # the engine derives it from the rates (and the scheduled Requests)
        extra = int(np.diff(arrays.sched_off).max()) if arrays.sched_off is not None and len(arrays.sched_off) > 1 else 0
        return g.log_capacity(horizon_s, extra)

    def _resident_stations(self) -> int:
        """Stations one cooperative launch of the asynchronous network engine holds: one 256-lane workgroup per CU (its
        LDS rings and bags fill the CU)."""
        import torch

        return torch.cuda.get_device_properties(self._device).multi_processor_count * 256

    def _run_time_shared(self, g, arrays, net, end_ns: int, horizon_s: float, wall0: float, cancelled_ns=()) -> SimulationSummary:
        """A network with more stations than one cooperative launch holds: contiguous segments take turns on the device
        under the asynchronous-rounds protocol of the multi-GPU path (happy_simulator_amd/sharded.py) -- the same bits as
        one engine, a few dozen rounds instead of tens of thousands of windows."""
        from .sharded import LocalComm, ShardedNetwork

        world = -(-arrays.n // self._resident_stations())
        with ShardedNetwork.on_gpu(arrays, net, LocalComm(world), horizon_ns=end_ns, start_ns=self._start_time.nanoseconds,
                                   seed=self._seed, device=self._device, log_capacity=self._log_cap(g, horizon_s, arrays, True),
                                   **({"msg_capacity": self._msg_capacity} if self._msg_capacity else {})) as sn:
            es = sn.run_until(end_ns)
            stats, counts, t_ns, created_ns, net_stats = sn.collect(arrays.n, net.n_links)
            if self._probes:
                write_back_probes(g, sn)
        write_back(g, stats, counts, t_ns, created_ns, net_stats, device=self._device)
        write_back_shared_sink_probes(g)
        drained = es.final_time_ns <= end_ns
        self._events_cancelled = sum(1 for t in cancelled_ns if drained or t <= es.final_time_ns)
        self._engine_summary = es
        self._events_processed = es.events_processed
        self._current_time = Instant(es.final_time_ns)
        self._summary = self._build_summary(_time.monotonic() - wall0)
        return self._summary

    # core/simulation.py:543-591
    def _build_summary(self, wall_elapsed: float) -> SimulationSummary:
        duration_s = (self._current_time - self._start_time).to_seconds()
        eps = self._events_processed / duration_s if duration_s > 0 else 0.0
        return SimulationSummary(duration_s=duration_s, total_events_processed=self._events_processed,
                                 events_cancelled=self._events_cancelled, events_per_second=eps,
                                 wall_clock_seconds=wall_elapsed,
                                 # (eager like the reference for ordinary sizes; beyond 4 096 entities the loop is deferred until
                                 #  somebody reads the dict -- it would cost more than the device run)
                                 entities=(entity_summaries(self._entities) if len(self._entities) <= 4096 else
                                           LazyEntities(lambda ents=self._entities: entity_summaries(ents))))


def entity_summaries(entities) -> dict[str, EntitySummary]:
    """Simulation._build_entity_summaries (core/simulation.py:560-591): `events_handled` is the first int among the
    attributes count / events_received / stats_processed; QueuedResources report accepted / dropped (peak depth is never
    tracked by the reference, :572)."""
    out: dict[str, EntitySummary] = {}
    for comp in entities:
        if not isinstance(comp, Entity):
            continue
        queue_stats = None
        if isinstance(comp, Server):
            queue_stats = QueueStats(peak_depth=0, total_accepted=comp.stats_accepted, total_dropped=comp.stats_dropped)
        handled = 0
        for attr in ("count", "events_received", "stats_processed"):
            val = getattr(comp, attr, None)
            if isinstance(val, int):
                handled = val
                break
        out[comp.name] = EntitySummary(name=comp.name, entity_type=type(comp).__name__, events_handled=handled,
                                       queue_stats=queue_stats)
    return out
