"""Graph lowering: the user's Source / Server / Sink / NetworkLink / RandomRouter objects -> station LPs
(struct-of-arrays) and, when stations are connected, the link table of the windowed network engine.

Walks `sources` and `entities` the way the reference's topology discovery does
(happysimulator/visual/topology.py:81-146: follow `downstream_entities()`), and refuses -- explicitly,
never silently -- anything the engine does not run.

A station LP = [Source] -> Server -> egress, where the egress is one of
    nothing | a Sink-like collector | NetworkLink -> another station's Server
            | RandomRouter([Sink-like | NetworkLink, Sink-like | NetworkLink])
Every entity of station i draws from Philox streams with stream base i (DESIGN.md "Random streams").
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .engine import NetworkArrays, StationArrays
from .entities import (ClientKeyEventProvider, ConsistentHash, ConstantLatency, ConstantRateProfile, Counter, Entity,
                       ExponentialLatency, LatencyTracker, LinearRampProfile, LoadBalancer, NetworkLink, Probe, Random, RandomRouter,
                       RoundRobin, Server, SimpleEventProvider, Sink, Source, _RecordSink)

_SINKS = (Sink, Counter, LatencyTracker)


class UnsupportedTopology(NotImplementedError):
    """The entity graph contains something the GPU engine does not lower (stay on the reference's CPU loop)."""


class Station:
    """One station LP.  (A slotted class with shared empty tuples as defaults: lowering 65 536 chains builds 65 536 of these, and
    a dataclass with four list factories was a third of that time.)"""
    __slots__ = ("probes", "source", "more_sources", "server", "sink", "router", "links", "link_ids", "next_server", "next_station")

    def __init__(self, source=None, server=None, sink=None):
        self.probes = ()            # the Probes sampling this station's entities (up to 4: engine slots)
        self.source = source
        self.more_sources = ()      # further Sources feeding the same Server (up to 3: engine slots 1..3)
        self.server = server
        self.sink = sink
        self.router = None
        self.links = ()             # NetworkLink objects leaving this station (router order)
        self.link_ids = ()          # their indices in LoweredGraph.links
        self.next_server = None     # tandem queues: Server(downstream=<Server>) -- that Server, and its station
        self.next_station = -1


class PlainChains:
    """The commonest graph -- n independent `Source -> Server -> [Sink]` chains, every Source a constant-rate one with a Server of
    its own (BASELINE configs[1], the headline grid) -- lowered WITHOUT one Python object per station: the per-object walk of
    lower() and write_back() was 3 200 x the device run at 65 536 chains.  Holds the entities in station order
    and the engine's struct-of-arrays, both built with list comprehensions over the homogeneous lists."""

    def __init__(self, sources, servers, sinks, source_station, arrays):
        self.sources = sources                  # station order
        self.servers = servers
        self.sinks = sinks                      # the Server's collector or None
        self.source_station = source_station    # int32[len(sources)]: the station of every Source in `sources=[...]` order
        self.arrays = arrays

    def stations(self) -> list:
        out = []
        for src, sv, sk in zip(self.sources, self.servers, self.sinks):
            st = Station(source=src, server=sv, sink=sk)
            out.append(st)
        return out


def _plain_chains(sources, entities):
    """PlainChains when `sources` / `entities` are exactly that shape, else None (lower() then walks the graph).  Every check is
    a C-level pass (map / attrgetter / set) over the homogeneous lists, and every object is visited ONCE per pass that needs it
    (one multi-attribute getter per class: at 65 536 chains the passes are bound by the cache misses of walking 200 000 Python
    objects, ~0.2 us per object and visit): ~0.1 s for 65 536 chains."""
    from operator import attrgetter as ag, is_

    n = len(sources)
    if n < 64 or set(map(type, sources)) != {Source}:
        return None
    cols = list(zip(*map(ag("_event_provider._target", "_event_provider._stop_after", "_time_provider.kind", "_time_provider.profile"),
                         sources)))
    servers, stops, kinds, profiles = (list(c) for c in cols)
    if set(map(type, servers)) != {Server}:
        return None
    ent_types = set(map(type, entities))
    if not ent_types <= {Server, Sink, Counter, LatencyTracker}:
        return None
    ent_servers = [e for e in entities if type(e) is Server]
    if len(ent_servers) != n:
        return None
    # station order = the Servers' order in `entities` (the entity stream numbering, see lower()); the usual case: the same order
    in_order = all(map(is_, ent_servers, servers))
    order = None
    if in_order:
        station_of_source = np.arange(n, dtype=np.int32)
    else:
        server_ids = set(map(id, servers))
        if len(server_ids) != n or set(map(id, ent_servers)) != server_ids:
            return None
        pos = dict(zip(map(id, ent_servers), range(n)))
        station_of_source = np.fromiter(map(pos.__getitem__, map(id, servers)), np.int32, n)
        order = np.argsort(station_of_source, kind="stable")
    downs, svcs, concs, caps = (list(c) for c in zip(*map(ag("_downstream", "_service_time", "_concurrency", "_policy._capacity"), servers)))
    if not set(map(type, downs)) <= {Sink, Counter, LatencyTracker, type(None)}:
        return None
    real = [d for d in downs if d is not None]
    if len(set(map(id, real))) != len(real):            # a collector shared by several Servers: the general path merges its records
        return None
    if in_order and len(set(map(id, servers))) != n:    # (one Server behind several Sources: the general path)
        return None
    svc_types = set(map(type, svcs))
    if not svc_types <= {ExponentialLatency, ConstantLatency}:
        return None
    conc = np.fromiter(concs, np.int32, n)
    if set(map(type, profiles)) != {ConstantRateProfile}:
        return None
    rate = np.fromiter(map(ag("rate"), profiles), np.float64, n)
    if conc.max() > 32 or not (rate > 0).all():
        return None

    def perm(lst):
        return lst if in_order else [lst[k] for k in order]

    a = StationArrays.uniform(n)
    kind_set = set(kinds)
    kind = (np.full(n, N.SRC_POISSON if kind_set == {"poisson"} else N.SRC_CONSTANT, np.uint8) if len(kind_set) == 1 else
            np.fromiter((N.SRC_POISSON if k == "poisson" else N.SRC_CONSTANT for k in kinds), np.uint8, n))
    stop = (np.full(n, -1, np.int64) if set(stops) == {None} else
            np.fromiter((-1 if x is None else x.nanoseconds for x in stops), np.int64, n))
    svk = (np.full(n, N.LAT_EXPONENTIAL if svc_types == {ExponentialLatency} else N.LAT_CONSTANT, np.uint8) if len(svc_types) == 1 else
           np.fromiter((N.LAT_EXPONENTIAL if type(x) is ExponentialLatency else N.LAT_CONSTANT for x in svcs), np.uint8, n))
    svm = np.fromiter(map(ag("_mean_latency"), svcs), np.float64, n)
    inf = float("inf")
    qcap = np.full(n, -1, np.int64) if set(caps) == {inf} else np.fromiter((-1 if c == inf else int(c) for c in caps), np.int64, n)
    egr = (np.full(n, N.EGRESS_SINK, np.uint8) if len(real) == n else
           np.fromiter((N.EGRESS_NONE if d is None else N.EGRESS_SINK for d in downs), np.uint8, n))
    take = (lambda x: x) if in_order else (lambda x: x[order])
    a.src_kind[:] = take(kind); a.src_rate[:] = take(rate); a.src_stop_after_ns[:] = take(stop)
    a.concurrency[:] = take(conc); a.svc_kind[:] = take(svk); a.svc_mean_s[:] = take(svm); a.queue_cap[:] = take(qcap)
    a.egress[:] = take(egr)
    return PlainChains(perm(list(sources)), perm(servers), perm(downs), station_of_source, a)


class LoweredGraph:
    """stations: one Station per LP; links: (NetworkLink, source station, destination station).  `plain`: the graph is n plain
    chains (PlainChains) -- the Station objects are then only built if somebody asks for them (probes, schedule(), partitions)."""

    def __init__(self, plain: "PlainChains | None" = None):
        self._stations: list[Station] | None = None if plain is not None else []
        self.links: list = []
        self.plain = plain

    @property
    def stations(self) -> list[Station]:
        if self._stations is None:
            self._stations = self.plain.stations()
        return self._stations

    @property
    def is_network(self) -> bool:
        if self.plain is not None and self._stations is None:
            return False
        return bool(self.links) or any(st.router is not None for st in self.stations)

    def arrays(self) -> StationArrays:
        if self.plain is not None and self._stations is None:
            return self.plain.arrays
        n = len(self.stations)
        a = StationArrays.uniform(n)
        for i, st in enumerate(self.stations):
            if st.source is not None:
                prov = st.source._time_provider
                a.src_kind[i] = N.SRC_POISSON if prov.kind == "poisson" else N.SRC_CONSTANT
                a.src_rate[i] = float(prov.profile.peak_rate)
                pr = prov.profile
                if not isinstance(pr, ConstantRateProfile):
                    if a.src_profile_kind is None:
                        a.src_profile_kind = np.zeros(n, np.uint8)
                        a.src_profile_params = np.zeros((n, 4), np.float64)
                    if isinstance(pr, LinearRampProfile):
                        a.src_profile_kind[i] = N.PROF_LINEAR_RAMP
                        a.src_profile_params[i, :3] = (pr.duration_s, pr.start_rate, pr.end_rate)
                    else:
                        a.src_profile_kind[i] = N.PROF_SPIKE
                        a.src_profile_params[i] = (pr.baseline_rate, pr.spike_rate, pr.warmup_s, pr.spike_duration_s)
                stop = st.source._event_provider._stop_after
                a.src_stop_after_ns[i] = -1 if stop is None else stop.nanoseconds
            else:
                a.src_kind[i] = N.SRC_NONE
                a.src_rate[i] = 1.0
            if st.server is not None:
                sv = st.server
                a.concurrency[i] = sv.concurrency
                svc = sv.service_time
                a.svc_kind[i] = N.LAT_EXPONENTIAL if isinstance(svc, ExponentialLatency) else N.LAT_CONSTANT
                a.svc_mean_s[i] = svc.mean
                cap = sv._policy.capacity
                a.queue_cap[i] = -1 if cap == float("inf") else int(cap)
            else:
                a.svc_kind[i] = N.LAT_NO_SERVER
                a.svc_mean_s[i] = 0.0
            a.egress[i] = N.EGRESS_SINK if st.sink is not None else N.EGRESS_NONE
            if st.next_server is not None:                        # tandem queues (include/hs_engine.h downstream_lp)
                if a.downstream_lp is None:
                    a.downstream_lp = np.full(n, -1, np.int32)
                a.egress[i] = N.EGRESS_SERVER
                a.downstream_lp[i] = st.next_station
            for slot, src in enumerate(st.more_sources):
                if a.src_more_kind is None:
                    a.src_more_kind = np.full((3, n), N.SRC_NONE, np.uint8)
                    a.src_more_rate = np.ones((3, n), np.float64)
                    a.src_more_stop_after_ns = np.full((3, n), -1, np.int64)
                prov = src._time_provider
                a.src_more_kind[slot, i] = N.SRC_POISSON if prov.kind == "poisson" else N.SRC_CONSTANT
                a.src_more_rate[slot, i] = float(prov.profile.peak_rate)
                stop = src._event_provider._stop_after
                a.src_more_stop_after_ns[slot, i] = -1 if stop is None else stop.nanoseconds
            for slot, pr in enumerate(st.probes):
                if a.probe_metric is None:
                    a.probe_metric = np.full(n, N.PROBE_NONE, np.uint8)
                    a.probe_interval_s = np.ones(n, np.float64)
                if slot > 0 and a.probe_metric_more is None:
                    a.probe_metric_more = np.full((3, n), N.PROBE_NONE, np.uint8)
                    a.probe_interval_more = np.ones((3, n), np.float64)
                m = pr.metric
                code = N.PROBE_METRICS[Probe.engine_metric(m)]
                if slot == 0:
                    a.probe_metric[i], a.probe_interval_s[i] = code, pr.interval
                else:
                    a.probe_metric_more[slot - 1, i], a.probe_interval_more[slot - 1, i] = code, pr.interval
        return a

    def network_arrays(self, bag_capacity: int = 0) -> NetworkArrays:
        n, nl = len(self.stations), len(self.links)
        eg = np.zeros(n, np.uint8)
        rt = np.full((4, n), -1, np.int32)
        rtk = np.full(n, 2, np.uint8)
        lof = np.full(n, -1, np.int32)
        for i, st in enumerate(self.stations):
            if st.router is not None:
                eg[i] = N.EGRESS_ROUTER
                ids = iter(st.link_ids)
                tg = [(-1 if isinstance(t, _SINKS) else next(ids)) for t in st.router.targets]
                rtk[i] = len(tg)
                rt[:len(tg), i] = tg
            elif st.links:
                eg[i] = N.EGRESS_LINK
                lof[i] = st.link_ids[0]
            else:
                eg[i] = N.EGRESS_SINK if st.sink is not None else N.EGRESS_NONE
        jk = np.full(nl, N.LAT_CONSTANT, np.uint8)
        jm = np.zeros(nl, np.float64)
        for l, (lk, _, _) in enumerate(self.links):
            if lk.jitter is not None:      # ExponentialLatency: one draw per packet; ConstantLatency: a constant on top (link.py:195-200)
                jk[l] = N.LAT_EXPONENTIAL if isinstance(lk.jitter, ExponentialLatency) else N.LAT_CONSTANT
                jm[l] = lk.jitter.mean
        return NetworkArrays(
            egress_kind=eg, router_target0=rt[0], router_target1=rt[1], link_of=lof,
            router_n_targets=rtk if (rtk != 2).any() else None,
            router_target2=rt[2] if (rtk > 2).any() else None, router_target3=rt[3] if (rtk > 3).any() else None,
            link_src=np.array([s for _, s, _ in self.links], np.int32).reshape(nl),
            link_dst=np.array([d for _, _, d in self.links], np.int32).reshape(nl),
            link_lat_min_s=np.array([lk.latency.mean for lk, _, _ in self.links], np.float64).reshape(nl),
            link_jitter_kind=jk, link_jitter_mean_s=jm,
            link_loss_rate=(np.array([lk.packet_loss_rate for lk, _, _ in self.links], np.float64).reshape(nl)
                            if any(lk.packet_loss_rate for lk, _, _ in self.links) else None),
            bag_capacity=bag_capacity)

    def log_capacity(self, horizon_s: float, extra: int = 0) -> int:
        """Records per station.  A station admits what its own Source generates plus what its upstream links deliver, and a
        sender cannot forward more than it serves: solve  inflow[d] = rate[d] + sum over links s -> d of
        share * min(inflow[s], concurrency[s] / mean_service[s])  to its fixed point (share = 1 behind a plain link, 1 / number
        of targets behind a RandomRouter) -- also on deep tandems and cycles, where a fixed number of hops undercounts
        (round-1 advisor finding: a 10-station tandem overflowed its logs).  `extra`: injected Requests per station."""
        n = len(self.stations)
        rate = np.array([(st.source.rate if st.source is not None else 0.0) + sum(x.rate for x in st.more_sources)
                         for st in self.stations], np.float64)
        mu = np.array([(st.server.concurrency / st.server.service_time.mean) if (st.server is not None and
                       st.server.service_time.mean > 0) else np.inf for st in self.stations], np.float64)
        share = np.array([1.0 / max(len(self.stations[s].router.targets), 1) if self.stations[s].router is not None else 1.0
                          for _, s, _ in self.links], np.float64)
        src = np.array([s_ for _, s_, _ in self.links], np.int64)
        dst = np.array([d_ for _, _, d_ in self.links], np.int64)
        inflow = rate.copy()
        for _ in range(4 * n + 16):                       # (one vectorised pass per iteration: a deep tandem needs ~n of them)
            out = np.minimum(inflow, mu)
            nxt = rate.copy()
            if len(src):
                np.add.at(nxt, dst, share * out[src])
            if np.allclose(nxt, inflow, rtol=1e-9, atol=1e-12):
                inflow = nxt
                break
            inflow = nxt
        lam = float(inflow.max()) if n else 0.0
        mean = lam * horizon_s
        return int(mean + 10.0 * (mean + 1.0) ** 0.5 + 64) + int(extra)


def attach_probes(g: LoweredGraph, probes: list) -> None:
    """Probe(target, metric, interval) -> the station that owns `target` (one probe per station)."""
    from .entities import Probe

    owner = {}
    shared = set()
    for i, st in enumerate(g.stations):
        for obj in (st.source, st.server, st.sink):
            if obj is not None:
                if id(obj) in owner:
                    shared.add(id(obj))
                owner[id(obj)] = i
    for pr in probes or []:
        if not isinstance(pr, Probe):
            raise UnsupportedTopology(f"probe {type(pr).__name__} is not a lowered Probe")
        i = owner.get(id(pr.target))
        if i is None:
            if any(pr.target is x for st in g.stations for x in st.more_sources):
                raise UnsupportedTopology(f"probe '{pr.name}': only the first Source of a Server is sampled on the engine")
            raise UnsupportedTopology(f"probe '{pr.name}': its target is not an entity of this Simulation")
        if id(pr.target) in shared:
            # A Sink behind several Servers (`events_received` is the sum over its stations): the probe ticks on the FIRST station
            # that feeds the Sink -- its tick chain, its two events per tick and its place in the election of the event beyond
            # end_time are those of any probe -- and the values are read off the Sink's merged records after the run
            # (write_back_shared_sink_probes).  A tick on the very nanosecond of one of the Sink's records is refused there.
            i = min(k for k, st_k in enumerate(g.stations) if st_k.sink is pr.target)
            g.shared_sink_probes = (*getattr(g, "shared_sink_probes", ()), pr)
        st = g.stations[i]
        if len(st.probes) >= 4:
            raise UnsupportedTopology(f"station of '{pr.target.name}' already has four probes (the engine's slots per station)")
        if Probe.engine_metric(pr.metric) not in N.PROBE_METRICS:
            raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not sampled on the engine "
                                      f"(lowered: {', '.join(sorted(Probe._LOWERED))})")
        kinds = {"generated_count": Source, "_generated_count": Source, "events_received": _SINKS}
        want = kinds.get(pr.metric, Server)
        if not isinstance(pr.target, want):
            raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not an attribute of {type(pr.target).__name__}")
        st.probes = (*st.probes, pr)


def plain_probe_arrays(pc: "PlainChains", probes: list, arrays):
    """attach_probes() for PlainChains without a Station object per chain: fills the engine's probe arrays (slot 0 and the three
    further slots, in the order the probes are listed per station) and returns [(station, slot, scale)] per probe in `probes=[...]`
    order, or None when a probe needs the general lowering (a Sink shared by several Servers, a target outside the chains)."""
    from .entities import Probe

    n = len(pc.servers)
    owner = dict(zip(map(id, pc.servers), range(n)))
    owner.update(zip(map(id, pc.sources), range(n)))
    sink_ids = [id(sk) for sk in pc.sinks if sk is not None]
    if len(set(sink_ids)) != len(sink_ids):
        return None                                   # (a collector behind several Servers: write_back_shared_sink_probes)
    owner.update((id(sk), i) for i, sk in enumerate(pc.sinks) if sk is not None)
    metric = np.full((4, n), N.PROBE_NONE, np.uint8)
    interval = np.ones((4, n))
    used = np.zeros(n, np.int8)
    kinds = {"generated_count": Source, "_generated_count": Source, "events_received": _SINKS}
    where = []
    for pr in probes:
        if not isinstance(pr, Probe):
            raise UnsupportedTopology(f"probe {type(pr).__name__} is not a lowered Probe")
        i = owner.get(id(pr.target))
        if i is None:
            return None                               # (attach_probes words the refusal)
        slot = int(used[i])
        if slot >= 4:
            raise UnsupportedTopology(f"station of '{pr.target.name}' already has four probes (the engine's slots per station)")
        if Probe.engine_metric(pr.metric) not in N.PROBE_METRICS:
            raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not sampled on the engine "
                                      f"(lowered: {', '.join(sorted(Probe._LOWERED))})")
        if not isinstance(pr.target, kinds.get(pr.metric, Server)):
            raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not an attribute of {type(pr.target).__name__}")
        used[i] = slot + 1
        metric[slot, i] = N.PROBE_METRICS[Probe.engine_metric(pr.metric)]
        interval[slot, i] = pr.interval
        where.append((i, slot, Probe.value_map(pr.metric, pc.servers[i])))
    arrays.probe_metric, arrays.probe_interval_s = metric[0].copy(), interval[0].copy()
    if (used > 1).any():
        arrays.probe_metric_more, arrays.probe_interval_more = metric[1:].copy(), interval[1:].copy()
    arrays.probe_order = np.array([w[0] for w in where], np.int32)
    arrays.probe_slot_order = np.array([w[1] for w in where], np.uint8)
    return where


def write_back_plain_probes(probes: list, where: list, records: "LazyRecords", lazy: bool = True) -> None:
    """lazy: the samples stay on the device until a probe's Data is first read (65 536 read-backs of a few hundred bytes each are
    seconds of host time); every such Data holds `records` -- the owner of the engine -- so the engine lives as long as any unread
    Data does, whatever happens to the Simulation.  Not lazy: the samples are read now."""
    for pr, (i, slot, scale) in zip(probes, where):
        if lazy:
            pr.data_sink._set_lazy(lambda i=i, slot=slot, records=records: records.engine().read_probe(i, slot), scale)
        else:
            t, v = records.engine().read_probe(i, slot)
            pr.data_sink._set(t, v, scale)


def write_back_probes(g: LoweredGraph, eng) -> None:
    for i, st in enumerate(g.stations):
        for slot, pr in enumerate(st.probes):
            t, v = eng.read_probe(i, slot)
            pr.data_sink._set(t, v, Probe.value_map(pr.metric, st.server))


def write_back_shared_sink_probes(g: LoweredGraph) -> None:
    """After write_back(): a probe on a Sink that several stations feed (attach_probes) samples `events_received` = the number of
    the Sink's merged records before the tick.  When a tick shares its nanosecond with one of those records the reference's heap
    order decides whether the record counts -- not reconstructed here: refused by name, never guessed."""
    for pr in getattr(g, "shared_sink_probes", ()):
        t = np.asarray(pr.data_sink._t_ns, np.int64)
        rec = np.asarray(pr.target.completion_ns, np.int64)          # in processing order = ascending
        before, upto = np.searchsorted(rec, t, side="left"), np.searchsorted(rec, t, side="right")
        if (before != upto).any():
            k = int(np.flatnonzero(before != upto)[0])
            raise UnsupportedTopology(f"probe '{pr.name}': its tick at {int(t[k])} ns falls on the nanosecond of a record of the shared "
                                      f"Sink '{pr.target.name}'; the order inside that nanosecond is not reconstructed for a Sink behind "
                                      "several Servers")
        pr.data_sink._set(t, before.astype(np.int64), None)


def lower(sources: list, entities: list) -> LoweredGraph:
    sources = list(sources or [])
    entities = list(entities or [])
    plain = _plain_chains(sources, entities)
    if plain is not None:
        return LoweredGraph(plain)
    g = LoweredGraph()
    station_of_server: dict[int, int] = {}
    used_sinks: dict[int, int] = {}
    used_links: dict[int, int] = {}
    used_routers: dict[int, int] = {}
    fed_by: dict[int, str] = {}

    def check_sink(obj, owner):
        # one collector may hang behind several stations (`[Server(..., downstream=sink) for ...]`): every station logs its
        # own completions and write_back() merges them by completion time on the device (hs_merge_sink_records)
        used_sinks.setdefault(id(obj), len(g.stations))

    def check_server(sv: Server):
        if not isinstance(sv.service_time, (ExponentialLatency, ConstantLatency)):
            raise UnsupportedTopology(
                f"server '{sv.name}': service distribution {type(sv.service_time).__name__} is not lowered")
        if sv.concurrency > 32:
            raise UnsupportedTopology(f"server '{sv.name}': concurrency {sv.concurrency} > 32 is not lowered yet")

    def check_link(lk: NetworkLink, owner: str):
        if id(lk) in used_links:
            raise UnsupportedTopology(f"link '{lk.name}' is used by several senders (not lowered)")
        if not isinstance(lk.latency, ConstantLatency) or not (lk.latency.mean > 0):
            raise UnsupportedTopology(
                f"link '{lk.name}': the base latency must be a ConstantLatency > 0 -- it is the lookahead of the "
                "conservative time windows (the reference enforces min_latency > 0 the same way, parallel/link.py:41-45)")
        if lk.jitter is not None and not isinstance(lk.jitter, (ExponentialLatency, ConstantLatency)):
            raise UnsupportedTopology(f"link '{lk.name}': jitter {type(lk.jitter).__name__} is not lowered")
        # packet_loss_rate is lowered (the link's own LOSS stream).  bandwidth_bps is accepted as it is: requests built by
        # the lowered event providers carry no payload_size, so the transmission time (link.py:209-214) is 0 s and
        # bytes_transmitted stays 0, which is what the reference computes for them as well.
        if not isinstance(lk.egress, Server):
            raise UnsupportedTopology(f"link '{lk.name}' must deliver to a Server (got {type(lk.egress).__name__})")
        used_links[id(lk)] = len(g.stations)

    def attach_egress(st: Station, obj, owner: str):
        if obj is None:
            return
        if isinstance(obj, _SINKS):
            check_sink(obj, owner)
            st.sink = obj
        elif isinstance(obj, NetworkLink):
            check_link(obj, owner)
            st.links = (*st.links, obj)
        elif isinstance(obj, Server):
            # tandem queues (components/server/server.py:271-272): the completion is the next Server's arrival, at the same instant
            fed_by[id(obj)] = owner            # (several upstream Servers per Server: the engine's single-heap path, hs_engine.h)
            st.next_server = obj
        elif isinstance(obj, RandomRouter):
            if id(obj) in used_routers:
                raise UnsupportedTopology(f"router '{obj.name}' has several upstreams (not lowered)")
            used_routers[id(obj)] = len(g.stations)
            if not (1 <= len(obj.targets) <= 4):
                raise UnsupportedTopology(f"router '{obj.name}': 1..4 targets are lowered, got {len(obj.targets)}")
            if sum(isinstance(t, NetworkLink) for t in obj.targets) > 2:
                raise UnsupportedTopology(f"router '{obj.name}': at most two NetworkLink targets are lowered")
            st.router = obj
            for t in obj.targets:
                if isinstance(t, _SINKS):
                    if st.sink is not None and st.sink is not t:
                        raise UnsupportedTopology(f"router '{obj.name}': at most one distinct Sink target (it may be listed "
                                                  "several times)")
                    check_sink(t, f"router '{obj.name}'")
                    st.sink = t
                elif isinstance(t, NetworkLink):
                    check_link(t, f"router '{obj.name}'")
                    st.links = (*st.links, t)
                else:
                    raise UnsupportedTopology(
                        f"router '{obj.name}' targets {type(t).__name__}: only Sink-like collectors and NetworkLinks")
        else:
            raise UnsupportedTopology(
                f"{owner} forwards to {type(obj).__name__} '{getattr(obj, 'name', obj)}': only Sink / Counter / "
                "LatencyTracker / NetworkLink / RandomRouter leave a station on the engine")

    def add_server_station(sv: Server, src: Source | None):
        check_server(sv)
        station_of_server[id(sv)] = len(g.stations)
        st = Station(source=src, server=sv)
        attach_egress(st, sv.downstream, f"server '{sv.name}'")
        g.stations.append(st)

    for src in sources:
        if not isinstance(src, Source):
            raise UnsupportedTopology(f"source {type(src).__name__} is not a lowered Source")
        if not (src.rate > 0):
            raise UnsupportedTopology(f"source '{src.name}': rate must be > 0")
        tgt = src._event_provider._target
        if isinstance(tgt, Server):
            if id(tgt) in station_of_server:
                # several Sources feeding one Server: entities of their own on the same station (engine slots 1..3)
                st = g.stations[station_of_server[id(tgt)]]
                if len(st.more_sources) >= 3:
                    raise UnsupportedTopology(f"server '{tgt.name}' is fed by more than four Sources (the engine's slots per station)")
                for x in (st.source, src):
                    if not isinstance(x._time_provider.profile, ConstantRateProfile):
                        raise UnsupportedTopology(f"source '{x.name}': a time-varying profile next to further Sources of the "
                                                  "same Server is not lowered")
                st.more_sources = (*st.more_sources, src)
            else:
                add_server_station(tgt, src)
        elif isinstance(tgt, _SINKS):
            st = Station(source=src)
            check_sink(tgt, f"source '{src.name}'")
            st.sink = tgt
            g.stations.append(st)
        else:
            raise UnsupportedTopology(
                f"source '{src.name}' targets {type(tgt).__name__}: only Server / Sink / Counter / LatencyTracker")

    for ent in entities:
        if isinstance(ent, Server):
            if id(ent) not in station_of_server:      # no Source of its own: fed by links, or never sees an event
                add_server_station(ent, None)
        elif isinstance(ent, (Sink, Counter, LatencyTracker, Source, NetworkLink, RandomRouter)):
            continue
        elif isinstance(ent, Entity):
            raise UnsupportedTopology(f"entity {type(ent).__name__} '{ent.name}' is not lowered to the engine")
        else:
            raise UnsupportedTopology(f"object {ent!r} is not an Entity")
    if not g.stations:
        raise UnsupportedTopology("nothing to simulate: no sources and no servers")

    # Station order defines the entity stream numbering (station i's entities draw from stream base i): Servers in the
    # order the user listed them in `entities` (construction order of the topology), then whatever is only reachable
    # from a Source, in source order.
    pos = {id(e): k for k, e in enumerate(entities)}
    src_pos = {id(s): k for k, s in enumerate(sources)}

    def order_key(st: Station):
        if st.server is not None and id(st.server) in pos:
            return (0, pos[id(st.server)])
        if st.server is None and st.sink is not None and id(st.sink) in pos:
            return (0, pos[id(st.sink)])          # Source -> Sink station: its collector's place
        return (1, src_pos.get(id(st.source), 0))

    g.stations.sort(key=order_key)
    station_of_server = {id(st.server): i for i, st in enumerate(g.stations) if st.server is not None}

    for i, st in enumerate(g.stations):                     # tandem queues: the downstream Servers' stations
        if st.next_server is None:
            continue
        st.next_station = station_of_server.get(id(st.next_server), -1)
        if st.next_station < 0:
            raise UnsupportedTopology(f"server '{st.server.name}' forwards to server '{st.next_server.name}', which is not listed in "
                                      "`entities` of this Simulation")
    if any(st.next_server is not None for st in g.stations) and (any(st.links or st.router is not None for st in g.stations)):
        raise UnsupportedTopology("Server(downstream=<Server>) inside a network of NetworkLinks / RandomRouters is not lowered yet")
    # resolve link destinations (every Server is a station by now) in station order, router-target order
    for i, st in enumerate(g.stations):
        for lk in st.links:
            dst = station_of_server.get(id(lk.egress))
            if dst is None:
                raise UnsupportedTopology(
                    f"link '{lk.name}' delivers to server '{lk.egress.name}', which is not part of this Simulation")
            st.link_ids = (*st.link_ids, len(g.links))
            g.links.append((lk, i, dst))
    return g


class LazyRecords:
    """The Sink records of a run, left on the device until a Sink's lists are first read (0.5 GB at the headline size: the download
    and the per-Sink slicing were most of Simulation.run()'s wall time).  Owns the engine until then."""

    def __init__(self, eng, counts: np.ndarray, keep_engine: bool = False):
        self._eng = eng
        self._keep = keep_engine               # probes read their samples from the engine later (write_back_plain_probes)
        self.off = np.zeros(len(counts) + 1, np.int64)
        np.cumsum(counts, out=self.off[1:])
        self._t = self._cr = None

    def fetch(self):
        if self._t is None:
            _, self._t, self._cr = self._eng.read_sinks()
            if not self._keep:
                self._eng.close()
                self._eng = None
        return self._t, self._cr

    def engine(self):
        if self._eng is None:
            raise RuntimeError("the engine of this run has been closed")
        return self._eng

    def count(self, i: int) -> int:
        return int(self.off[i + 1] - self.off[i])

    SINGLE_READS = 64        # Sinks read one by one (a device-side gather of that Sink's column each) before everything comes down

    def records(self, i: int):
        # one Sink's records: its column of the device logs, gathered on the device (hs_engine_read_sink: microseconds) -- the
        # first `latencies_s` of a 65 536-chain run used to wait for the download of all 0.5 GB.  A caller that walks the Sinks
        # gets the bulk download after a few of them.
        if self._t is None and self._eng is not None:
            self._singles = getattr(self, "_singles", 0) + 1
            if self._singles <= self.SINGLE_READS:
                return self._eng.read_sink(i, cap=max(self.count(i), 1))
        t, cr = self.fetch()
        a, b = int(self.off[i]), int(self.off[i + 1])
        return t[a:b], cr[a:b]

    def close(self):
        if self._eng is not None:
            self._eng.close()
            self._eng = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_back_plain(pc: PlainChains, stats: dict, records: LazyRecords, device: int = 0) -> None:
    """write_back() for PlainChains: every object is BOUND to its row of the run's result arrays (entities._Stat: one attribute
    store per object instead of one per counter) and the Sinks to the lazily downloaded records -- the attributes users read
    (`stats_accepted`, `stats.requests_completed`, `generated_count`, `events_received`, `latencies_s` ...) are unchanged.
    The binding itself (4 x n attribute stores) is deferred until a result attribute of any lowered entity is first touched
    (entities._PENDING): at 65 536 chains it costs 100 x the device run."""
    from . import entities as E

    def bind_row(i):
        src, sv, sk = pc.sources[i], pc.servers[i], pc.sinks[i]
        b = (stats, i)
        src._bound = b
        src._event_provider._bound = b
        sv._bound = b
        sv._queue._bound = b
        if sk is not None:
            sk._lazy = (records, i)
            sk._device = device

    class _Bind:
        """The deferred binding of one run: everything at once (call), or the chain of ONE object (find_and_bind: entities._resolve)."""

        lookups = 0

        def __call__(self):
            for i in range(len(pc.sources)):
                bind_row(i)

        def find_and_bind(self, obj) -> bool:
            t = type(obj)
            try:                                           # (list.index: an identity search at C speed, ~1 ms per 65 536 objects)
                if t is Source:
                    i = pc.sources.index(obj)
                elif t is Server:
                    i = pc.servers.index(obj)
                elif t in (Sink, Counter, LatencyTracker):
                    i = pc.sinks.index(obj)
                else:
                    return False                           # a provider / queue object: the bulk binding
            except ValueError:
                return False                               # not an object of this run
            bind_row(i)
            return True

    E._flush_pending()             # (an earlier run's results first: bindings apply in run order)
    E._PENDING.append(_Bind())


def write_back(g: LoweredGraph, stats: dict, counts: np.ndarray, t_ns: np.ndarray, created_ns: np.ndarray,
               net_stats: dict | None = None, lo: int = 0, hi: int | None = None, device: int = 0) -> None:
    """Put the engine's per-LP results onto the user's objects, under the attribute names the reference uses.
    `stats` / `counts` are indexed by station; `lo:hi` restricts the write-back to one shard's stations (the sink
    records `t_ns` / `created_ns` are then that shard's records only)."""
    off = 0
    hi = len(g.stations) if hi is None else hi
    per_sink: dict[int, tuple] = {}
    for i, st in enumerate(g.stations):
        if not (lo <= i < hi):
            continue
        if st.source is not None:
            st.source._generated_count = int(stats["generated"][i])
            ep = st.source._event_provider
            ep._generated = int(stats["accepted"][i] + stats["dropped"][i]) if st.server is not None else int(
                counts[i])
            if st.more_sources:              # the Server's arrivals are the union: per Source, its own tick count
                ep._generated = st.source._generated_count
                for slot, x in enumerate(st.more_sources):
                    x._generated_count = int(stats["generated_more"][slot][i])
                    x._event_provider._generated = x._generated_count
        if st.server is not None:
            sv = st.server
            sv._queue.stats_accepted = int(stats["accepted"][i])
            sv._queue.stats_dropped = int(stats["dropped"][i])
            sv._queue.depth = int(stats["queue_depth"][i])
            sv._requests_completed = int(stats["completed"][i])
            sv._requests_rejected = int(stats["rejected"][i])
            sv._total_service_time = float(stats["total_service_s"][i])
            sv._active = int(stats["active"][i])
        c = int(counts[i])
        if st.sink is not None:
            per_sink.setdefault(id(st.sink), (st.sink, []))[1].append((t_ns[off:off + c], created_ns[off:off + c]))
        off += c
        if net_stats is not None:
            for lk, l in zip(st.links, st.link_ids):
                lk.packets_sent = int(net_stats["link_packets_sent"][l])
                lk.packets_dropped = int(net_stats["link_packets_dropped"][l])
                lk._entered = int(net_stats["link_entered"][l])
            if st.router is not None:
                st.router.stats_routed = int(net_stats["routed"][i])
                tc = {}
                ids = iter(st.link_ids)
                seen_sink = False
                for t in st.router.targets:
                    if isinstance(t, _SINKS):
                        if not seen_sink:                  # (listed several times: one counter, by name -- random_router.py:37)
                            tc[t.name] = tc.get(t.name, 0) + c
                        seen_sink = True
                    else:
                        tc[t.name] = tc.get(t.name, 0) + int(net_stats["link_entered"][next(ids)])
                st.router.target_counts = {k: v for k, v in tc.items() if v}
    for sink, parts in per_sink.values():
        t = np.concatenate([p[0] for p in parts]) if len(parts) > 1 else parts[0][0].copy()
        cr = np.concatenate([p[1] for p in parts]) if len(parts) > 1 else parts[0][1].copy()
        if len(parts) > 1:          # a Sink shared by several stations: its lists are in global processing order
            t, cr = np.ascontiguousarray(t, np.int64), np.ascontiguousarray(cr, np.int64)
            rc = N.lib().hs_merge_sink_records(device, len(t), t.ctypes.data, cr.ctypes.data)
            if rc != N.HS_OK:
                raise N.EngineError(rc, (N.lib().hs_lb_last_error(None) or b"").decode())
        sink._set_records(t, cr)
        sink._device = device           # Sink.latency_stats() sorts on the GPU the run used (one rank per GPU: not device 0)


# ----------------------------------------------------------------------------------------------
# load-balancer topologies (BASELINE configs[4]):  Sources -> LoadBalancer(ConsistentHash) -> Servers -> Sink(s)
# ----------------------------------------------------------------------------------------------
@dataclass
class LbGraph:
    sources: list
    lb: LoadBalancer
    backends: list
    sinks: list                    # one shared collector, one per backend, or [] (no downstream anywhere)
    shared_sink: bool
    probes: list = field(default_factory=list)       # Probes on backend Servers / Sinks, in `probes=[...]` order

    def probe_arrays(self):
        """(target kind, target index, metric id, interval) per probe for LoadBalancerEngine.set_probes."""
        kinds, idx, met, iv = [], [], [], []
        for pr in self.probes:
            if isinstance(pr.target, Server):
                kinds.append(0)
                idx.append(next(j for j, b in enumerate(self.backends) if b is pr.target))
                m = pr.metric
                met.append(N.PROBE_METRICS[Probe.engine_metric(m)])
            elif isinstance(pr.target, Source):
                kinds.append(2)
                idx.append(next(j for j, x in enumerate(self.sources) if x is pr.target))
                met.append(N.PROBE_METRICS["generated_count"])
            else:
                kinds.append(1)
                idx.append(next(j for j, k in enumerate(self.sinks) if k is pr.target))
                met.append(N.PROBE_METRICS["events_received"])
            iv.append(pr.interval)
        return kinds, idx, met, iv

    def engine_arrays(self):
        from .lb_engine import LbBackendArrays, LbSourceArrays

        S, B = len(self.sources), len(self.backends)
        src = LbSourceArrays(
            n=S, src_rate=np.array([s.rate for s in self.sources], np.float64),
            n_clients=np.array([getattr(s._event_provider, "_n_clients", 1) for s in self.sources], np.int64),
            src_kind=np.array([N.SRC_POISSON if s._time_provider.kind == "poisson" else N.SRC_CONSTANT
                               for s in self.sources], np.uint8),
            src_stop_after_ns=np.array([-1 if s._event_provider._stop_after is None
                                        else s._event_provider._stop_after.nanoseconds for s in self.sources], np.int64))
        for i, s in enumerate(self.sources):           # Source.with_profile in front of the LoadBalancer (src_rate = the peak)
            pr = s._time_provider.profile
            if isinstance(pr, ConstantRateProfile):
                continue
            if src.src_profile_kind is None:
                src.src_profile_kind = np.zeros(S, np.uint8)
                src.src_profile_params = np.zeros((S, 4), np.float64)
            if isinstance(pr, LinearRampProfile):
                src.src_profile_kind[i] = N.PROF_LINEAR_RAMP
                src.src_profile_params[i, :3] = (pr.duration_s, pr.start_rate, pr.end_rate)
            else:
                src.src_profile_kind[i] = N.PROF_SPIKE
                src.src_profile_params[i] = (pr.baseline_rate, pr.spike_rate, pr.warmup_s, pr.spike_duration_s)
        caps = [b._policy.capacity for b in self.backends]
        be = LbBackendArrays(
            n=B, names=[b.name for b in self.backends],
            concurrency=np.array([b.concurrency for b in self.backends], np.int32),
            svc_kind=np.array([N.LAT_EXPONENTIAL if isinstance(b.service_time, ExponentialLatency) else N.LAT_CONSTANT
                               for b in self.backends], np.uint8),
            svc_mean_s=np.array([b.service_time.mean for b in self.backends], np.float64),
            queue_cap=np.array([-1 if c == float("inf") else int(c) for c in caps], np.int64),
            egress=np.full(B, N.EGRESS_SINK if self.sinks else N.EGRESS_NONE, np.uint8))
        return src, be


def find_load_balancer(sources: list, entities: list):
    """The LoadBalancer of a load-balancer topology, or None when the graph has none."""
    from operator import attrgetter as ag

    lbs = [e for e in (entities or []) if isinstance(e, LoadBalancer)] if any(
        issubclass(t, LoadBalancer) for t in set(map(type, entities or []))) else []
    try:                                           # (one C-level pass; an object without the attributes is no lowered Source)
        targets = list(map(ag("_event_provider._target"), sources or []))
    except AttributeError:
        targets = [getattr(getattr(s, "_event_provider", None), "_target", None) for s in sources or []]
    if any(issubclass(t, LoadBalancer) for t in set(map(type, targets))):
        for t in targets:
            if isinstance(t, LoadBalancer) and all(t is not x for x in lbs):
                lbs.append(t)
    if not lbs:
        return None
    if len(lbs) > 1:
        raise UnsupportedTopology("several LoadBalancers in one Simulation are not lowered")
    return lbs[0]


def lower_lb(sources: list, entities: list, lb: LoadBalancer) -> LbGraph:
    """Stream numbering: Source i (list order) -> stream base i; backend j (add_backend order) -> len(sources) + j."""
    sources = list(sources or [])
    if not sources:
        raise UnsupportedTopology("a load-balancer topology needs at least one Source")
    for s in sources:
        if not isinstance(s, Source):
            raise UnsupportedTopology(f"source {type(s).__name__} is not a lowered Source")
        ep = s._event_provider
        if isinstance(lb.strategy, ConsistentHash) and not isinstance(ep, ClientKeyEventProvider):
            raise UnsupportedTopology(
                f"source '{s.name}': requests for a key-based LoadBalancer must come from a ClientKeyEventProvider "
                "(ConsistentHash falls back to RoundRobin for key-less requests: use strategy=RoundRobin() for those)")
        if not isinstance(ep, (ClientKeyEventProvider, SimpleEventProvider)):
            raise UnsupportedTopology(f"source '{s.name}': event provider {type(ep).__name__} is not lowered")
        if ep._target is not lb:
            raise UnsupportedTopology(f"source '{s.name}' does not target the LoadBalancer '{lb.name}'")
        if not (s.rate > 0):
            raise UnsupportedTopology(f"source '{s.name}': rate must be > 0")
    if not isinstance(lb.strategy, (ConsistentHash, RoundRobin, Random)):
        raise UnsupportedTopology(f"strategy {type(lb.strategy).__name__} is not lowered")
    backends = lb.all_backends
    if not backends:
        raise UnsupportedTopology("a LoadBalancer without backends rejects every request; nothing to lower")
    downs = []
    for b in backends:
        if not isinstance(b, Server):
            raise UnsupportedTopology(f"backend '{b.name}' is a {type(b).__name__}: only Server backends are lowered")
        if not isinstance(b.service_time, (ExponentialLatency, ConstantLatency)):
            raise UnsupportedTopology(f"backend '{b.name}': service distribution {type(b.service_time).__name__} is not lowered")
        if b.concurrency > 32:
            raise UnsupportedTopology(f"backend '{b.name}': concurrency {b.concurrency} > 32 is not lowered yet")
        d = b.downstream
        if d is not None and not isinstance(d, _SINKS):
            raise UnsupportedTopology(f"backend '{b.name}' forwards to {type(d).__name__}: only Sink-like collectors")
        downs.append(d)
    known = {id(lb)} | {id(b) for b in backends} | {id(d) for d in downs if d is not None} | {id(s) for s in sources}
    for e in entities or []:
        if id(e) not in known:
            raise UnsupportedTopology(f"entity '{getattr(e, 'name', e)}' is not part of the load-balancer topology")
    if all(d is None for d in downs):
        sinks, shared = [], False
    elif all(d is downs[0] for d in downs):
        sinks, shared = [downs[0]], True
    elif all(d is not None for d in downs) and len({id(d) for d in downs}) == len(downs):
        sinks, shared = list(downs), False
    else:
        raise UnsupportedTopology("backends must all share ONE Sink, each have their own, or all have none")
    return LbGraph(sources=sources, lb=lb, backends=backends, sinks=sinks, shared_sink=shared)


def attach_lb_probes(g: LbGraph, probes: list) -> None:
    """Probe.on(<backend Server> | <Sink>, metric, interval) on a load-balancer graph (csrc/hs_lb.hip section 5)."""
    from .entities import Probe

    if probes and not g.sinks:
        raise UnsupportedTopology("probes on a load-balancer graph are read off the backends' completion logs: a Sink downstream is needed")
    for pr in probes or []:
        if not isinstance(pr, Probe):
            raise UnsupportedTopology(f"probe {type(pr).__name__} is not a lowered Probe")
        if Probe.engine_metric(pr.metric) not in N.PROBE_METRICS:
            raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not sampled on the engine "
                                      f"(lowered: {', '.join(sorted(Probe._LOWERED))})")
        if any(pr.target is b for b in g.backends):
            if pr.metric in ("generated_count", "_generated_count", "events_received"):
                raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not an attribute of Server")
        elif any(pr.target is k for k in g.sinks):
            if pr.metric != "events_received":
                raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not an attribute of {type(pr.target).__name__}")
        elif any(pr.target is x for x in g.sources):
            if pr.metric not in ("generated_count", "_generated_count"):
                raise UnsupportedTopology(f"probe '{pr.name}': metric '{pr.metric}' is not an attribute of Source")
            if pr.target._event_provider._stop_after is not None:
                raise UnsupportedTopology(f"probe '{pr.name}': a Source with stop_after keeps ticking without Requests, which the "
                                          "engine's tick log does not hold (not sampled on load-balancer graphs)")
        else:
            raise UnsupportedTopology(f"probe '{pr.name}': on a load-balancer graph the Sources, the backend Servers and the Sinks are "
                                      f"sampled (not {type(pr.target).__name__} '{getattr(pr.target, 'name', pr.target)}')")
        g.probes.append(pr)


def write_back_lb(g: LbGraph, stats: dict, eng) -> None:
    """Engine results -> the user's objects, under the reference's attribute names."""
    for j, pr in enumerate(g.probes):
        t, v = eng.read_probe(j)
        pr.data_sink._set(t, v, Probe.value_map(pr.metric, pr.target))
    for i, s in enumerate(g.sources):
        s._generated_count = int(stats["generated"][i])
    lb = g.lb
    lb._requests_received, lb._requests_forwarded, lb._requests_failed, lb._no_backend_available, lb._in_flight_count = (
        int(v) for v in stats["lb"])
    if isinstance(lb.strategy, RoundRobin):
        lb.strategy._index += lb._requests_forwarded              # one select per forwarded Request (strategies.py:66-67)
    for j, b in enumerate(g.backends):
        b._queue.stats_accepted = int(stats["accepted"][j])
        b._queue.stats_dropped = int(stats["dropped"][j])
        b._queue.depth = int(stats["queue_depth"][j])
        b._requests_completed = int(stats["completed"][j])
        b._requests_rejected = int(stats["rejected"][j])
        b._total_service_time = float(stats["total_service_s"][j])
        b._active = int(stats["active"][j])
        lb._backends[b.name].total_requests = int(stats["total_requests"][j])
    if g.shared_sink:
        g.sinks[0]._set_records(*eng.read_sink(0))
        g.sinks[0]._device_latency_stats = eng.latency_stats()      # Sink.latency_stats(): sorted on the device
    else:
        for j, k in enumerate(g.sinks):
            k._set_records(*eng.read_sink(j, cap=int(stats["sink_received"][j])))
