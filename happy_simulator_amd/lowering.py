"""Graph lowering: the user's Source / Server / Sink objects -> station LPs (struct-of-arrays).

Walks `sources` and `entities` the way the reference's topology discovery does
(happysimulator/visual/topology.py:81-146: follow `downstream_entities()`), and refuses -- explicitly,
never silently -- anything the engine does not run.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .engine import StationArrays
from .entities import (ConstantLatency, Counter, Entity, ExponentialLatency, LatencyTracker, Server, Sink, Source,
                       _RecordSink)


class UnsupportedTopology(NotImplementedError):
    """The entity graph contains something the GPU engine does not lower (stay on the reference's CPU loop)."""


@dataclass
class Station:
    source: Source | None = None
    server: Server | None = None
    sink: _RecordSink | None = None


@dataclass
class LoweredGraph:
    stations: list[Station] = field(default_factory=list)

    def arrays(self) -> StationArrays:
        n = len(self.stations)
        a = StationArrays.uniform(n)
        for i, st in enumerate(self.stations):
            if st.source is not None:
                prov = st.source._time_provider
                a.src_kind[i] = N.SRC_POISSON if prov.kind == "poisson" else N.SRC_CONSTANT
                a.src_rate[i] = float(prov.profile.rate)
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
        return a


def lower(sources: list, entities: list) -> LoweredGraph:
    sources = list(sources or [])
    entities = list(entities or [])
    g = LoweredGraph()
    used_servers: dict[int, int] = {}
    used_sinks: dict[int, int] = {}

    def check_sink(obj, owner):
        if not isinstance(obj, (Sink, Counter, LatencyTracker)):
            raise UnsupportedTopology(
                f"{owner} forwards to {type(obj).__name__} '{getattr(obj, 'name', obj)}': only Sink / Counter / "
                "LatencyTracker terminate a station on the engine")
        if id(obj) in used_sinks:
            raise UnsupportedTopology(
                f"sink '{obj.name}' has several upstreams; merged sinks need cross-LP ordering (not lowered yet)")
        used_sinks[id(obj)] = len(g.stations)

    def check_server(sv: Server):
        if not isinstance(sv.service_time, (ExponentialLatency, ConstantLatency)):
            raise UnsupportedTopology(
                f"server '{sv.name}': service distribution {type(sv.service_time).__name__} is not lowered")
        if sv.concurrency > 16:
            raise UnsupportedTopology(f"server '{sv.name}': concurrency {sv.concurrency} > 16 is not lowered yet")

    for src in sources:
        if not isinstance(src, Source):
            raise UnsupportedTopology(f"source {type(src).__name__} is not a lowered Source")
        if not (src.rate > 0):
            raise UnsupportedTopology(f"source '{src.name}': rate must be > 0")
        tgt = src._event_provider._target
        st = Station(source=src)
        if isinstance(tgt, Server):
            if id(tgt) in used_servers:
                raise UnsupportedTopology(f"server '{tgt.name}' is fed by several sources (not lowered yet)")
            used_servers[id(tgt)] = len(g.stations)
            check_server(tgt)
            st.server = tgt
            if tgt.downstream is not None:
                check_sink(tgt.downstream, f"server '{tgt.name}'")
                st.sink = tgt.downstream
        elif isinstance(tgt, (Sink, Counter, LatencyTracker)):
            check_sink(tgt, f"source '{src.name}'")
            st.sink = tgt
        else:
            raise UnsupportedTopology(
                f"source '{src.name}' targets {type(tgt).__name__}: only Server / Sink / Counter / LatencyTracker")
        g.stations.append(st)

    for ent in entities:
        if isinstance(ent, Server):
            if id(ent) not in used_servers:       # a server nobody feeds: it exists, it never sees an event
                check_server(ent)
                used_servers[id(ent)] = len(g.stations)
                st = Station(server=ent)
                if ent.downstream is not None:
                    check_sink(ent.downstream, f"server '{ent.name}'")
                    st.sink = ent.downstream
                g.stations.append(st)
        elif isinstance(ent, (Sink, Counter, LatencyTracker, Source)):
            continue
        elif isinstance(ent, Entity):
            raise UnsupportedTopology(f"entity {type(ent).__name__} '{ent.name}' is not lowered to the engine")
        else:
            raise UnsupportedTopology(f"object {ent!r} is not an Entity")
    if not g.stations:
        raise UnsupportedTopology("nothing to simulate: no sources and no servers")
    return g


def write_back(g: LoweredGraph, stats: dict, counts: np.ndarray, t_ns: np.ndarray, created_ns: np.ndarray) -> None:
    """Put the engine's per-LP results onto the user's objects, under the attribute names the reference uses."""
    off = 0
    for i, st in enumerate(g.stations):
        if st.source is not None:
            st.source._generated_count = int(stats["generated"][i])
            ep = st.source._event_provider
            ep._generated = int(stats["accepted"][i] + stats["dropped"][i]) if st.server is not None else int(
                counts[i])
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
            st.sink._set_records(t_ns[off:off + c].copy(), created_ns[off:off + c].copy())
        off += c
