"""General entity graphs: what the lowered entity classes can be wired into beyond the station shape of lowering.lower().

The station engines take `[Sources] -> Server -> {Sink | NetworkLink | RandomRouter}` with one sender per link, at most four
Sources per Server, four router targets ...; lowering.lower() refuses the rest by name.  `lower_general()` takes exactly those
graphs -- a NetworkLink with several senders (components/network/link.py:114-189), a RandomRouter with any number of targets,
Servers and routers among them, and with several upstreams (components/random_router.py:32-45), Server(downstream=<Server>)
next to links (components/server/server.py:64-122), any number of Sources per Server (load/source.py:93-180), any
FixedConcurrency (server/concurrency.py:67-141) -- and runs them on the device's single-heap loop (csrc/hs_graph.hip,
include/hs_engine.h "General entity graphs"): the reference's own (time, _sort_index) order event by event, ~2.4 us per event on
one lane.  Exact, not fast: Simulation only comes here with a graph lower() refused.

Entity streams (DESIGN.md section 3 -- the one definition that is the engine's own): the k-th Source of `sources=[...]` draws
ARRIVAL from stream base k; the s-th Server / l-th NetworkLink / r-th RandomRouter in node order (entities in `entities=[...]`
order, then whatever is only reachable downstream, in discovery order) draws SERVICE / LINK + LOSS / ROUTE from base s / l / r.
For a graph built station by station this is the numbering of the station engines.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .entities import (ClientKeyEventProvider, ConsistentHash, ConstantLatency, ConstantRateProfile, Counter, Entity, ExponentialLatency,
                       LatencyTracker, LinearRampProfile, LoadBalancer, NetworkLink, Probe, Random, RandomRouter, RoundRobin, Server,
                       SimpleEventProvider, Sink, Source)

_SINKS = (Sink, Counter, LatencyTracker)
DEFAULT_MAX_EVENTS = 200_000_000          # ~ minutes on the one lane; Simulation(max_graph_events=) raises it


def _ptr(a):
    return None if a is None else a.__array_interface__["data"][0]      # (the address; `.ctypes.data_as` costs 3 us per array)


class GraphArrays:
    """Struct-of-arrays form of hs_graph_nodes (include/hs_engine.h)."""

    def __init__(self, n: int):
        self.n = n
        self.kind = np.zeros(n, np.uint8)
        self.target = np.full(n, -1, np.int32)
        self.stream_base = np.zeros(n, np.uint64)
        self.src_kind = np.full(n, N.SRC_POISSON, np.uint8)
        self.src_rate = np.ones(n, np.float64)
        self.src_stop_after_ns = np.full(n, -1, np.int64)
        self.concurrency = np.ones(n, np.int32)
        self.lat_kind = np.full(n, N.LAT_CONSTANT, np.uint8)
        self.lat_mean_s = np.zeros(n, np.float64)
        self.link_lat_min_s = np.zeros(n, np.float64)
        self.link_loss_rate = np.zeros(n, np.float64)
        self.queue_cap = np.full(n, -1, np.int64)
        self.rt_off = np.zeros(n, np.int32)
        self.rt_cnt = np.zeros(n, np.int32)
        self.rt_targets = np.zeros(0, np.int32)
        self.src_profile_kind = None         # [n] uint8 / [n, 4] float64: Sources with a time-varying profile
        self.src_profile_params = None
        self.probe_metric = None             # [n] uint8 / [n] float64: PROBE nodes
        self.probe_interval_s = None
        self.lb_strategy = None              # [n] uint8 / [n] int32: LB nodes; names (bytes) / name_off [n + 1] int32: their backends' names
        self.lb_vnodes = None
        self.names = None
        self.name_off = None
        self.src_n_clients = None            # [n] int64: Sources with a ClientKeyEventProvider

    def struct(self) -> N.GraphNodes:
        s = N.GraphNodes()
        s.n_nodes = self.n
        for name in ("kind", "target", "stream_base", "src_kind", "src_rate", "src_stop_after_ns", "concurrency", "lat_kind",
                     "lat_mean_s", "link_lat_min_s", "link_loss_rate", "queue_cap", "rt_off", "rt_cnt"):
            setattr(s, name, _ptr(getattr(self, name)))
        self.rt_targets = np.ascontiguousarray(self.rt_targets, np.int32)
        s.rt_targets = _ptr(self.rt_targets) if len(self.rt_targets) else None
        s.n_rt = len(self.rt_targets)
        for name in ("src_profile_kind", "src_profile_params", "probe_metric", "probe_interval_s", "lb_strategy", "lb_vnodes", "name_off",
                     "src_n_clients"):
            setattr(s, name, _ptr(getattr(self, name)))
        if self.names is not None:
            self._names_buf = C.create_string_buffer(self.names, max(len(self.names), 1))
            s.names = C.cast(self._names_buf, C.c_void_p)
        return s


class GraphEngine:
    """ctypes handle of one hs_graph.  No CPU fallback: without the library or a GPU the constructor raises."""

    def __init__(self, arrays: GraphArrays, *, seed: int = 42, start_ns: int = 0, device: int = 0, max_events: int = 0,
                 heap_capacity: int = 0, request_capacity: int = 0, record_capacity: int = 0, profile_budget: int = 0):
        self._lib = N.lib()
        self.arrays = arrays
        cfg = N.GraphConfig(struct_size=C.sizeof(N.GraphConfig), device=device, start_ns=start_ns, seed=seed,
                            heap_capacity=heap_capacity, request_capacity=request_capacity, record_capacity=record_capacity,
                            max_events=max_events, profile_budget=profile_budget)
        h = C.c_void_p()
        nodes = arrays.struct()
        rc = self._lib.hs_graph_create(C.byref(cfg), C.byref(nodes), C.byref(h))
        if rc != N.HS_OK:
            msg = (self._lib.hs_graph_last_error(None) or b"").decode()
            if rc == N.HS_E_NO_DEVICE:
                raise N.EngineUnavailable(msg)
            raise N.EngineError(rc, msg)
        self._h = h

    def _check(self, rc: int):
        if rc < 0:
            raise N.EngineError(rc, (self._lib.hs_graph_last_error(self._h) or b"").decode())
        return rc

    def schedule(self, node: int, time_ns: int) -> None:
        self._check(self._lib.hs_graph_schedule(self._h, int(node), int(time_ns)))

    def run_until(self, end_ns: int) -> None:
        self._check(self._lib.hs_graph_run_until(self._h, int(end_ns)))

    @staticmethod
    def run_many(engines: list, end_ns: int) -> None:
        """hs_graph_run_many: independent graphs (replicas, sweep points) to `end_ns` side by side, one workgroup each."""
        if not engines:
            return
        hs_ = (C.c_void_p * len(engines))(*[e._h for e in engines])
        rc = engines[0]._lib.hs_graph_run_many(hs_, len(engines), int(end_ns))
        engines[0]._check(rc)

    def summary(self) -> N.Summary:
        s = N.Summary()
        self._check(self._lib.hs_graph_get_summary(self._h, C.byref(s)))
        return s

    def stats(self) -> dict:
        n, nrt = self.arrays.n, max(len(self.arrays.rt_targets), 1)
        out = {k: np.zeros(nrt if k == "rt_taken" else (n, 6) if k == "lb" else n, np.float64 if k == "total_service_s" else np.int64)
               for k in N.GRAPH_STATS}
        st = N.GraphStats()
        for k, a in out.items():
            setattr(st, k, _ptr(a))
        self._check(self._lib.hs_graph_get_stats(self._h, C.byref(st)))
        out["rt_taken"] = out["rt_taken"][:len(self.arrays.rt_targets)]
        return out

    def records(self):
        """(Sink node, completion ns, created_at ns) of every Sink record, in processing order."""
        n = self._check(self._lib.hs_graph_read_records(self._h, None, None, None, 0))
        node, t, cr = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int64)
        if n:
            self._check(self._lib.hs_graph_read_records(self._h, _ptr(node), _ptr(t), _ptr(cr), n))
        return node, t, cr

    def close(self):
        if self._h:
            self._lib.hs_graph_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class GeneralGraph:
    """The node list of one Simulation: Sources in `sources=[...]` order, then the other entities."""

    def __init__(self, nodes: list, arrays: GraphArrays, node_of: dict):
        self.nodes = nodes              # entity objects by node id
        self.arrays = arrays
        self.node_of = node_of          # id(entity) -> node id


def lower_general(sources: list, entities: list, probes=None) -> GeneralGraph:
    """`sources` / `entities` of a Simulation -> the node arrays of the single-heap engine.  Refuses by name what that engine
    does not run either (lowering.UnsupportedTopology)."""
    from .lowering import UnsupportedTopology

    nodes: list = []
    node_of: dict[int, int] = {}

    def add(ent):
        if id(ent) not in node_of:
            node_of[id(ent)] = len(nodes)
            nodes.append(ent)
        return node_of[id(ent)]

    for src in sources or []:
        if not isinstance(src, Source):
            raise UnsupportedTopology(f"source {type(src).__name__} is not a lowered Source")
        if id(src) in node_of:
            raise UnsupportedTopology(f"source '{src.name}' is listed twice")
        add(src)
    n_src = len(nodes)
    for pr in probes or []:                                   # PROBE nodes behind the Sources, in `probes=[...]` order
        if not isinstance(pr, Probe):
            raise UnsupportedTopology(f"probe {type(pr).__name__} is not a lowered Probe")
        if id(pr) in node_of:
            raise UnsupportedTopology(f"probe '{pr.name}' is listed twice")
        add(pr)
    n_front = len(nodes)
    lowered = (Server, NetworkLink, RandomRouter, LoadBalancer) + _SINKS

    def check(ent, where):
        if isinstance(ent, Source):
            raise UnsupportedTopology(f"{where}: a Source takes no Requests")
        if not isinstance(ent, lowered):
            raise UnsupportedTopology(f"{where}: {type(ent).__name__} '{getattr(ent, 'name', ent)}' is not lowered to the engine")

    for ent in entities or []:
        if isinstance(ent, Source):
            if id(ent) not in node_of:
                raise UnsupportedTopology(f"source '{ent.name}' is listed in entities but not in sources: it would never start")
            continue
        if not isinstance(ent, Entity):
            raise UnsupportedTopology(f"object {ent!r} is not an Entity")
        check(ent, "entities")
        add(ent)
    # whatever is only reachable downstream (the reference never needs it listed: events carry their target), in discovery order
    k = 0
    while k < len(nodes):
        ent = nodes[k]
        k += 1
        if isinstance(ent, Probe):
            continue
        for d in ent.downstream_entities():
            if d is None:
                continue
            check(d, f"'{ent.name}' forwards to it")
            if isinstance(d, Server) and id(d) not in node_of:
                # (the reference hands a clock to what `entities` lists: an unlisted Server would fail on its first event)
                raise UnsupportedTopology(f"'{ent.name}' forwards to server '{d.name}', which is not listed in `entities` of this "
                                          "Simulation (not part of this Simulation)")
            add(d)
    n = len(nodes)
    a = GraphArrays(n)
    counters = {Server: 0, NetworkLink: 0, RandomRouter: 0}
    rt: list[int] = []
    for i, ent in enumerate(nodes):
        if isinstance(ent, Source):
            ep, prov = ent._event_provider, ent._time_provider
            if not isinstance(ep, (SimpleEventProvider, ClientKeyEventProvider)):
                raise UnsupportedTopology(f"source '{ent.name}': event provider {type(ep).__name__} is not lowered on a general graph")
            tgt = ep._target
            n_clients = getattr(ep, "_n_clients", 0)
            if isinstance(tgt, LoadBalancer) and isinstance(tgt.strategy, Random):
                n_clients = len(tgt.all_backends)             # the Request's draw IS the choice: backends[int(u * len)] (entities.Random)
            if n_clients:
                if a.src_n_clients is None:
                    a.src_n_clients = np.zeros(n, np.int64)
                a.src_n_clients[i] = n_clients
            if not (ent.rate > 0):
                raise UnsupportedTopology(f"source '{ent.name}': rate must be > 0")
            if not isinstance(ep._target, Entity) or id(ep._target) not in node_of:
                raise UnsupportedTopology(f"source '{ent.name}' has no lowered target")
            a.kind[i] = N.NODE_SOURCE
            a.target[i] = node_of[id(ep._target)]
            a.stream_base[i] = i                              # (Sources are nodes 0 .. n_src - 1, in `sources` order)
            a.src_kind[i] = N.SRC_POISSON if prov.kind == "poisson" else N.SRC_CONSTANT
            a.src_rate[i] = float(prov.profile.peak_rate)
            a.src_stop_after_ns[i] = -1 if ep._stop_after is None else ep._stop_after.nanoseconds
            pr = prov.profile
            if not isinstance(pr, ConstantRateProfile):        # its ticks come from the tick-table kernel, like the station engines'
                if a.src_profile_kind is None:
                    a.src_profile_kind = np.zeros(n, np.uint8)
                    a.src_profile_params = np.zeros((n, 4), np.float64)
                if isinstance(pr, LinearRampProfile):
                    a.src_profile_kind[i] = N.PROF_LINEAR_RAMP
                    a.src_profile_params[i, :3] = (pr.duration_s, pr.start_rate, pr.end_rate)
                else:
                    a.src_profile_kind[i] = N.PROF_SPIKE
                    a.src_profile_params[i] = (pr.baseline_rate, pr.spike_rate, pr.warmup_s, pr.spike_duration_s)
        elif isinstance(ent, Probe):
            m = Probe.engine_metric(ent.metric)
            if m not in N.PROBE_METRICS:
                raise UnsupportedTopology(f"probe '{ent.name}': metric '{ent.metric}' is not sampled on the engine")
            tgt = ent.target
            if id(tgt) not in node_of:
                raise UnsupportedTopology(f"probe '{ent.name}': its target is not an entity of this Simulation")
            want = Source if m == "generated_count" else _SINKS if m == "events_received" else Server
            if not isinstance(tgt, want):
                raise UnsupportedTopology(f"probe '{ent.name}': metric '{ent.metric}' is not an attribute of {type(tgt).__name__}")
            if a.probe_metric is None:
                a.probe_metric = np.full(n, N.PROBE_NONE, np.uint8)
                a.probe_interval_s = np.ones(n, np.float64)
            a.kind[i] = N.NODE_PROBE
            a.target[i] = node_of[id(tgt)]
            a.probe_metric[i] = N.PROBE_METRICS[m]
            a.probe_interval_s[i] = ent.interval
        elif isinstance(ent, Server):
            svc = ent.service_time
            if not isinstance(svc, (ExponentialLatency, ConstantLatency)):
                raise UnsupportedTopology(f"server '{ent.name}': service distribution {type(svc).__name__} is not lowered")
            a.kind[i] = N.NODE_SERVER
            a.stream_base[i] = counters[Server]
            counters[Server] += 1
            a.concurrency[i] = ent.concurrency
            a.lat_kind[i] = N.LAT_EXPONENTIAL if isinstance(svc, ExponentialLatency) else N.LAT_CONSTANT
            a.lat_mean_s[i] = svc.mean
            cap = ent._policy.capacity
            a.queue_cap[i] = -1 if cap == float("inf") else int(cap)
            d = ent.downstream
            a.target[i] = -1 if d is None else node_of[id(d)]
        elif isinstance(ent, NetworkLink):
            if not isinstance(ent.latency, ConstantLatency) or not (ent.latency.mean >= 0):
                raise UnsupportedTopology(f"link '{ent.name}': the base latency must be a ConstantLatency >= 0")
            if ent.jitter is not None and not isinstance(ent.jitter, (ExponentialLatency, ConstantLatency)):
                raise UnsupportedTopology(f"link '{ent.name}': jitter {type(ent.jitter).__name__} is not lowered")
            a.kind[i] = N.NODE_LINK
            a.stream_base[i] = counters[NetworkLink]
            counters[NetworkLink] += 1
            a.link_lat_min_s[i] = ent.latency.mean
            if ent.jitter is not None:
                a.lat_kind[i] = N.LAT_EXPONENTIAL if isinstance(ent.jitter, ExponentialLatency) else N.LAT_CONSTANT
                a.lat_mean_s[i] = ent.jitter.mean
            a.link_loss_rate[i] = ent.packet_loss_rate
            a.target[i] = -1 if ent.egress is None else node_of[id(ent.egress)]
        elif isinstance(ent, LoadBalancer):
            backends = ent.all_backends
            for b in backends:
                if not isinstance(b, Server):
                    raise UnsupportedTopology(f"backend '{b.name}' of '{ent.name}' is a {type(b).__name__}: only Server backends are lowered")
            if a.lb_strategy is None:
                a.lb_strategy = np.full(n, N.LB_ROUND_ROBIN, np.uint8)
                a.lb_vnodes = np.zeros(n, np.int32)
            st = ent.strategy
            a.kind[i] = N.NODE_LB
            a.lb_strategy[i] = (N.LB_CONSISTENT_HASH if isinstance(st, ConsistentHash) else N.LB_ROUND_ROBIN if isinstance(st, RoundRobin)
                                else N.LB_RANDOM)
            a.lb_vnodes[i] = getattr(st, "virtual_nodes", 0)
            a.rt_off[i] = len(rt)
            a.rt_cnt[i] = len(backends)
            rt.extend(node_of[id(b)] for b in backends)
        elif isinstance(ent, RandomRouter):
            if not ent.targets:
                raise UnsupportedTopology(f"router '{ent.name}' has no targets")
            a.kind[i] = N.NODE_ROUTER
            a.stream_base[i] = counters[RandomRouter]
            counters[RandomRouter] += 1
            a.rt_off[i] = len(rt)
            a.rt_cnt[i] = len(ent.targets)
            rt.extend(node_of[id(t)] for t in ent.targets)
        else:
            a.kind[i] = N.NODE_SINK
    a.rt_targets = np.array(rt, np.int32)
    if a.lb_strategy is not None:
        _check_keys(nodes, node_of, a)
        blobs = [(x.name.encode() if isinstance(x, Server) else b"") for x in nodes]     # (only an LB's backends need their name)
        a.names = b"".join(blobs)
        a.name_off = np.zeros(n + 1, np.int32)
        a.name_off[1:] = np.cumsum([len(b) for b in blobs])
    assert all(a.kind[i] == N.NODE_SOURCE for i in range(n_src)) and all(a.kind[i] == N.NODE_PROBE for i in range(n_src, n_front))
    return GeneralGraph(nodes, a, node_of)


MAX_PARTS = 2048          # heaps one Simulation is spread over (hs_graph_run_parts): components beyond that share heaps


def split_parts(a: GraphArrays, max_parts: int = MAX_PARTS):
    """The graph's parts no Request can cross -- its connected components, packed into at most `max_parts` groups of neighbouring
    components -- as [(node ids ascending, rt positions, GraphArrays of the part)], or None when the graph is one component.
    Node order inside a part is the Simulation's (Sources first, then Probes: hs_graph_nodes' contract)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    n = a.n
    rt = np.asarray(a.rt_targets, np.int64)
    has_t = np.nonzero(a.target >= 0)[0]
    rt_owner = np.repeat(np.arange(n, dtype=np.int64), a.rt_cnt.astype(np.int64))
    if len(rt_owner) != len(rt):                       # (rt_off / rt_cnt do not tile rt_targets: leave it to the one heap)
        return None
    src = np.concatenate([has_t, rt_owner])
    dst = np.concatenate([a.target[has_t].astype(np.int64), rt])
    n_comp, label = connected_components(coo_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(n, n)), directed=False)
    if n_comp < 2:
        return None
    # components in the order of their first node; neighbours share a heap when there are more components than heaps
    first = np.full(n_comp, n, np.int64)
    np.minimum.at(first, label, np.arange(n))
    rank = np.empty(n_comp, np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(n_comp)
    k = min(n_comp, max_parts)
    part_of_node = rank[label] * k // n_comp
    order = np.argsort(part_of_node, kind="stable")                    # node ids grouped by part, ascending inside a part
    bounds = np.searchsorted(part_of_node[order], np.arange(k + 1))
    rt_off = np.asarray(a.rt_off, np.int64)
    new_index = np.empty(n, np.int64)
    per_node = [nm for nm in ("kind", "stream_base", "src_kind", "src_rate", "src_stop_after_ns", "concurrency", "lat_kind", "lat_mean_s",
                              "link_lat_min_s", "link_loss_rate", "queue_cap", "src_profile_kind", "src_profile_params", "probe_metric",
                              "probe_interval_s", "lb_strategy", "lb_vnodes", "src_n_clients") if getattr(a, nm) is not None]
    names = None if a.names is None else [a.names[a.name_off[i]:a.name_off[i + 1]] for i in range(n)]
    parts = []
    for p in range(k):
        ids = order[bounds[p]:bounds[p + 1]]
        m = len(ids)
        new_index[ids] = np.arange(m)
        b = GraphArrays(m)
        for nm in per_node:
            setattr(b, nm, np.ascontiguousarray(getattr(a, nm)[ids]))
        t = a.target[ids].astype(np.int64)
        b.target = np.where(t >= 0, new_index[np.maximum(t, 0)], -1).astype(np.int32)
        cnt = a.rt_cnt[ids].astype(np.int64)
        b.rt_cnt = cnt.astype(np.int32)
        b.rt_off = (np.cumsum(cnt) - cnt).astype(np.int32)
        total = int(cnt.sum())
        if total:
            pos = np.repeat(rt_off[ids] - (np.cumsum(cnt) - cnt), cnt) + np.arange(total)
            b.rt_targets = new_index[rt[pos]].astype(np.int32)
        else:
            pos = np.zeros(0, np.int64)
            b.rt_targets = np.zeros(0, np.int32)
        if names is not None:
            blobs = [names[i] for i in ids]
            b.names = b"".join(blobs)
            b.name_off = np.zeros(m + 1, np.int32)
            b.name_off[1:] = np.cumsum([len(x) for x in blobs])
        parts.append((ids, pos, b))
    return parts


class PartRun:
    """The engines of one Simulation's parts behind GraphEngine's reading interface (summary / stats / records), merged back into the
    Simulation's node numbering."""

    def __init__(self, arrays: GraphArrays, parts, engines):
        self.arrays, self.parts, self.engines = arrays, parts, engines

    def run(self, end_ns: int) -> bool:
        """hs_graph_run_parts: True when the parts' results are the Simulation's, False when the run is undecided (one heap decides)."""
        hs_ = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        rc = self.engines[0]._lib.hs_graph_run_parts(hs_, len(self.engines), int(end_ns))
        self.engines[0]._check(rc)
        return rc == 0

    def summary(self) -> N.Summary:
        subs = [e.summary() for e in self.engines]
        out = N.Summary()
        out.events_processed = sum(s.events_processed for s in subs)
        out.final_time_ns = max(s.final_time_ns for s in subs)           # (the one event beyond the end, or the last event of a drained run)
        for k in range(N.EV_KINDS):
            out.events_by_kind[k] = sum(s.events_by_kind[k] for s in subs)
        out.requests_completed = sum(s.requests_completed for s in subs)
        out.sink_records = sum(s.sink_records for s in subs)
        out.launches = sum(s.launches for s in subs)
        out.last_run_ms = max(s.last_run_ms for s in subs)
        out.kernel_ms = out.last_run_ms
        return out

    def stats(self) -> dict:
        n, nrt = self.arrays.n, len(self.arrays.rt_targets)
        out = {k: np.zeros(nrt if k == "rt_taken" else (n, 6) if k == "lb" else n, np.float64 if k == "total_service_s" else np.int64)
               for k in N.GRAPH_STATS}
        for (ids, pos, _b), e in zip(self.parts, self.engines):
            st = e.stats()
            for k, v in st.items():
                if k == "rt_taken":
                    out[k][pos] = v
                else:
                    out[k][ids] = v
        return out

    def records(self):
        node, t, cr = [], [], []
        for (ids, _pos, _b), e in zip(self.parts, self.engines):
            nd, tt, cc = e.records()
            node.append(ids[nd].astype(np.int32))
            t.append(tt)
            cr.append(cc)
        return np.concatenate(node), np.concatenate(t), np.concatenate(cr)

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _reaches(nodes, node_of, start) -> set:
    """Node ids a Request that enters `start` can visit."""
    seen, todo = set(), [node_of[id(start)]]
    while todo:
        i = todo.pop()
        if i in seen:
            continue
        seen.add(i)
        todo.extend(node_of[id(d)] for d in nodes[i].downstream_entities() if d is not None)
    return seen


def _check_keys(nodes, node_of, a) -> None:
    """A Random LoadBalancer takes its Requests from Sources that aim at it directly (their KEY draw is the choice; a Request
    without one would ask the process-wide `random`, which the engine's streams do not define).  A key-less Request at a
    ConsistentHash LoadBalancer is the reference's own case: the strategy's fallback RoundRobin (strategies.py:362,420-421)."""
    from .lowering import UnsupportedTopology

    rnd = [i for i, x in enumerate(nodes) if isinstance(x, LoadBalancer) and isinstance(x.strategy, Random)]
    if not rnd:
        return
    for src in nodes:
        if not isinstance(src, Source):
            continue
        tgt = src._event_provider._target
        reach = _reaches(nodes, node_of, tgt)
        for j in rnd:
            if j in reach and tgt is not nodes[j]:
                raise UnsupportedTopology(f"source '{src.name}' reaches the Random LoadBalancer '{nodes[j].name}' through other entities: "
                                          "only Sources that aim at it directly carry the draw it chooses by")


def keyless_hazard(g: "GeneralGraph", target) -> str | None:
    """Simulation.schedule(): a scheduled Request carries no client id -- the name of a Random LoadBalancer it could reach."""
    for j in _reaches(g.nodes, g.node_of, target):
        x = g.nodes[j]
        if isinstance(x, LoadBalancer) and isinstance(x.strategy, Random):
            return x.name
    return None


def write_back_general(g: GeneralGraph, stats: dict, rec_node: np.ndarray, rec_t: np.ndarray, rec_cr: np.ndarray, device: int = 0) -> None:
    """The run's per-node results onto the user's objects, under the attribute names the reference uses (lowering.write_back's
    counterpart)."""
    a = g.arrays
    order = np.argsort(rec_node, kind="stable")              # per Sink, still in processing order
    bounds = np.searchsorted(rec_node[order], np.arange(a.n + 1))
    for i, ent in enumerate(g.nodes):
        if isinstance(ent, Source):
            ent._generated_count = int(stats["generated"][i])
            ent._event_provider._generated = int(stats["payloads"][i])
        elif isinstance(ent, Server):
            ent._queue.stats_accepted = int(stats["accepted"][i])
            ent._queue.stats_dropped = int(stats["dropped"][i])
            ent._queue.depth = int(stats["queue_depth"][i])
            ent._requests_completed = int(stats["completed"][i])
            ent._requests_rejected = int(stats["rejected"][i])
            ent._total_service_time = float(stats["total_service_s"][i])
            ent._active = int(stats["active"][i])
        elif isinstance(ent, NetworkLink):
            ent.packets_sent = int(stats["packets_sent"][i])
            ent.packets_dropped = int(stats["packets_dropped"][i])
            ent._entered = int(stats["entered"][i])
        elif isinstance(ent, LoadBalancer):
            (ent._requests_received, ent._requests_forwarded, ent._requests_failed, ent._no_backend_available,
             ent._in_flight_count, selections) = (int(v) for v in stats["lb"][i])
            if isinstance(ent.strategy, RoundRobin):
                ent.strategy._index += selections                  # one select per forwarded Request (strategies.py:66-67)
            elif isinstance(ent.strategy, ConsistentHash):
                ent.strategy._fallback._index += selections        # ... per KEY-LESS Request (strategies.py:362,420-421)
            off = int(a.rt_off[i])
            for q, b in enumerate(ent.all_backends):
                ent._backends[b.name].total_requests = int(stats["rt_taken"][off + q])
        elif isinstance(ent, Probe):
            sel = order[bounds[i]:bounds[i + 1]]                 # its samples: (time, sampled integer)
            ent.data_sink._set(rec_t[sel].copy(), rec_cr[sel].copy(), Probe.value_map(ent.metric, ent.target))
        elif isinstance(ent, RandomRouter):
            ent.stats_routed = int(stats["routed"][i])
            tc: dict[str, int] = {}
            off = int(a.rt_off[i])
            for q, t in enumerate(ent.targets):               # target_counts[target.name] += 1 (random_router.py:37)
                c = int(stats["rt_taken"][off + q])
                if c:
                    tc[t.name] = tc.get(t.name, 0) + c
            ent.target_counts = tc
        else:
            sel = order[bounds[i]:bounds[i + 1]]
            ent._set_records(rec_t[sel].copy(), rec_cr[sel].copy())
            ent._device = device
