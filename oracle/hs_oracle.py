"""ctypes binding of the CPU oracle (oracle/libhs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhs_oracle.so")

SOURCE, SERVER, SINK, LINK, ROUTER, LB, PROBE = 0, 1, 2, 3, 4, 5, 6
M_DEPTH, M_ACTIVE, M_ACCEPTED, M_DROPPED, M_COMPLETED, M_RECEIVED, M_GENERATED = range(7)
ARR_POISSON, ARR_CONSTANT = 0, 1
LAT_EXP, LAT_CONST = 0, 1
RNG_PHILOX, RNG_MT19937 = 0, 1
PROF_CONSTANT, PROF_LINEAR_RAMP, PROF_SPIKE, PROF_GENERAL_CONSTANT = 0, 1, 2, 3
EV_KINDS = 15
EV_NAMES = ["source", "enqueue", "notify", "poll", "deliver", "work", "continuation", "sink", "link", "link_cont",
            "route", "lb", "lb_resp", "probe_tick", "probe"]
STREAM_ARRIVAL, STREAM_SERVICE, STREAM_LINK, STREAM_ROUTE, STREAM_KEY = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("hs_oracle.c", "hs_oracle.h", "hs_rng_ref.h")]
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Graph(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32),
        ("kind", C.POINTER(C.c_int32)),
        ("target", C.POINTER(C.c_int32)),
        ("stream_base", C.POINTER(C.c_uint64)),
        ("arr_kind", C.POINTER(C.c_int32)),
        ("rate", C.POINTER(C.c_double)),
        ("stop_after_ns", C.POINTER(C.c_int64)),
        ("concurrency", C.POINTER(C.c_int32)),
        ("lat_kind", C.POINTER(C.c_int32)),
        ("lat_mean", C.POINTER(C.c_double)),
        ("lat_min", C.POINTER(C.c_double)),
        ("queue_cap", C.POINTER(C.c_int64)),
        ("rt_off", C.POINTER(C.c_int32)),
        ("rt_cnt", C.POINTER(C.c_int32)),
        ("rt_targets", C.POINTER(C.c_int32)),
        ("n_rt", C.c_int32),
        ("n_clients", C.POINTER(C.c_int64)),
        ("vnodes", C.POINTER(C.c_int32)),
        ("names", C.c_char_p),
        ("name_off", C.POINTER(C.c_int32)),
        ("prof_kind", C.POINTER(C.c_int32)),
        ("prof_p", C.POINTER(C.c_double)),
        ("probe_metric", C.POINTER(C.c_int32)),
        ("loss", C.POINTER(C.c_double)),
        ("ploss", C.POINTER(C.c_double)),
    ]


class _Params(C.Structure):
    _fields_ = [
        ("start_ns", C.c_int64),
        ("end_ns", C.c_int64),
        ("seed", C.c_uint64),
        ("rng_mode", C.c_int32),
        ("mt_seed_py", C.c_uint32),
        ("mt_seed_np", C.c_uint32),
        ("trace_cap", C.c_int64),
        ("coord_seed", C.c_uint32),
    ]


class _Summary(C.Structure):
    _fields_ = [
        ("events_processed", C.c_int64),
        ("events_by_kind", C.c_int64 * EV_KINDS),
        ("final_time_ns", C.c_int64),
        ("heap_peak", C.c_int64),
        ("sort_index_next", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.hso_create.restype = C.c_void_p
        L.hso_create.argtypes = [C.POINTER(_Graph), C.POINTER(_Params)]
        L.hso_schedule.restype = C.c_int
        L.hso_schedule.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.hso_run_until.restype = C.c_int
        L.hso_run_until.argtypes = [C.c_void_p, C.c_int64]
        L.hso_get_summary.argtypes = [C.c_void_p, C.POINTER(_Summary)]
        L.hso_get_node_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.hso_get_net_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hso_get_lb_stats.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hso_lb_select.restype = C.c_int32
        L.hso_lb_select.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
        L.hso_md5.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        L.hso_profile_next_arrival.restype = C.c_int64
        L.hso_profile_next_arrival.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_double]
        L.hso_sink_count.restype = C.c_int64
        L.hso_sink_count.argtypes = [C.c_void_p, C.c_int32]
        L.hso_read_sink.restype = C.c_int64
        L.hso_read_sink.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
        L.hso_read_trace.restype = C.c_int64
        L.hso_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.hso_destroy.argtypes = [C.c_void_p]
        L.hso_uniform.restype = C.c_double
        L.hso_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.hso_log.restype = C.c_double
        L.hso_log.argtypes = [C.c_double]
        L.hso_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hso_mt_py_random.restype = C.c_double
        L.hso_mt_py_random.argtypes = [C.c_uint32, C.c_int64]
        L.hso_mt_np_random.restype = C.c_double
        L.hso_mt_np_random.argtypes = [C.c_uint32, C.c_int64]
        _lib = L
    return _lib


@dataclass
class Graph:
    """Node-array description of a lowered entity graph (oracle side)."""

    kind: list = field(default_factory=list)
    target: list = field(default_factory=list)
    stream_base: list = field(default_factory=list)
    arr_kind: list = field(default_factory=list)
    rate: list = field(default_factory=list)
    stop_after_ns: list = field(default_factory=list)
    concurrency: list = field(default_factory=list)
    lat_kind: list = field(default_factory=list)
    lat_mean: list = field(default_factory=list)
    lat_min: list = field(default_factory=list)
    queue_cap: list = field(default_factory=list)
    rt_off: list = field(default_factory=list)
    rt_cnt: list = field(default_factory=list)
    rt_targets: list = field(default_factory=list)   # flat target list of all routers / load balancers (not per node)
    n_clients: list = field(default_factory=list)
    vnodes: list = field(default_factory=list)
    names: list = field(default_factory=list)        # entity names (only an LB's backends need theirs)
    prof_kind: list = field(default_factory=list)
    prof_p: list = field(default_factory=list)       # 4 parameters per node
    probe_metric: list = field(default_factory=list)
    loss: list = field(default_factory=list)
    ploss: list = field(default_factory=list)        # PartitionLink.packet_loss of the partition pair a link crosses (coord_seed)

    def _add(self, **kw) -> int:
        defaults = dict(
            kind=0, target=-1, stream_base=len(self.kind), arr_kind=0, rate=0.0, stop_after_ns=-1,
            concurrency=1, lat_kind=LAT_CONST, lat_mean=0.0, lat_min=0.0, queue_cap=-1, rt_off=0, rt_cnt=0,
            n_clients=0, vnodes=0, names="", prof_kind=PROF_CONSTANT, prof_p=(0.0, 0.0, 0.0, 0.0), probe_metric=0, loss=0.0,
            ploss=0.0,
        )
        defaults.update(kw)
        for k, v in defaults.items():
            getattr(self, k).append(v)
        return len(self.kind) - 1

    def source(self, arr_kind, rate, target=-1, stop_after_ns=-1, stream_base=None, n_clients=0, profile=None) -> int:
        """profile: None (ConstantRateProfile(rate)) | ("ramp", duration_s, start_rate, end_rate) |
        ("spike", baseline_rate, spike_rate, warmup_s, spike_duration_s)."""
        kw = dict(kind=SOURCE, arr_kind=arr_kind, rate=float(rate), target=target, stop_after_ns=stop_after_ns,
                  n_clients=int(n_clients))
        if profile is not None:
            kw["prof_kind"] = PROF_LINEAR_RAMP if profile[0] == "ramp" else PROF_SPIKE
            kw["prof_p"] = tuple(float(x) for x in profile[1:]) + (0.0,) * (5 - len(profile))
        if stream_base is not None:
            kw["stream_base"] = stream_base
        return self._add(**kw)

    def server(self, lat_kind, lat_mean, concurrency=1, queue_cap=-1, target=-1, stream_base=None, name="") -> int:
        kw = dict(kind=SERVER, lat_kind=lat_kind, lat_mean=float(lat_mean), concurrency=concurrency,
                  queue_cap=queue_cap, target=target, names=name)
        if stream_base is not None:
            kw["stream_base"] = stream_base
        return self._add(**kw)

    def sink(self) -> int:
        return self._add(kind=SINK)

    def probe(self, target: int, metric: int, interval: float) -> int:
        """Probe(target, metric, data, interval): add it AFTER every source (probes start after sources)."""
        return self._add(kind=PROBE, target=target, arr_kind=ARR_CONSTANT, probe_metric=metric,
                         prof_kind=PROF_GENERAL_CONSTANT, prof_p=(1.0 / interval, 0.0, 0.0, 0.0))

    def link(self, lat_min, jitter_mean=None, target=-1, stream_base=None, loss=0.0, jitter_kind="exp", ploss=0.0) -> int:
        """NetworkLink(latency=ConstantLatency(lat_min), jitter=ExponentialLatency(jitter_mean) | ConstantLatency(jitter_mean)
        (jitter_kind="const") | None, egress=target, packet_loss_rate=loss); ploss: the PartitionLink.packet_loss of the partition
        pair the link crosses (drawn from the coordinator's generator, `run(coord_seed=)`)."""
        kw = dict(kind=LINK, lat_min=float(lat_min), target=target, loss=float(loss), ploss=float(ploss),
                  lat_kind=LAT_CONST if (jitter_mean is None or jitter_kind == "const") else LAT_EXP,
                  lat_mean=0.0 if jitter_mean is None else float(jitter_mean))
        if stream_base is not None:
            kw["stream_base"] = stream_base
        return self._add(**kw)

    def router(self, targets, stream_base=None) -> int:
        """RandomRouter(targets=[...]) with the Philox-plugged uniform choice."""
        kw = dict(kind=ROUTER, rt_off=len(self.rt_targets), rt_cnt=len(targets))
        if stream_base is not None:
            kw["stream_base"] = stream_base
        self.rt_targets.extend(int(t) for t in targets)
        return self._add(**kw)

    def load_balancer(self, backends, vnodes, stream_base=None) -> int:
        """LoadBalancer(backends=[...], strategy=ConsistentHash(virtual_nodes=vnodes)); backends need names."""
        kw = dict(kind=LB, rt_off=len(self.rt_targets), rt_cnt=len(backends), vnodes=int(vnodes))
        if stream_base is not None:
            kw["stream_base"] = stream_base
        self.rt_targets.extend(int(t) for t in backends)
        return self._add(**kw)

    def __len__(self) -> int:
        return len(self.kind)


def mm1_chains(n: int, rate=8.0, mean=0.1, arr_kind=ARR_POISSON, lat_kind=LAT_EXP, concurrency=1,
               queue_cap=-1, stop_after_ns=-1, stream_base0: int = 0) -> Graph:
    """n independent Source -> Server -> Sink chains; chain i uses stream base stream_base0 + i
    for BOTH its source (arrival stream) and its server (service stream).

    Node order: all sources (list order = chain order), then per chain server, sink."""
    g = Graph()
    srcs = [g.source(arr_kind, rate, stream_base=stream_base0 + i, stop_after_ns=stop_after_ns) for i in range(n)]
    for i in range(n):
        sv = g.server(lat_kind, mean, concurrency=concurrency, queue_cap=queue_cap, stream_base=stream_base0 + i)
        sk = g.sink()
        g.target[srcs[i]] = sv
        g.target[sv] = sk
    return g


class Result:
    pass


def _create(g: Graph, end_ns: int, start_ns: int, seed: int, rng_mode: int, mt_seed_py: int, mt_seed_np: int,
            trace_cap: int, coord_seed: int = 42):
    """hso_create for a Graph; returns the handle (the C side copies every array)."""
    L = lib()
    n = len(g)
    arrs = {
        "kind": np.asarray(g.kind, np.int32), "target": np.asarray(g.target, np.int32),
        "stream_base": np.asarray(g.stream_base, np.uint64), "arr_kind": np.asarray(g.arr_kind, np.int32),
        "rate": np.asarray(g.rate, np.float64), "stop_after_ns": np.asarray(g.stop_after_ns, np.int64),
        "concurrency": np.asarray(g.concurrency, np.int32), "lat_kind": np.asarray(g.lat_kind, np.int32),
        "lat_mean": np.asarray(g.lat_mean, np.float64), "lat_min": np.asarray(g.lat_min, np.float64),
        "queue_cap": np.asarray(g.queue_cap, np.int64), "rt_off": np.asarray(g.rt_off, np.int32),
        "rt_cnt": np.asarray(g.rt_cnt, np.int32),
        "rt_targets": np.asarray(g.rt_targets if g.rt_targets else [0], np.int32),
        "n_clients": np.asarray(g.n_clients, np.int64), "vnodes": np.asarray(g.vnodes, np.int32),
        "prof_kind": np.asarray(g.prof_kind, np.int32),
        "prof_p": np.asarray(g.prof_p, np.float64).reshape(-1),
        "probe_metric": np.asarray(g.probe_metric, np.int32),
        "loss": np.asarray(g.loss, np.float64),
        "ploss": np.asarray(g.ploss, np.float64),
    }
    enc = [nm.encode() for nm in g.names]
    names_blob = b"".join(enc) + b"\0"
    name_off = np.zeros(n + 1, np.int32)
    name_off[1:] = np.cumsum([len(b) for b in enc])
    G = _Graph()
    G.n_nodes = n
    G.n_rt = len(g.rt_targets)
    for name, a in arrs.items():
        ftype = dict(_Graph._fields_)[name]
        setattr(G, name, a.ctypes.data_as(ftype))
    G.names = names_blob
    G.name_off = name_off.ctypes.data_as(C.POINTER(C.c_int32))
    P = _Params(start_ns, end_ns, seed, rng_mode, mt_seed_py, mt_seed_np, trace_cap, coord_seed)
    return L.hso_create(C.byref(G), C.byref(P))


def run(g: Graph, end_ns: int, start_ns: int = 0, seed: int = 42, rng_mode: int = RNG_PHILOX,
        mt_seed_py: int = 42, mt_seed_np: int = 42, trace_cap: int = 0, windows: list | None = None,
        lb_probe: int = 0, schedule: list | None = None, coord_seed: int = 42) -> Result:
    """Run the oracle once; returns a Result with summary, per-node stats, sink records, trace.
    schedule: [(node, time_ns), ...] = Simulation.schedule(Event(time, "Request", target=node)) calls before run()."""
    L = lib()
    n = len(g)
    import time as _time

    h = _create(g, end_ns, start_ns, seed, rng_mode, mt_seed_py, mt_seed_np, trace_cap, coord_seed)
    try:
        for node, t_ns in (schedule or []):
            if L.hso_schedule(h, int(node), int(t_ns)) != 0:
                raise ValueError(f"oracle: node {node} takes no scheduled Request")
        t0 = _time.perf_counter()
        for w_end in (windows or []):
            L.hso_run_until(h, int(w_end))
        rc = L.hso_run_until(h, int(end_ns))
        run_seconds = _time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("oracle: unsupported event kind")
        S = _Summary()
        L.hso_get_summary(h, C.byref(S))
        r = Result()
        r.run_seconds = run_seconds  # wall time of the event loop only (graph build excluded)
        r.events_processed = S.events_processed
        r.events_by_kind = np.array(list(S.events_by_kind), np.int64)
        r.final_time_ns = S.final_time_ns
        r.heap_peak = S.heap_peak
        names = ["generated", "accepted", "dropped", "completed", "rejected", "total_service_s",
                 "received", "depth", "active"]
        bufs = {nm: np.zeros(n, np.float64 if nm == "total_service_s" else np.int64) for nm in names}
        L.hso_get_node_stats(h, *[bufs[nm].ctypes.data for nm in names])
        for nm in names:
            setattr(r, nm, bufs[nm])
        r.packets_sent = np.zeros(n, np.int64)
        r.routed = np.zeros(n, np.int64)
        L.hso_get_net_stats(h, r.packets_sent.ctypes.data, r.routed.ctypes.data)
        r.lbs = {}
        for i in range(n):
            if g.kind[i] == LB:
                st = np.zeros(6, np.int64)
                tot = np.zeros(max(g.rt_cnt[i], 1), np.int64)
                ring = np.zeros(max(g.rt_cnt[i] * g.vnodes[i], 1), np.int32)
                L.hso_get_lb_stats(h, i, st.ctypes.data, tot.ctypes.data, ring.ctypes.data)
                r.lbs[i] = dict(stats=st[:5], strategy_index=int(st[5]), total_requests=tot[:g.rt_cnt[i]], ring_backend=ring[:g.rt_cnt[i] * g.vnodes[i]],
                                select=[L.hso_lb_select(h, i, str(c).encode()) for c in range(lb_probe)])
        r.sinks = {}
        for i in range(n):
            if g.kind[i] in (SINK, PROBE):        # a probe's samples come back as (sample ns, value)
                c = L.hso_sink_count(h, i)
                t = np.zeros(c, np.int64)
                cr = np.zeros(c, np.int64)
                L.hso_read_sink(h, i, t.ctypes.data, cr.ctypes.data, c)
                r.sinks[i] = (t, cr)
        if trace_cap:
            t = np.zeros(trace_cap, np.int64); k = np.zeros(trace_cap, np.int32)
            nd = np.zeros(trace_cap, np.int32); ix = np.zeros(trace_cap, np.int64)
            m = L.hso_read_trace(h, t.ctypes.data, k.ctypes.data, nd.ctypes.data, ix.ctypes.data, trace_cap)
            r.trace = (t[:m], k[:m], nd[:m], ix[:m])
        return r
    finally:
        L.hso_destroy(h)


def run_blocks_parallel(graphs: list, end_ns: int, seed: int = 42) -> dict:
    """Independent graphs (blocks of chains: what the reference's ParallelRunner does with processes,
    parallel/runner.py:43-142), each in its own oracle instance on its own host thread -- ctypes drops the GIL for the
    call, so the event loops really run side by side.  Instances are created first; the clock covers the event loops
    only (first start to last finish).  Returns dict(events, wall_seconds, threads)."""
    import threading
    import time as _time

    L = lib()
    hs = [_create(g, end_ns, 0, seed, RNG_PHILOX, 42, 42, 0) for g in graphs]
    gate = threading.Barrier(len(hs) + 1)
    rcs = [0] * len(hs)

    def work(i):
        gate.wait()
        rcs[i] = L.hso_run_until(hs[i], int(end_ns))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(hs))]
    for t in ts:
        t.start()
    gate.wait()
    t0 = _time.perf_counter()
    for t in ts:
        t.join()
    wall = _time.perf_counter() - t0
    events = 0
    try:
        if any(rcs):
            raise RuntimeError("oracle: unsupported event kind")
        for h in hs:
            S = _Summary()
            L.hso_get_summary(h, C.byref(S))
            events += S.events_processed
    finally:
        for h in hs:
            L.hso_destroy(h)
    return dict(events=int(events), wall_seconds=wall, threads=len(hs))


def lb_topology(n_sources, n_backends, rate, mean, vnodes, n_clients, concurrency=1, queue_cap=-1,
                stop_after_ns=-1, shared_sink=True, strategy="chash") -> Graph:
    """S sources (Poisson, client ids from their KEY stream) -> LoadBalancer(ConsistentHash(vnodes)) -> B Server backends
    -> one shared Sink or one per backend (tests/golden/make_golden.py run_lb_case).  Node order: sources 0..S-1,
    LB = S, backends S+1..S+B, sinks after; stream bases: source i -> i, backend j -> S + j; names "srv<j>"."""
    def per(v, m):
        return list(v) if isinstance(v, (list, tuple)) else [v] * m
    S, B = n_sources, n_backends
    g = Graph()
    rate, mean, conc, qcap = per(rate, S), per(mean, B), per(concurrency, B), per(queue_cap, B)
    # strategy (hs_oracle.c on_lb): "chash" ConsistentHash(vnodes); "round_robin": vnodes field 0; "random": -1, and every Request
    # draws its backend index int(u * B) from its Source's KEY stream
    if strategy == "round_robin":
        vnodes = 0
    elif strategy == "random":
        vnodes, n_clients = -1, B
    for i in range(S):
        g.source(ARR_POISSON, rate[i], target=S, stop_after_ns=stop_after_ns, stream_base=i, n_clients=n_clients)
    lb = g._add(kind=LB)          # placeholder, filled in below once the backend nodes exist
    assert lb == S
    bes = [g.server(LAT_EXP, mean[j], concurrency=conc[j], queue_cap=-1 if qcap[j] is None else qcap[j],
                    stream_base=S + j, name=f"srv{j}") for j in range(B)]
    sinks = [g.sink() for _ in range(1 if shared_sink else B)]
    for j, b in enumerate(bes):
        g.target[b] = sinks[0] if shared_sink else sinks[j]
    g.rt_off[lb] = len(g.rt_targets)
    g.rt_cnt[lb] = B
    g.vnodes[lb] = int(vnodes)
    g.rt_targets.extend(bes)
    return g


def profile_next_arrival(profile, t_start_ns: int, target_area: float) -> int:
    """ArrivalTimeProvider.next_arrival_time (general path) for ("ramp", ...) / ("spike", ...) profiles; -1 = raises."""
    kind = PROF_LINEAR_RAMP if profile[0] == "ramp" else PROF_SPIKE
    p = (C.c_double * 4)(*(tuple(float(x) for x in profile[1:]) + (0.0,) * (5 - len(profile))))
    return int(lib().hso_profile_next_arrival(kind, p, int(t_start_ns), float(target_area)))


def md5(data: bytes) -> bytes:
    out = (C.c_uint8 * 16)()
    lib().hso_md5(data, len(data), out)
    return bytes(out)


def uniform(seed: int, sid: int, k: int) -> float:
    return lib().hso_uniform(seed, sid, k)


def log(x: float) -> float:
    return lib().hso_log(x)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().hso_philox(c, k, o)
    return list(o)
