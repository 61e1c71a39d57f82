"""Thin object wrapper over the C ABI: one `StationEngine` = one `hs_engine` handle on one GPU.

Host-side plumbing only (numpy buffers in, numpy buffers out).  All simulation
work happens in libhs_hip.so; nothing here computes event logic.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
import os
from dataclasses import dataclass

import numpy as np

from . import _native as N


@dataclass
class StationArrays:
    """Struct-of-arrays description of the station LPs (see include/hs_engine.h `hs_stations`)."""

    n: int
    src_kind: np.ndarray
    src_rate: np.ndarray
    src_stop_after_ns: np.ndarray
    concurrency: np.ndarray
    svc_kind: np.ndarray
    svc_mean_s: np.ndarray
    queue_cap: np.ndarray
    egress: np.ndarray
    seed: np.ndarray | None = None
    stream_base: np.ndarray | None = None
    src_profile_kind: np.ndarray | None = None      # N.PROF_*; None = constant rate everywhere
    src_profile_params: np.ndarray | None = None    # [n, 4]
    probe_metric: np.ndarray | None = None          # N.PROBE_METRICS ids, N.PROBE_NONE = no probe on the LP
    probe_interval_s: np.ndarray | None = None
    # Simulation.schedule(): Requests injected before run(); station i gets sched_time_ns[sched_off[i]:sched_off[i + 1]]
    sched_off: np.ndarray | None = None             # [n + 1] int64
    sched_time_ns: np.ndarray | None = None         # ascending per station, ties in the caller's order
    # construction order of the reference's pre-run events (include/hs_engine.h): None = LP order / array order
    source_order: np.ndarray | None = None          # LP indices of the Sources in `sources=[...]` order
    probe_order: np.ndarray | None = None           # LP indices of the Probes in `probes=[...]` order
    probe_metric_more: np.ndarray | None = None     # [3, n] further probes of an LP (slots 1..3; N.PROBE_NONE = none)
    probe_interval_more: np.ndarray | None = None   # [3, n]
    probe_slot_order: np.ndarray | None = None      # slot of every entry of probe_order (None = 0 everywhere)
    sched_rank: np.ndarray | None = None            # like sched_time_ns: the Event's position among all constructed Events
    # several Sources feeding one Server: slots 1..3 of an LP (slot 0 = src_kind / src_rate / src_stop_after_ns)
    src_more_kind: np.ndarray | None = None         # [3, n] N.SRC_*; N.SRC_NONE = none
    src_more_rate: np.ndarray | None = None         # [3, n]
    src_more_stop_after_ns: np.ndarray | None = None  # [3, n]; None = never
    source_slot_order: np.ndarray | None = None     # slot of every entry of source_order (None = 0 everywhere)
    # tandem queues, Server(downstream=<Server>): egress[i] == N.EGRESS_SERVER forwards station i's completions to the Server of
    # station downstream_lp[i] (include/hs_engine.h)
    downstream_lp: np.ndarray | None = None         # [n] int32

    @staticmethod
    def uniform(n: int, *, src_kind=N.SRC_POISSON, rate=8.0, stop_after_ns=-1, concurrency=1,
                svc_kind=N.LAT_EXPONENTIAL, mean=0.1, queue_cap=-1, egress=N.EGRESS_SINK) -> "StationArrays":
        return StationArrays(
            n=n,
            src_kind=np.full(n, src_kind, np.uint8), src_rate=np.full(n, rate, np.float64),
            src_stop_after_ns=np.full(n, stop_after_ns, np.int64), concurrency=np.full(n, concurrency, np.int32),
            svc_kind=np.full(n, svc_kind, np.uint8), svc_mean_s=np.full(n, mean, np.float64),
            queue_cap=np.full(n, queue_cap, np.int64), egress=np.full(n, egress, np.uint8),
        )


@dataclass
class NetworkArrays:
    """Links / routers between stations (see include/hs_engine.h `hs_network`)."""

    egress_kind: np.ndarray            # [n] N.EGRESS_*
    router_target0: np.ndarray         # [n] -1 = the station's Sink, >= 0 = link index
    router_target1: np.ndarray
    link_of: np.ndarray                # [n] link index for EGRESS_LINK
    link_src: np.ndarray               # [n_links]
    link_dst: np.ndarray
    link_lat_min_s: np.ndarray
    link_jitter_kind: np.ndarray       # N.LAT_EXPONENTIAL (jitter) | N.LAT_CONSTANT (none)
    link_jitter_mean_s: np.ndarray
    router_stream_base: np.ndarray | None = None
    link_stream_base: np.ndarray | None = None
    link_loss_rate: np.ndarray | None = None   # [n_links] NetworkLink.packet_loss_rate; None = lossless
    router_n_targets: np.ndarray | None = None # [n] len(RandomRouter.targets), 1..4; None = 2 everywhere
    router_target2: np.ndarray | None = None   # [n] third / fourth target of routers with more than two
    router_target3: np.ndarray | None = None
    link_drop_capacity: np.ndarray | None = None  # [n_links] packets of a table-decided loss (PartitionLink.packet_loss); None / 0 = none
    bag_capacity: int = 0
    # one shard of a partitioned network (happy_simulator_amd/sharded.py): network-wide endpoints and link ids
    n_global_lp: int = 0
    link_gid: np.ndarray | None = None
    n_global_links: int = 0

    @property
    def n_links(self) -> int:
        return int(len(self.link_dst))


# Engines that are still open when the interpreter exits (a Simulation keeps its engine while Sink records / Probe samples are on
# the device: lowering.LazyRecords) are closed BEFORE Python tears its modules down -- destroying one from a late __del__, after
# the HIP runtime has started to unload, aborts the process in free().
_LIVE: "weakref.WeakSet" = weakref.WeakSet()


def _close_live_engines():
    for e in list(_LIVE):
        try:
            e.close()
        except Exception:
            pass


atexit.register(_close_live_engines)


class EngineSummary:
    def __init__(self, s: N.Summary):
        self.events_processed = int(s.events_processed)
        self.events_by_kind = np.array(list(s.events_by_kind), np.int64)
        self.events_cancelled = int(s.events_cancelled)
        self.final_time_ns = int(s.final_time_ns)
        self.requests_completed = int(s.requests_completed)
        self.sink_records = int(s.sink_records)
        self.last_run_ms = float(s.last_run_ms)
        self.kernel_ms = float(s.kernel_ms)
        self.launches = int(s.launches)
        self.window_ns = int(s.window_ns)
        self.overflow = bool(s.overflow)


class StationEngine:
    """GPU-resident engine for `n` station LPs on one device."""

    def __init__(self, stations: StationArrays, *, mode: int, horizon_ns: int, start_ns: int = 0, seed: int = 42,
                 lp_base: int = 0, device: int = 0, log_capacity: int = 0, network: "NetworkArrays | None" = None):
        self._lib = N.lib()
        if self._lib.hs_device_count() <= 0:
            raise N.EngineUnavailable("no HIP device visible: the engine has no CPU fallback")
        self.n = int(stations.n)
        self.mode = mode
        self._h = C.c_void_p()
        cfg = N.Config(C.sizeof(N.Config), device, self.n, mode, start_ns, horizon_ns, seed, lp_base, log_capacity)
        self._check(self._lib.hs_engine_create(C.byref(cfg), C.byref(self._h)), create=True)
        st = N.Stations()
        keep = []
        for name, dtype in (("src_kind", np.uint8), ("src_rate", np.float64), ("src_stop_after_ns", np.int64),
                            ("concurrency", np.int32), ("svc_kind", np.uint8), ("svc_mean_s", np.float64),
                            ("queue_cap", np.int64), ("egress", np.uint8), ("seed", np.uint64),
                            ("stream_base", np.uint64), ("src_profile_kind", np.uint8),
                            ("src_profile_params", np.float64), ("probe_metric", np.uint8),
                            ("probe_interval_s", np.float64)):
            a = getattr(stations, name)
            if a is None:
                setattr(st, name, None)
                continue
            a = np.ascontiguousarray(a, dtype)
            if a.shape != ((self.n, 4) if name == "src_profile_params" else (self.n,)):
                raise ValueError(f"{name} has the wrong shape for {self.n} stations")
            keep.append(a)
            setattr(st, name, a.ctypes.data)
        if stations.sched_off is not None:
            off = np.ascontiguousarray(stations.sched_off, np.int64)
            if off.shape != (self.n + 1,):
                raise ValueError(f"sched_off must have shape ({self.n + 1},)")
            tt = np.ascontiguousarray(stations.sched_time_ns if stations.sched_time_ns is not None else [], np.int64)
            if tt.shape != (int(off[-1]),):
                raise ValueError("sched_time_ns must hold sched_off[-1] times")
            keep += [off, tt]
            st.sched_off, st.sched_time_ns = off.ctypes.data, (tt.ctypes.data if len(tt) else None)
            if stations.sched_rank is not None:
                co = np.ascontiguousarray(stations.sched_rank, np.int64)
                if co.shape != tt.shape or (len(co) and co.min() < 0):
                    raise ValueError("sched_rank must hold one position >= 0 per scheduled time")
                keep.append(co)
                st.sched_rank = co.ctypes.data if len(co) else None
        if stations.src_more_kind is not None:
            mk = np.ascontiguousarray(stations.src_more_kind, np.uint8)
            mr = np.ascontiguousarray(stations.src_more_rate, np.float64)
            if mk.shape != (3, self.n) or mr.shape != (3, self.n):
                raise ValueError("src_more_kind / src_more_rate must have shape (3, n)")
            keep += [mk, mr]
            st.src_more_kind, st.src_more_rate = mk.ctypes.data, mr.ctypes.data
            if stations.src_more_stop_after_ns is not None:
                ms = np.ascontiguousarray(stations.src_more_stop_after_ns, np.int64)
                if ms.shape != (3, self.n):
                    raise ValueError("src_more_stop_after_ns must have shape (3, n)")
                keep.append(ms)
                st.src_more_stop_after_ns = ms.ctypes.data
        if stations.source_order is not None:       # (the engine checks that every (LP, slot) Source is listed exactly once)
            a = np.ascontiguousarray(stations.source_order, np.int32)
            if stations.src_more_kind is None and \
                    sorted(a.tolist()) != np.flatnonzero(np.asarray(stations.src_kind) != N.SRC_NONE).tolist():
                raise ValueError("source_order must list exactly the LPs that carry a Source, each once")
            # the engine reads one entry per Source from this pointer: the length must be the number of Sources
            n_src = int(np.count_nonzero(np.asarray(stations.src_kind) != N.SRC_NONE))
            if stations.src_more_kind is not None:
                n_src += int(np.count_nonzero(np.asarray(stations.src_more_kind) != N.SRC_NONE))
            if len(a) != n_src:
                raise ValueError(f"source_order must have one entry per Source ({n_src}), got {len(a)}")
            keep.append(a)
            st.source_order = a.ctypes.data if len(a) else None
            if stations.source_slot_order is not None:
                b = np.ascontiguousarray(stations.source_slot_order, np.uint8)
                if b.shape != a.shape:
                    raise ValueError("source_slot_order must have one entry per entry of source_order")
                keep.append(b)
                st.source_slot_order = b.ctypes.data if len(b) else None
        if stations.probe_metric_more is not None:
            pm = np.ascontiguousarray(stations.probe_metric_more, np.uint8)
            pi = np.ascontiguousarray(stations.probe_interval_more, np.float64)
            if pm.shape != (3, self.n) or pi.shape != (3, self.n):
                raise ValueError("probe_metric_more / probe_interval_more must have shape (3, n)")
            keep += [pm, pi]
            st.probe_metric_more, st.probe_interval_more = pm.ctypes.data, pi.ctypes.data
        if stations.probe_order is not None:        # (the engine checks that every (LP, slot) probe is listed exactly once)
            a = np.ascontiguousarray(stations.probe_order, np.int32)
            n_prb = 0 if stations.probe_metric is None else int(np.count_nonzero(np.asarray(stations.probe_metric) != N.PROBE_NONE))
            if stations.probe_metric_more is not None:
                n_prb += int(np.count_nonzero(np.asarray(stations.probe_metric_more) != N.PROBE_NONE))
            if len(a) != n_prb:                     # the engine reads one entry per Probe from this pointer
                raise ValueError(f"probe_order must have one entry per Probe ({n_prb}), got {len(a)}")
            keep.append(a)
            st.probe_order = a.ctypes.data if len(a) else None
            if stations.probe_slot_order is not None:
                b = np.ascontiguousarray(stations.probe_slot_order, np.uint8)
                if b.shape != a.shape:
                    raise ValueError("probe_slot_order must have one entry per entry of probe_order")
                keep.append(b)
                st.probe_slot_order = b.ctypes.data if len(b) else None
        self._tandem = stations.downstream_lp is not None
        if stations.downstream_lp is not None:
            a = np.ascontiguousarray(stations.downstream_lp, np.int32)
            if a.shape != (self.n,):
                raise ValueError(f"downstream_lp has the wrong shape for {self.n} stations")
            keep.append(a)
            st.downstream_lp = a.ctypes.data
        self.n_links = 0
        try:
            self._check(self._lib.hs_engine_set_stations(self._h, C.byref(st)))
            if os.environ.get("HS_PROF_BUDGET_LOG2"):
                self.set_profile_budget(1 << int(os.environ["HS_PROF_BUDGET_LOG2"]))
            if network is not None:
                self._set_network(network)
        except Exception:
            self.close()
            raise
        _LIVE.add(self)

    def _set_network(self, net: "NetworkArrays"):
        nw = N.Network()
        keep = []

        def put(name, arr, dtype, length):
            if arr is None:
                setattr(nw, name, None)
                return
            a = np.ascontiguousarray(arr, dtype)
            if a.shape != (length,):
                raise ValueError(f"{name} must have shape ({length},)")
            keep.append(a)
            setattr(nw, name, a.ctypes.data if length else None)

        nl = net.n_links
        put("egress_kind", net.egress_kind, np.uint8, self.n)
        put("router_target0", net.router_target0, np.int32, self.n)
        put("router_target1", net.router_target1, np.int32, self.n)
        put("link_of", net.link_of, np.int32, self.n)
        put("router_stream_base", net.router_stream_base, np.uint64, self.n)
        nw.n_links = nl
        put("link_dst", net.link_dst, np.int32, nl)
        put("link_src", net.link_src, np.int32, nl)
        put("link_lat_min_s", net.link_lat_min_s, np.float64, nl)
        put("link_jitter_kind", net.link_jitter_kind, np.uint8, nl)
        put("link_jitter_mean_s", net.link_jitter_mean_s, np.float64, nl)
        put("link_stream_base", net.link_stream_base, np.uint64, nl)
        put("link_loss_rate", net.link_loss_rate, np.float64, nl)
        put("router_n_targets", net.router_n_targets, np.uint8, self.n)
        put("router_target2", net.router_target2, np.int32, self.n)
        put("router_target3", net.router_target3, np.int32, self.n)
        put("link_drop_capacity", net.link_drop_capacity, np.int64, nl)
        nw.bag_capacity = int(net.bag_capacity)
        nw.n_global_lp = int(net.n_global_lp)
        put("link_gid", net.link_gid, np.int64, nl)
        nw.n_global_links = int(net.n_global_links)
        self._check(self._lib.hs_engine_set_network(self._h, C.byref(nw)))
        self.n_links = nl

    def set_link_drops(self, link: int, drops: np.ndarray):
        """Packet number e of `link` (a local link index with a loss table, NetworkArrays.link_drop_capacity) is lost iff drops[e]."""
        bits = np.packbits(np.asarray(drops, bool), bitorder="little")
        words = np.zeros((len(bits) + 3) // 4 * 4, np.uint8)
        words[:len(bits)] = bits
        w = np.ascontiguousarray(words.view(np.uint32))
        self._check(self._lib.hs_engine_set_link_drops(self._h, int(link), w.ctypes.data if len(w) else None, int(len(drops))))

    def send_log(self) -> np.ndarray:
        """[k, 3] int64 {send time ns, network-wide link id, packet number} of every packet that entered a table-decided link since
        the last reset (no particular order)."""
        n = int(self._check(self._lib.hs_engine_read_send_log(self._h, None, 0)))
        out = np.zeros((max(n, 1), 3), np.int64)
        if n:
            self._check(self._lib.hs_engine_read_send_log(self._h, out.ctypes.data, n))
        return out[:n]

    def net_stats(self) -> dict:
        out = {"routed": np.zeros(self.n, np.int64), "link_entered": np.zeros(max(self.n_links, 1), np.int64),
               "link_packets_sent": np.zeros(max(self.n_links, 1), np.int64),
               "link_packets_dropped": np.zeros(max(self.n_links, 1), np.int64)}
        st = N.NetStats(**{k: v.ctypes.data for k, v in out.items()})
        self._check(self._lib.hs_engine_get_net_stats(self._h, C.byref(st)))
        for k in ("link_entered", "link_packets_sent", "link_packets_dropped"):
            out[k] = out[k][:self.n_links]
        return out

    # -- error plumbing ------------------------------------------------------------------------
    def _check(self, rc: int, create: bool = False):
        if rc >= 0:
            return rc
        msg = (self._lib.hs_last_global_error() if create or not self._h else self._lib.hs_last_error(self._h))
        msg = msg.decode() if msg else ""
        if rc == N.HS_E_NO_DEVICE:
            raise N.EngineUnavailable(msg)
        if rc == N.HS_E_INVALID:
            raise ValueError(msg)
        raise N.EngineError(rc, msg)

    # -- run control ---------------------------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        """Enqueue on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); None = own stream."""
        if hip_stream is None:
            self._check(self._lib.hs_engine_set_stream(self._h, None, 0))
        else:
            self._check(self._lib.hs_engine_set_stream(self._h, C.c_void_p(int(hip_stream)), 1))

    def reset(self):
        self._check(self._lib.hs_engine_reset(self._h))

    def run_until(self, end_ns: int):
        self._check(self._lib.hs_engine_run_until(self._h, int(end_ns)))

    def run_until_async(self, end_ns: int):
        self._check(self._lib.hs_engine_run_until_async(self._h, int(end_ns)))

    def tandem_path(self) -> int:
        """0: no Server forwards to a Server; 1: passes of the station kernel; 2: the single-heap loop (include/hs_engine.h)."""
        return int(self._lib.hs_engine_tandem_path(self._h))

    def prologue_path(self) -> int:
        """0: no pre-run events that need the prologue; 1: the prologue was skipped (no pre-run event shared its nanosecond with
        another event of its LP); 2: the run went through the single-heap prologue (include/hs_engine.h)."""
        return int(self._lib.hs_engine_prologue_path(self._h))

    def window_path(self) -> int:
        """Network engines, what the last run_until did: 0 first run since the reset; 1 continued from the state the run before it
        left; 2 nothing moved; 3 repeated from the start (include/hs_engine.h hs_engine_window_path)."""
        return int(self._lib.hs_engine_window_path(self._h))

    def synchronize(self):
        self._check(self._lib.hs_engine_synchronize(self._h))

    def bench_runs(self, end_ns: int, repeats: int):
        """`repeats` x (reset + run) back to back on the engine stream.
        Returns (per-run kernel ms [repeats], total device ms)."""
        k = np.zeros(repeats, np.float32)
        tot = C.c_float(0)
        self._check(self._lib.hs_engine_bench_runs(self._h, int(end_ns), repeats, k.ctypes.data, C.byref(tot)))
        return k, float(tot.value)

    def set_debug_flags(self, flags: int):
        self._lib.hs_debug_set_flags(self._h, flags)

    def set_profile_budget(self, intervals_per_lane: int):
        """Evaluation budget of the tick-table kernel (csrc/hs_tables.hpp): adaptive-Simpson intervals one lane may visit for one
        arrival of a time-varying Source; an arrival beyond it is refused by name.  A run-time setting (default 2**24, or
        2**HS_PROF_BUDGET_LOG2 from the environment); takes effect at the next reset / run."""
        self._check(self._lib.hs_engine_set_profile_budget(self._h, int(intervals_per_lane)))

    # -- results -------------------------------------------------------------------------------
    def summary(self) -> EngineSummary:
        s = N.Summary()
        self._check(self._lib.hs_engine_get_summary(self._h, C.byref(s)))
        return EngineSummary(s)

    def lp_stats(self) -> dict:
        n = self.n
        out = {
            "generated": np.zeros(n, np.int64), "accepted": np.zeros(n, np.int64), "dropped": np.zeros(n, np.int64),
            "completed": np.zeros(n, np.int64), "rejected": np.zeros(n, np.int64),
            "total_service_s": np.zeros(n, np.float64), "sink_received": np.zeros(n, np.int64),
            "queue_depth": np.zeros(n, np.int64), "active": np.zeros(n, np.int32), "events": np.zeros(n, np.int64),
            "final_time_ns": np.zeros(n, np.int64),
        }
        st = N.LpStats(**{k: v.ctypes.data for k, v in out.items()})
        self._check(self._lib.hs_engine_get_lp_stats(self._h, C.byref(st)))
        return out

    def read_sink(self, lp: int, cap: int | None = None):
        if cap is None:
            cap = 1 << 22
        t = np.zeros(cap, np.int64)
        cr = np.zeros(cap, np.int64)
        got = self._check(self._lib.hs_engine_read_sink(self._h, lp, t.ctypes.data, cr.ctypes.data, cap))
        return t[:got], cr[:got]

    def source_generated(self, slot: int) -> np.ndarray:
        """Source.generated_count of the Sources in slot 1..3 of every LP (slot 0: lp_stats()["generated"])."""
        out = np.zeros(self.n, np.int64)
        self._check(self._lib.hs_engine_read_source_generated(self._h, int(slot), out.ctypes.data))
        return out

    def read_probe(self, lp: int, slot: int = 0, cap: int = 1 << 22):
        """(sample ns, value) of the Probe in `slot` of the LP, in sampling order."""
        t = np.zeros(cap, np.int64)
        v = np.zeros(cap, np.int64)
        got = self._check(self._lib.hs_engine_read_probe_slot(self._h, lp, slot, t.ctypes.data, v.ctypes.data, cap))
        return t[:got].copy(), v[:got].copy()

    def read_sinks(self):
        """All sink records: (counts[n], t_ns[total], created_ns[total]) concatenated in LP order."""
        total = self.summary().sink_records
        if self._tandem:            # a Server that forwards to a Server logs its forwards where a Sink's records would go
            total = int(self.lp_stats()["sink_received"].sum())
        counts = np.zeros(self.n, np.int64)
        t = np.zeros(max(total, 1), np.int64)
        cr = np.zeros(max(total, 1), np.int64)
        got = self._check(self._lib.hs_engine_read_sinks(self._h, counts.ctypes.data, t.ctypes.data, cr.ctypes.data,
                                                        max(total, 1)))
        return counts, t[:got], cr[:got]

    def close(self):
        if self._h:
            self._lib.hs_engine_destroy(self._h)
            self._h = C.c_void_p()
        _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def debug_draws(seed: int, sid: int, k0: int, n: int, rate: float, device: int = 0):
    """Device-side (uniform, exp1, ns) for draws k0..k0+n-1 of one stream -- test hook."""
    L = N.lib()
    u = np.zeros(n, np.float64)
    e = np.zeros(n, np.float64)
    ns = np.zeros(n, np.int64)
    rc = L.hs_debug_draws(device, seed, sid, k0, n, rate, u.ctypes.data, e.ctypes.data, ns.ctypes.data)
    if rc == N.HS_E_NO_DEVICE:
        raise N.EngineUnavailable(L.hs_last_global_error().decode())
    if rc < 0:
        raise N.EngineError(rc, L.hs_last_global_error().decode())
    return u, e, ns


def debug_const_div(a: np.ndarray, b: float, device: int = 0):
    """Device-side a / b by the constant-divisor sequence, by IEEE division, and seconds_from_ns(int(a)) -- test hook."""
    L = N.lib()
    a = np.ascontiguousarray(a, np.float64)
    qf, qi, qn = (np.zeros(len(a), np.float64) for _ in range(3))
    rc = L.hs_debug_const_div(device, float(b), len(a), a.ctypes.data, qf.ctypes.data, qi.ctypes.data, qn.ctypes.data)
    if rc == N.HS_E_NO_DEVICE:
        raise N.EngineUnavailable(L.hs_last_global_error().decode())
    if rc < 0:
        raise N.EngineError(rc, L.hs_last_global_error().decode())
    return qf, qi, qn
