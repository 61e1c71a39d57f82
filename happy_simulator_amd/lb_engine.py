"""Object wrapper over the load-balancer part of the C ABI (include/hs_engine.h, `hs_lb_*`): one
`LoadBalancerEngine` = one `hs_lb` handle on one GPU.

Host-side plumbing only (numpy buffers in, numpy buffers out); the md5 ring, the client -> backend table, the
sources, the sorts and the backends all live in libhs_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .engine import EngineSummary


@dataclass
class LbSourceArrays:
    """`hs_lb_sources`: the Sources in front of the LoadBalancer."""

    n: int
    src_rate: np.ndarray
    n_clients: np.ndarray
    src_kind: np.ndarray | None = None
    src_stop_after_ns: np.ndarray | None = None
    stream_base: np.ndarray | None = None
    src_profile_kind: np.ndarray | None = None      # N.PROF_*; None = constant rates (src_rate = the peak rate of a profile)
    src_profile_params: np.ndarray | None = None    # [n, 4]


@dataclass
class LbBackendArrays:
    """`hs_lb_backends`: the Servers behind the LoadBalancer, in add_backend order."""

    n: int
    names: list
    concurrency: np.ndarray | None = None
    svc_kind: np.ndarray | None = None
    svc_mean_s: np.ndarray | None = None
    queue_cap: np.ndarray | None = None
    egress: np.ndarray | None = None
    stream_base: np.ndarray | None = None


class LoadBalancerEngine:
    """S Sources -> LoadBalancer(ConsistentHash(virtual_nodes)) -> B Servers -> Sink(s) on one device."""

    def __init__(self, sources: LbSourceArrays, backends: LbBackendArrays, *, virtual_nodes: int, horizon_ns: int,
                 shared_sink: bool = True, start_ns: int = 0, seed: int = 42, device: int = 0, tick_capacity: int = 0,
                 strategy: int = N.LB_CONSISTENT_HASH):
        self._lib = N.lib()
        if self._lib.hs_device_count() <= 0:
            raise N.EngineUnavailable("no HIP device visible: the engine has no CPU fallback")
        self.S, self.B = int(sources.n), int(backends.n)
        self.virtual_nodes = int(virtual_nodes)
        self.shared_sink = bool(shared_sink)
        self._h = C.c_void_p()
        cfg = N.LbConfig(C.sizeof(N.LbConfig), device, self.S, self.B, start_ns, horizon_ns, seed, self.virtual_nodes,
                         1 if shared_sink else 0, tick_capacity, int(strategy), 0)
        self.strategy = int(strategy)
        keep = []

        def fill(struct, obj, n, fields):
            for name, dtype in fields:
                a = getattr(obj, name)
                if a is None:
                    setattr(struct, name, None)
                    continue
                a = np.ascontiguousarray(a, dtype)
                if a.shape != (n,):
                    raise ValueError(f"{name} must have shape ({n},)")
                keep.append(a)
                setattr(struct, name, a.ctypes.data)

        src = N.LbSources()
        fill(src, sources, self.S, (("src_kind", np.uint8), ("src_rate", np.float64), ("src_stop_after_ns", np.int64),
                                   ("n_clients", np.int64), ("stream_base", np.uint64), ("src_profile_kind", np.uint8)))
        if sources.src_profile_kind is not None:
            pp = np.ascontiguousarray(sources.src_profile_params, np.float64)
            if pp.shape != (self.S, 4):
                raise ValueError(f"src_profile_params must have shape ({self.S}, 4)")
            keep.append(pp)
            src.src_profile_params = pp.ctypes.data
        be = N.LbBackends()
        fill(be, backends, self.B, (("concurrency", np.int32), ("svc_kind", np.uint8), ("svc_mean_s", np.float64),
                                    ("queue_cap", np.int64), ("egress", np.uint8), ("stream_base", np.uint64)))
        if len(backends.names) != self.B:
            raise ValueError("one name per backend is required")
        enc = [str(nm).encode() for nm in backends.names]
        blob = C.create_string_buffer(b"".join(enc) + b"\0")
        name_off = np.zeros(self.B + 1, np.int32)
        name_off[1:] = np.cumsum([len(b) for b in enc])
        keep += [blob, name_off]
        be.names = C.cast(blob, C.c_void_p).value
        be.name_off = name_off.ctypes.data
        rc = self._lib.hs_lb_create(C.byref(cfg), C.byref(src), C.byref(be), C.byref(self._h))
        if rc != N.HS_OK:
            msg = (self._lib.hs_lb_last_error(None) or b"").decode()
            self._h = C.c_void_p()
            if rc == N.HS_E_NO_DEVICE:
                raise N.EngineUnavailable(msg)
            raise N.EngineError(rc, msg)
        if os.environ.get("HS_PROF_BUDGET_LOG2"):
            self.set_profile_budget(1 << int(os.environ["HS_PROF_BUDGET_LOG2"]))

    # ------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise N.EngineError(rc, (self._lib.hs_lb_last_error(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hs_lb_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def run(self, end_ns: int) -> None:
        """`Simulation.__init__` + `run()` to end_ns (one-event overshoot included)."""
        self._check(self._lib.hs_lb_run(self._h, int(end_ns)))

    def set_probes(self, target_kind, target_index, metric, interval_s) -> None:
        """Probe.on(<backend Server> | <Sink>, metric, interval) (include/hs_engine.h `hs_lb_set_probes`): target_kind 0 =
        backend Server, 1 = Sink; metric = N.PROBE_METRICS id; once, before the first run."""
        k = np.ascontiguousarray(target_kind, np.int32)
        i = np.ascontiguousarray(target_index, np.int32)
        m = np.ascontiguousarray(metric, np.uint8)
        iv = np.ascontiguousarray(interval_s, np.float64)
        if not (k.shape == i.shape == m.shape == iv.shape and k.ndim == 1):
            raise ValueError("one target kind, index, metric and interval per probe")
        self._check(self._lib.hs_lb_set_probes(self._h, len(k), k.ctypes.data, i.ctypes.data, m.ctypes.data, iv.ctypes.data))
        self.n_probes = len(k)

    def read_probe(self, probe: int, cap: int = 1 << 22):
        """(sample times ns, sampled values) of probe `probe` in the last run."""
        t = np.zeros(cap, np.int64)
        v = np.zeros(cap, np.int64)
        n = self._check(self._lib.hs_lb_read_probe(self._h, int(probe), t.ctypes.data, v.ctypes.data, cap))
        return t[:n].copy(), v[:n].copy()

    def bench_runs(self, end_ns: int, repeats: int):
        run_ms = np.zeros(repeats, np.float32)
        sort_ms = np.zeros(repeats, np.float32)
        self._check(self._lib.hs_lb_bench_runs(self._h, int(end_ns), repeats, run_ms.ctypes.data, sort_ms.ctypes.data))
        return run_ms, sort_ms

    def set_debug_flags(self, flags: int) -> None:
        self._check(self._lib.hs_debug_lb_flags(self._h, flags))

    def set_profile_budget(self, intervals_per_lane: int) -> None:
        """Evaluation budget of the tick-table kernel (StationEngine.set_profile_budget); before the first run."""
        self._check(self._lib.hs_lb_set_profile_budget(self._h, int(intervals_per_lane)))

    def summary(self) -> EngineSummary:
        s = N.Summary()
        self._check(self._lib.hs_lb_get_summary(self._h, C.byref(s)))
        return EngineSummary(s)

    def stats(self) -> dict:
        out = {
            "generated": np.zeros(self.S, np.int64), "lb": np.zeros(5, np.int64),
            "total_requests": np.zeros(self.B, np.int64), "accepted": np.zeros(self.B, np.int64),
            "dropped": np.zeros(self.B, np.int64), "completed": np.zeros(self.B, np.int64),
            "rejected": np.zeros(self.B, np.int64), "total_service_s": np.zeros(self.B, np.float64),
            "queue_depth": np.zeros(self.B, np.int64), "active": np.zeros(self.B, np.int32),
            "sink_received": np.zeros(self.B, np.int64),
        }
        st = N.LbStats()
        for k, a in out.items():
            setattr(st, k, a.ctypes.data)
        self._check(self._lib.hs_lb_get_stats(self._h, C.byref(st)))
        return out

    def read_sink(self, sink: int = 0, cap: int | None = None):
        """(completion ns, created_at ns) of one Sink in its processing order."""
        if cap is None:
            cap = int(self.summary().sink_records) if self.shared_sink else int(self.stats()["sink_received"][sink])
        t = np.zeros(max(cap, 1), np.int64)
        cr = np.zeros(max(cap, 1), np.int64)
        n = self._check(self._lib.hs_lb_read_sink(self._h, sink, t.ctypes.data, cr.ctypes.data, cap))
        return t[:n], cr[:n]

    def latency_stats(self) -> dict:
        """`Sink.latency_stats()` of the shared Sink, computed on the device."""
        out = (C.c_double * 6)()
        self._check(self._lib.hs_lb_latency_stats(self._h, out))
        return {"count": int(out[0]), "avg": out[1], "min": out[2], "max": out[3], "p50": out[4], "p99": out[5]}

    def ring(self) -> np.ndarray:
        out = np.zeros(self.B * self.virtual_nodes, np.int32)
        self._check(self._lib.hs_lb_ring(self._h, out.ctypes.data))
        return out

    def select(self, key: str) -> int:
        return int(self._check(self._lib.hs_lb_select(self._h, str(key).encode())))


def md5(data: bytes) -> bytes:
    """The ring's hash function as the library computes it (host code; no GPU needed)."""
    out = (C.c_uint8 * 16)()
    N.lib().hs_md5(data, len(data), out)
    return bytes(out)


def radix_sort(keys: np.ndarray, vals: np.ndarray, key_bits: int, device: int = 0):
    """Test hook: the engine's device radix sort on host arrays.  Returns (keys, vals, device ms)."""
    keys = np.ascontiguousarray(keys, np.uint64)
    vals = np.ascontiguousarray(vals, np.uint64)
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    ms = C.c_float(0)
    rc = N.lib().hs_debug_radix_sort(device, len(keys), key_bits, keys.ctypes.data, vals.ctypes.data, ko.ctypes.data,
                                     vo.ctypes.data, C.byref(ms))
    if rc != N.HS_OK:
        raise N.EngineError(rc, (N.lib().hs_lb_last_error(None) or b"").decode())
    return ko, vo, float(ms.value)
