"""ctypes binding of libhs_hip.so (the C ABI in include/hs_engine.h) and its build recipe.

The product path has NO CPU fallback: if the shared library is missing or no
MI355X is visible, the engine raises `EngineUnavailable` -- it never routes to
the oracle or to a Python loop.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhs_hip.so")
INCLUDE = os.path.join(_ROOT, "include")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",          # the time algebra / hs_log are fixed IEEE op sequences: never contract to FMA
    "-fno-fast-math", "-fPIC", "-shared", "-Wno-unused-value",
]

HS_OK = 0
HS_E_INVALID, HS_E_NO_DEVICE, HS_E_HIP, HS_E_UNSUPPORTED, HS_E_OVERFLOW, HS_E_STATE = -1, -2, -3, -4, -5, -6
MODE_SINGLE, MODE_REPLICAS = 0, 1
SRC_NONE, SRC_POISSON, SRC_CONSTANT = 0, 1, 2
PROF_CONSTANT, PROF_LINEAR_RAMP, PROF_SPIKE = 0, 1, 2
LAT_EXPONENTIAL, LAT_CONSTANT, LAT_NO_SERVER = 0, 1, 2
EGRESS_NONE, EGRESS_SINK, EGRESS_LINK, EGRESS_ROUTER, EGRESS_SERVER = 0, 1, 2, 3, 4
LB_CONSISTENT_HASH, LB_ROUND_ROBIN, LB_RANDOM = 0, 1, 2
NODE_SOURCE, NODE_SERVER, NODE_SINK, NODE_LINK, NODE_ROUTER, NODE_PROBE, NODE_LB = 0, 1, 2, 3, 4, 5, 6
EV_KINDS = 15
EV_NAMES = ("source", "enqueue", "notify", "poll", "deliver", "work", "continuation", "sink", "link", "link_cont",
            "route", "lb", "lb_resp", "probe_tick", "probe")
PROBE_METRICS = {"depth": 0, "active_requests": 1, "stats_accepted": 2, "stats_dropped": 3, "requests_completed": 4,
                 "_requests_completed": 4, "events_received": 5, "generated_count": 6}
PROBE_NONE = 255
ABI_VERSION = 16
IPC_HANDLE_BYTES = 64


class EngineUnavailable(RuntimeError):
    """The HIP engine cannot run here (library not built, or no GPU)."""


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"hs_engine error {code}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("n_lp", C.c_int32), ("mode", C.c_int32),
        ("start_ns", C.c_int64), ("horizon_ns", C.c_int64), ("seed", C.c_uint64), ("lp_base", C.c_uint64),
        ("log_capacity", C.c_int64),
    ]


class Stations(C.Structure):
    _fields_ = [
        ("src_kind", C.c_void_p), ("src_rate", C.c_void_p), ("src_stop_after_ns", C.c_void_p),
        ("concurrency", C.c_void_p), ("svc_kind", C.c_void_p), ("svc_mean_s", C.c_void_p),
        ("queue_cap", C.c_void_p), ("egress", C.c_void_p), ("seed", C.c_void_p), ("stream_base", C.c_void_p),
        ("src_profile_kind", C.c_void_p), ("src_profile_params", C.c_void_p),
        ("probe_metric", C.c_void_p), ("probe_interval_s", C.c_void_p),
        ("sched_off", C.c_void_p), ("sched_time_ns", C.c_void_p),
        ("source_order", C.c_void_p), ("probe_order", C.c_void_p),
        ("probe_metric_more", C.c_void_p), ("probe_interval_more", C.c_void_p), ("probe_slot_order", C.c_void_p),
        ("sched_rank", C.c_void_p),
        ("src_more_kind", C.c_void_p), ("src_more_rate", C.c_void_p), ("src_more_stop_after_ns", C.c_void_p),
        ("source_slot_order", C.c_void_p),
        ("downstream_lp", C.c_void_p),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("events_processed", C.c_int64), ("events_by_kind", C.c_int64 * EV_KINDS), ("events_cancelled", C.c_int64),
        ("final_time_ns", C.c_int64), ("requests_completed", C.c_int64), ("sink_records", C.c_int64),
        ("last_run_ms", C.c_double), ("kernel_ms", C.c_double), ("launches", C.c_int64), ("window_ns", C.c_int64),
        ("overflow", C.c_int32), ("reserved", C.c_int32),
    ]


class Network(C.Structure):
    _fields_ = [
        ("egress_kind", C.c_void_p), ("router_target0", C.c_void_p), ("router_target1", C.c_void_p),
        ("link_of", C.c_void_p), ("router_stream_base", C.c_void_p), ("n_links", C.c_int32),
        ("link_dst", C.c_void_p), ("link_lat_min_s", C.c_void_p), ("link_jitter_kind", C.c_void_p),
        ("link_jitter_mean_s", C.c_void_p), ("link_stream_base", C.c_void_p), ("link_src", C.c_void_p),
        ("bag_capacity", C.c_int32), ("n_global_lp", C.c_int32), ("link_gid", C.c_void_p),
        ("n_global_links", C.c_int64), ("link_loss_rate", C.c_void_p),
        ("router_n_targets", C.c_void_p), ("router_target2", C.c_void_p), ("router_target3", C.c_void_p),
        ("link_drop_capacity", C.c_void_p),
    ]


class Shard(C.Structure):
    _fields_ = [
        ("rank", C.c_int32), ("world", C.c_int32), ("shard_lo", C.c_void_p), ("outbox_dev", C.c_void_p),
        ("inbox_dev", C.c_void_p), ("msg_capacity", C.c_int32), ("reserved", C.c_int32), ("window_ns", C.c_int64),
        ("gvt_dev", C.c_void_p), ("cand_dev", C.c_void_p),
    ]


class NetStats(C.Structure):
    _fields_ = [("routed", C.c_void_p), ("link_entered", C.c_void_p), ("link_packets_sent", C.c_void_p),
                ("link_packets_dropped", C.c_void_p)]


class LpStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "generated", "accepted", "dropped", "completed", "rejected", "total_service_s", "sink_received",
        "queue_depth", "active", "events", "final_time_ns")]


class GraphConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("start_ns", C.c_int64), ("seed", C.c_uint64),
        ("heap_capacity", C.c_int64), ("request_capacity", C.c_int64), ("record_capacity", C.c_int64), ("max_events", C.c_int64),
        ("profile_budget", C.c_int64),
    ]


class GraphNodes(C.Structure):
    _fields_ = [("n_nodes", C.c_int32)] + [(n, C.c_void_p) for n in (
        "kind", "target", "stream_base", "src_kind", "src_rate", "src_stop_after_ns", "concurrency", "lat_kind", "lat_mean_s",
        "link_lat_min_s", "link_loss_rate", "queue_cap", "rt_off", "rt_cnt", "rt_targets")] + [("n_rt", C.c_int32)] + [
        (n, C.c_void_p) for n in ("src_profile_kind", "src_profile_params", "probe_metric", "probe_interval_s", "lb_strategy", "lb_vnodes",
                                  "names", "name_off", "src_n_clients")]


GRAPH_STATS = ("generated", "payloads", "accepted", "dropped", "completed", "rejected", "total_service_s", "queue_depth", "active",
               "received", "entered", "packets_sent", "packets_dropped", "routed", "rt_taken", "lb")


class GraphStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GRAPH_STATS]


class LbConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("n_sources", C.c_int32), ("n_backends", C.c_int32),
        ("start_ns", C.c_int64), ("horizon_ns", C.c_int64), ("seed", C.c_uint64), ("virtual_nodes", C.c_int32),
        ("shared_sink", C.c_int32), ("tick_capacity", C.c_int64), ("strategy", C.c_int32), ("reserved", C.c_int32),
    ]


class LbSources(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("src_kind", "src_rate", "src_stop_after_ns", "n_clients", "stream_base",
                                          "src_profile_kind", "src_profile_params")]


class LbBackends(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("concurrency", "svc_kind", "svc_mean_s", "queue_cap", "egress", "stream_base",
                                          "names", "name_off")]


class LbStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("generated", "lb", "total_requests", "accepted", "dropped", "completed",
                                          "rejected", "total_service_s", "queue_depth", "active", "sink_received")]


_TUS = ("hs_engine.hip", "hs_lb.hip", "hs_tables.hip", "hs_graph.hip")          # one object each ...
_INST_TU, _INST_GROUPS = "hs_inst.hip", 17                      # ... plus hs_inst.hip once per instantiation group (csrc/hs_kernels.hpp)
_STAMP_TU = "hs_stamp.hip"                                      # ... plus the build identity (the sources' hash, inside the .so)
_MARK, _MARK_END = b"HS_SRC_HASH=", b"=HS_SRC_HASH_END"


def sources() -> list[str]:
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [
        os.path.join(INCLUDE, "hs_engine.h")]


def _sources_hash() -> str:
    """Content hash of everything the library is built from (mtimes do not survive the copy to a GPU box)."""
    import hashlib
    hh = hashlib.sha256()
    for src in sources():
        hh.update(os.path.basename(src).encode())
        with open(src, "rb") as f:
            hh.update(f.read())
    hh.update(" ".join(HIPCC_FLAGS).encode())
    return hh.hexdigest()


def built_from(path: str) -> str | None:
    """The sources' hash (+ "|" + extra defines) the library at `path` was built from: it is compiled into the library
    (csrc/hs_stamp.hip), so a checkout that changes csrc/ cannot leave a stale binary looking current (no side file to trust)."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    while i >= 0:                                   # (the marker's own pieces also occur as separate literals: take the full form)
        j = blob.find(_MARK_END, i)
        if 0 <= j - i - len(_MARK) <= 4096 and b"\0" not in blob[i:j]:
            return blob[i + len(_MARK):j].decode(errors="replace")
        i = blob.find(_MARK, i + 1)
    return None


def is_stale() -> bool:
    return built_from(LIB_PATH) != _sources_hash() + "|"


def build(force: bool = False, verbose: bool = False, defines: tuple = (), lib_path: str | None = None) -> str:
    """Cross-compile libhs_hip.so for gfx950 with hipcc (works without a GPU): every translation unit -- and every group of
    kernel instantiations of hs_inst.hip -- is its own object, compiled in parallel (HS_BUILD_JOBS, default = CPU count),
    then linked.  Objects whose inputs did not change are kept.  `defines` / `lib_path`: an instrumented copy of the library
    (tools/cycles.py: -DHS_CYCLES ...) with its own object directory; load it with HS_HIP_LIB=<lib_path>."""
    import concurrent.futures

    os.makedirs(LIB_DIR, exist_ok=True)
    out = lib_path or LIB_PATH
    extra = [f"-D{d}" for d in defines]
    want = _sources_hash() + "|" + " ".join(extra)
    if not force and built_from(out) == want:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = os.path.join(LIB_DIR, "obj") if not lib_path else out + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    jobs = [(tu, [], os.path.join(obj_dir, tu.replace(".hip", ".o"))) for tu in _TUS]
    jobs += [(_INST_TU, [f"-DHS_INST={k}"], os.path.join(obj_dir, f"hs_inst_{k}.o")) for k in range(_INST_GROUPS)]
    jobs += [(_STAMP_TU, [f'-DHS_SOURCES_HASH="{want}"'], os.path.join(obj_dir, "hs_stamp.o"))]

    def inputs_hash(tu, defs):
        """What one object is built from: its translation unit, every header, the flags and its own defines (another .hip file
        changing does not touch it)."""
        import hashlib
        hh = hashlib.sha256()
        for src in sources():
            if src.endswith(".hip") and os.path.basename(src) != tu:
                continue
            hh.update(os.path.basename(src).encode())
            with open(src, "rb") as f:
                hh.update(f.read())
        hh.update(" ".join(flags + list(defs)).encode())
        return hh.hexdigest()

    def compile_one(job):
        tu, defs, obj = job
        tag = obj + ".stamp"
        mine = inputs_hash(tu, defs)
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == mine:
            return obj
        cmd = [hipcc, *flags, *defs, "-c", os.path.join(CSRC, tu), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(tag, "w") as f:
            f.write(mine)
        return obj

    n_jobs = int(os.environ.get("HS_BUILD_JOBS", "0")) or (os.cpu_count() or 4)
    # the longest compilations first (the asynchronous network kernels)
    order = sorted(jobs, key=lambda j: 0 if "hs_inst" in j[2] and any(f"_{k}.o" in j[2] for k in (8, 9, 10, 11, 12, 5, 6, 7)) else 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=n_jobs) as ex:
        objs = list(ex.map(compile_one, order))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *sorted(objs), "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    assert built_from(out) == want, "the build identity did not make it into the library"
    return out


_lib = None


def lib():
    """Load libhs_hip.so; raises EngineUnavailable (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stale = is_stale()
    if stale and os.path.exists(hipcc) and not os.environ.get("HS_HIP_LIB"):
        build()     # never run a library older than its sources
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the engine has no CPU fallback)")
    # One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7 + HSA runtime; whichever copy is mapped
    # first serves both.  torch first is the order every GPU test and bench runs in; this library first leaves torch (device
    # properties, the shard exchange tensors) with "No HIP GPUs are available" -- seen on MI355X with a script that ran a
    # network through Simulation.run() before anything had imported torch.  So torch, when installed, always goes first.
    if not os.environ.get("HS_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(os.environ.get("HS_HIP_LIB") or LIB_PATH)   # HS_HIP_LIB: an instrumented build (tools/cycles.py)
    P = C.POINTER
    L.hs_abi_version.restype = C.c_int
    L.hs_build_sources_hash.restype = C.c_char_p
    L.hs_device_count.restype = C.c_int
    L.hs_engine_create.restype = C.c_int
    L.hs_engine_create.argtypes = [P(Config), P(C.c_void_p)]
    L.hs_engine_set_stations.restype = C.c_int
    L.hs_engine_set_stations.argtypes = [C.c_void_p, P(Stations)]
    L.hs_engine_set_network.restype = C.c_int
    L.hs_engine_set_network.argtypes = [C.c_void_p, P(Network)]
    L.hs_engine_get_net_stats.restype = C.c_int
    L.hs_engine_get_net_stats.argtypes = [C.c_void_p, P(NetStats)]
    L.hs_engine_set_stream.restype = C.c_int
    L.hs_engine_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.hs_engine_shard_attach.restype = C.c_int
    L.hs_engine_shard_attach.argtypes = [C.c_void_p, P(Shard)]
    L.hs_engine_shard_begin.restype = C.c_int
    L.hs_engine_shard_begin.argtypes = [C.c_void_p, C.c_int64]
    for name in ("hs_engine_shard_window", "hs_engine_shard_inject", "hs_engine_shard_final"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p, C.c_int64]
    L.hs_engine_shard_progress.restype = C.c_int
    L.hs_engine_shard_progress.argtypes = [C.c_void_p, C.c_int64, P(C.c_int64)]
    L.hs_engine_shard_overshoot.restype = C.c_int
    L.hs_engine_shard_overshoot.argtypes = [C.c_void_p, C.c_int32]
    L.hs_engine_shard_async_setup.restype = C.c_int
    L.hs_engine_shard_async_setup.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    for name in ("hs_engine_shard_round", "hs_engine_shard_inject_async"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p]
    L.hs_engine_shard_ipc_export.restype = C.c_int
    L.hs_engine_shard_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_engine_shard_ipc_attach.restype = C.c_int
    L.hs_engine_shard_ipc_attach.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_engine_shard_ipc_buffers.restype = C.c_int
    L.hs_engine_shard_ipc_buffers.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_void_p)]
    L.hs_engine_shard_peers_local.restype = C.c_int
    L.hs_engine_shard_peers_local.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("hs_engine_shard_push", "hs_engine_shard_inject_ipc"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p]
    L.hs_engine_shard_live_export.restype = C.c_int
    L.hs_engine_shard_live_export.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_engine_shard_live_attach.restype = C.c_int
    L.hs_engine_shard_live_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("hs_engine_shard_live_run", "hs_engine_shard_live_wait"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p]
    L.hs_engine_shard_async_done.restype = C.c_int
    L.hs_engine_shard_async_done.argtypes = [C.c_void_p, P(C.c_int32)]
    L.hs_engine_set_link_drops.restype = C.c_int
    L.hs_engine_set_link_drops.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.hs_engine_read_send_log.restype = C.c_int64
    L.hs_engine_read_send_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_engine_reset.restype = C.c_int
    L.hs_engine_reset.argtypes = [C.c_void_p]
    for name in ("hs_engine_run_until", "hs_engine_run_until_async"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p, C.c_int64]
    L.hs_engine_tandem_path.restype = C.c_int
    L.hs_engine_tandem_path.argtypes = [C.c_void_p]
    L.hs_engine_prologue_path.restype = C.c_int
    L.hs_engine_prologue_path.argtypes = [C.c_void_p]
    L.hs_engine_window_path.restype = C.c_int
    L.hs_engine_window_path.argtypes = [C.c_void_p]
    L.hs_engine_synchronize.restype = C.c_int
    L.hs_engine_synchronize.argtypes = [C.c_void_p]
    L.hs_engine_bench_runs.restype = C.c_int
    L.hs_engine_bench_runs.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.hs_engine_get_summary.restype = C.c_int
    L.hs_engine_get_summary.argtypes = [C.c_void_p, P(Summary)]
    L.hs_engine_get_lp_stats.restype = C.c_int
    L.hs_engine_get_lp_stats.argtypes = [C.c_void_p, P(LpStats)]
    L.hs_engine_read_sink.restype = C.c_int64
    L.hs_engine_read_sink.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_engine_read_probe.restype = C.c_int64
    L.hs_engine_read_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_engine_read_probe_slot.restype = C.c_int64
    L.hs_engine_read_probe_slot.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_engine_read_source_generated.restype = C.c_int
    L.hs_engine_read_source_generated.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.hs_engine_read_sinks.restype = C.c_int64
    L.hs_engine_read_sinks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_last_error.restype = C.c_char_p
    L.hs_last_error.argtypes = [C.c_void_p]
    L.hs_last_global_error.restype = C.c_char_p
    L.hs_engine_destroy.restype = None
    L.hs_engine_destroy.argtypes = [C.c_void_p]
    L.hs_debug_set_flags.restype = C.c_int
    L.hs_debug_set_flags.argtypes = [C.c_void_p, C.c_int]
    L.hs_debug_async_counters.restype = C.c_int
    L.hs_debug_async_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_debug_draws.restype = C.c_int
    L.hs_debug_draws.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_debug_const_div.restype = C.c_int
    L.hs_debug_const_div.argtypes = [C.c_int32, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_engine_set_profile_budget.restype = C.c_int
    L.hs_engine_set_profile_budget.argtypes = [C.c_void_p, C.c_int64]
    L.hs_lb_set_profile_budget.restype = C.c_int
    L.hs_lb_set_profile_budget.argtypes = [C.c_void_p, C.c_int64]
    L.hs_debug_tick_table.restype = C.c_int64
    L.hs_debug_tick_table.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.hs_lb_create.restype = C.c_int
    L.hs_lb_create.argtypes = [P(LbConfig), P(LbSources), P(LbBackends), P(C.c_void_p)]
    L.hs_lb_run.restype = C.c_int
    L.hs_lb_run.argtypes = [C.c_void_p, C.c_int64]
    L.hs_lb_bench_runs.restype = C.c_int
    L.hs_lb_bench_runs.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.hs_lb_get_summary.restype = C.c_int
    L.hs_lb_get_summary.argtypes = [C.c_void_p, P(Summary)]
    L.hs_lb_get_stats.restype = C.c_int
    L.hs_lb_get_stats.argtypes = [C.c_void_p, P(LbStats)]
    L.hs_lb_set_probes.restype = C.c_int
    L.hs_lb_set_probes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_lb_read_probe.restype = C.c_int64
    L.hs_lb_read_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_lb_read_sink.restype = C.c_int64
    L.hs_lb_read_sink.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_lb_latency_stats.restype = C.c_int
    L.hs_lb_latency_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_lb_ring.restype = C.c_int
    L.hs_lb_ring.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_lb_select.restype = C.c_int32
    L.hs_lb_select.argtypes = [C.c_void_p, C.c_char_p]
    L.hs_lb_last_error.restype = C.c_char_p
    L.hs_lb_last_error.argtypes = [C.c_void_p]
    L.hs_lb_destroy.restype = None
    L.hs_lb_destroy.argtypes = [C.c_void_p]
    L.hs_md5.restype = None
    L.hs_md5.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
    L.hs_debug_lb_flags.restype = C.c_int
    L.hs_debug_lb_flags.argtypes = [C.c_void_p, C.c_int]
    L.hs_merge_sink_records.restype = C.c_int
    L.hs_merge_sink_records.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.hs_sink_latency_stats.restype = C.c_int
    L.hs_sink_latency_stats.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_set_float_sum_mode.restype = C.c_int
    L.hs_set_float_sum_mode.argtypes = [C.c_int]
    L.hs_debug_radix_sort.restype = C.c_int
    L.hs_debug_radix_sort.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.hs_graph_create.restype = C.c_int
    L.hs_graph_create.argtypes = [P(GraphConfig), P(GraphNodes), P(C.c_void_p)]
    L.hs_graph_schedule.restype = C.c_int
    L.hs_graph_schedule.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    L.hs_graph_run_until.restype = C.c_int
    L.hs_graph_run_until.argtypes = [C.c_void_p, C.c_int64]
    L.hs_graph_run_many.restype = C.c_int
    L.hs_graph_run_many.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int64]
    L.hs_graph_run_parts.restype = C.c_int
    L.hs_graph_run_parts.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int64]
    L.hs_graph_get_summary.restype = C.c_int
    L.hs_graph_get_summary.argtypes = [C.c_void_p, P(Summary)]
    L.hs_graph_get_stats.restype = C.c_int
    L.hs_graph_get_stats.argtypes = [C.c_void_p, P(GraphStats)]
    L.hs_graph_read_records.restype = C.c_int64
    L.hs_graph_read_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.hs_graph_last_error.restype = C.c_char_p
    L.hs_graph_last_error.argtypes = [C.c_void_p]
    L.hs_graph_destroy.restype = None
    L.hs_graph_destroy.argtypes = [C.c_void_p]
    if L.hs_abi_version() != ABI_VERSION:
        raise EngineUnavailable("libhs_hip.so ABI version mismatch; rebuild")
    import sys
    L.hs_set_float_sum_mode(1 if sys.version_info >= (3, 12) else 0)   # `sum(floats)`: Neumaier-compensated from CPython 3.12 on
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "hs_abi_version", "hs_build_sources_hash", "hs_device_count", "hs_engine_create", "hs_engine_set_stations", "hs_engine_set_network",
    "hs_engine_get_net_stats", "hs_engine_set_link_drops", "hs_engine_read_send_log", "hs_engine_set_stream", "hs_engine_shard_attach", "hs_engine_shard_begin",
    "hs_engine_shard_window", "hs_engine_shard_inject", "hs_engine_shard_progress", "hs_engine_shard_final",
    "hs_engine_shard_overshoot", "hs_engine_shard_async_setup", "hs_engine_shard_round",
    "hs_engine_shard_inject_async", "hs_engine_shard_async_done", "hs_engine_shard_ipc_export", "hs_engine_shard_ipc_attach",
    "hs_engine_shard_ipc_buffers", "hs_engine_shard_peers_local", "hs_engine_shard_push", "hs_engine_shard_inject_ipc",
    "hs_engine_shard_live_export", "hs_engine_shard_live_attach",
    "hs_engine_shard_live_run", "hs_engine_shard_live_wait", "hs_engine_reset",
    "hs_engine_run_until", "hs_engine_run_until_async", "hs_engine_synchronize", "hs_engine_tandem_path", "hs_engine_prologue_path", "hs_engine_window_path", "hs_engine_bench_runs",
    "hs_engine_get_summary", "hs_engine_get_lp_stats", "hs_engine_read_sink", "hs_engine_read_sinks", "hs_engine_read_probe",
    "hs_engine_read_probe_slot", "hs_engine_read_source_generated",
    "hs_last_error", "hs_last_global_error", "hs_engine_destroy", "hs_debug_draws", "hs_debug_set_flags",
    "hs_debug_const_div", "hs_debug_async_counters",
    "hs_lb_create", "hs_lb_run", "hs_lb_bench_runs", "hs_lb_get_summary", "hs_lb_get_stats", "hs_lb_read_sink",
    "hs_lb_set_probes", "hs_lb_read_probe",
    "hs_lb_latency_stats",
    "hs_lb_ring", "hs_lb_select", "hs_lb_last_error", "hs_lb_destroy", "hs_md5", "hs_debug_radix_sort", "hs_merge_sink_records",
    "hs_sink_latency_stats", "hs_set_float_sum_mode",
    "hs_debug_lb_flags", "hs_engine_set_profile_budget", "hs_lb_set_profile_budget", "hs_debug_tick_table",
    "hs_graph_create", "hs_graph_schedule", "hs_graph_run_until", "hs_graph_run_many", "hs_graph_run_parts", "hs_graph_get_summary", "hs_graph_get_stats", "hs_graph_read_records",
    "hs_graph_last_error", "hs_graph_destroy",
)
