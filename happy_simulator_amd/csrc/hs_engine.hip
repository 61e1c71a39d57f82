// hs_engine.hip -- libhs_hip.so: HIP kernels (gfx950) and the C ABI declared in include/hs_engine.h.
//
// Replaces `Simulation._execute_until` / `_build_summary` (happysimulator/core/simulation.py:449-505,
// :543-591) and the replica fan-out of happysimulator/parallel for station LPs.  Data layout, kernels and
// their rooflines are described in DESIGN.md.  There is no CPU fallback in this file by design.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <utility>

#define HS_KERNELS_MAIN
#include "hs_kernels.hpp"
#include "hs_tables_api.hpp"

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_global_error;

struct hs_engine {
    bool pending_async = false;   // hs_engine_run_until_async enqueued a run whose results hs_engine_synchronize has not finalised yet

    hs_config cfg{};
    int C = 1;                 // departure slots compiled for (>= max concurrency)
    bool have_stations = false;
    bool initialised = false;  // reset done
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::vector<void *> allocs;
    std::vector<std::pair<void *, size_t>> uncached;   // buffers of the process-wide uncached pool this engine holds (uncached_alloc)
    StationParams P{};
    StationState X{};
    RecordLogs L{};
    Totals *tot = nullptr;
    Candidate *cands = nullptr;
    bool is_net = false;
    bool any_xsrc = false;     // some LP has more than one Source (general path + prologue)
    int n_pass = 0;            // tandem queues (Server -> Server): passes of the station kernel, 0 = none
    // ... whose nanosecond ties the passes' lineage key does not decide (Totals::undecided), or which stand next to Probes / scheduled
    // Requests / several Sources per Server, run on the single-heap loop (hs_exact.hpp) from start to end instead
    bool exact_only = false;
    bool net_on_heap = false;     // ... a network that moved there behind an undecided election (tandem_fallback); hs_engine_reset moves it back
    std::vector<int32_t> h_src_lp;     // the Sources in `sources=[...]` order (hs_engine_set_stations), kept for setup_exact_plain
    std::vector<uint8_t> h_src_slot;
    bool tandem_fan_in = false;   // some Server is the downstream of several Servers: no passes, the single heap from the start
    bool exact_prologue = false;   // the prologue takes part in ordinary runs (pre-run events whose indices run-time events can pass)
    // ... but only where a pre-run event shares its nanosecond with another event of its LP, or while the run has created fewer
    // events than there are pre-run events, can the second counter change an order -- and the prologue is one lane for the whole
    // engine (65 536 chains with a Probe each: 2.2 s before a 3 ms run).  A station engine therefore runs WITHOUT it first; the
    // kernels report the coincidences (Totals::undecided bit 2), and only then is the run repeated, window by window, behind the
    // prologue (prologue_fallback).  Debug flag 1 << 16 keeps the prologue in every run.  Round 4: network engines driven with
    // hs_engine_run_until do the same (NetStation::run_group reports the coincidences, messages included) -- a 16 384-station ring with
    // a Probe per station spent 0.5 s on the single lane before a run of milliseconds; shards of a partitioned network keep it.
    bool lazy_prologue = false, lazy_failed = false;
    int64_t n_init = 0;            // pre-run events of the engine
    std::vector<int64_t> window_ends;   // tandem queues: the end times of the run_until calls since the last reset (replayed on the single heap)
    bool uni_stations = false; // every LP: Poisson Source, exponential single-worker Server, unbounded queue, no stop_after
    bool uni_grid = false;     // ... and a Sink behind every Server: hs_station_run<1, false, true, true>
    bool f64_times = false;    // every time of a run is a whole number of ns in [0, 2^52): the UNI kernels' exact binary64 time algebra
    bool net_uni = false;      // ... and every router has exactly one NetworkLink (exponential jitter, no loss): hs_net_async<1, false, true>
    bool any_timevarying = false, any_sched = false;   // (subsets of any_profile: what a network does not lower)
    bool any_profile = false;  // some source has a time-varying rate profile (or a probe: same kernel instantiation)
    bool any_probe = false;
    int n_probe_slots = 1;     // probe slots in use (max probes on one LP): sizes the probe logs
    NetParams NP{};
    NetState NX{};
    ShardCtl SC{};             // wend_slots == nullptr: the engine holds the whole network
    bool net_global = false;   // link endpoints are network-wide station indices (set_network with n_global_lp > 0)
    int32_t n_global_lp = 0;
    int32_t *gid2local = nullptr;
    int64_t n_gid = 0;
    const int64_t *inbox = nullptr;
    int64_t *shard_gvt = nullptr;
    int64_t final_win = 0;
    bool external_stream = false;
    hipStream_t own_stream = nullptr;
    std::vector<int32_t> h_link_dst, h_link_src, h_gid2local;
    int64_t window_ns = 0;
    bool net_ran = false;
    int64_t net_last_end = INT64_MIN;   // end_ns of the run since the last reset (windows: see hs_engine_run_until_async)
    bool net_resume = false;            // ... and this run_until continues from the state that run left (hs_net_resume first)
    int64_t net_resume_from = 0;        // ... whose end_ns this was
    int net_window_path = 0;            // hs_engine_window_path
    int xs_phase_seen = 0;              // ... and the prologue's phase (XState::phase)
    Totals tot_seen{};                  // the totals hs_engine_run_until read behind the last run (valid until the next launch):
    bool tot_seen_valid = false;        //   the next window's decision without another round trip to the device
    std::vector<int64_t> drop_off_host;  // table-decided link losses (hs_network.link_drop_capacity): bit offsets per link
    uint32_t *drop_bits_dev = nullptr;
    LossTables loss_host{};              // (host copy of the device object: the send log's pointers)
    bool async_ok = false;     // the network can run on hs_net_async (whole network on this engine, queues allocated)
    int round_iters = 0;       // > 0: hs_net_async runs one exchange round of a shard (that many iterations), not a whole run
    // asynchronous shard rounds (hs_engine_shard_async_*): the network's cross-shard links
    int n_cross = 0;
    const int32_t *cross_local = nullptr;   // [n_cross] local link index (-1: this shard does not touch the link)
    const uint8_t *cross_role = nullptr;    // [n_cross] bit 0: the source station is here, bit 1: the destination is
    int64_t *cross_bounds = nullptr;        // device int64[n_cross + 1], owned by the caller (all-reduced with MAX)
    bool shard_async = false;
    // device-side exchange (hs_engine_shard_ipc_*): this rank's exchange buffers, written by the peers, and the peers' buffers
    int64_t *ipc_inbox = nullptr, *ipc_bounds = nullptr;      // [2][world][row] / [2][world][n_cross + 1], uncached device memory
    int64_t **peer_inbox_dev = nullptr, **peer_bounds_dev = nullptr;   // device arrays [world] of the ranks' buffers as mapped here
    std::vector<void *> ipc_opened;                           // peer mappings to close
    // LIVE exchange (ShardCtl::live): the ranks' link-queue arrays as mapped here, the links' indices at their destination ranks
    bool live_ready = false;
    int64_t **live_rec_dev = nullptr, **live_ea_dev = nullptr;
    unsigned long long **live_head_dev = nullptr;
    int32_t *live_link_dev = nullptr;
    hipStream_t live_stream = nullptr;   // the live launch's own stream: it waits for the peers' launches, nothing may queue behind it
    hipEvent_t live_ev = nullptr;
    bool ipc_ready = false;
    int ipc_parity = 0;
    bool net_pf = false;       // the network has probes / profiles / scheduled Requests: the PF instantiation of hs_net_async
    int round_iters_cfg = 0;
    int async_fit = -1;        // -1 unknown, 0 the grid is not co-resident (segments take turns: run_net_segments), 1 it is
    long long async_resident_blocks = 0;   // workgroups of hs_net_async one cooperative launch holds
    int async_lanes = 64;      // LPs per wavefront in hs_net_async
    int n_blocks = 0;
    int flags = 0;
    // prologue (hs_exact.hpp): SINGLE mode with probes / scheduled Requests
    bool exact = false;
    XState *xs = nullptr;
    XState xs_host{};          // the device pointers / capacities of *xs (phase etc. are reset from it)
    XInit XI{};
    // K lanes per LP (hs_kernels_wide.hpp): the uniform grid with fewer LPs than the device has lanes
    WideCtl *wide_ctl = nullptr; int32_t *wide_bail = nullptr;
    WavePart *wave_parts = nullptr;   // one wavefront per LP (hs_kernels_wave.hpp): the workgroups' partial totals
    int wide_K = 0;            // 0: one lane per LP (hs_station_run)
    bool fresh = false;        // nothing has run since the last reset (the wide kernel starts from empty queues)
    // hs_engine_reset on an engine whose next run is one wavefront per LP: the bootstrap is DEFERRED into that run's kernel
    // (hs_station_wave<NW, true>); anything else that touches the state first launches the reset kernel after all (ensure_reset)
    bool reset_pending = false;
    mutable long long device_lanes = 0;   // CUs x 4 SIMDs x 64 lanes of the engine's device (wide_lanes)
    // tick tables (hs_tables.hpp): Sources with a time-varying profile and Probes
    TickRow *tab_rows = nullptr; int n_tab_rows = 0; int64_t tab_cap = 0;
    int64_t *tab_times = nullptr, *tab_count = nullptr;
    unsigned long long *tab_status = nullptr;
    bool tables_built = false;
    long long lane_budget = kDefaultLaneBudget;
    double last_run_ms = 0.0, last_kernel_ms = 0.0;
    int64_t launches = 0;
    std::string error;
};

namespace {

int fail(hs_engine *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf;
    g_global_error = buf;
    return code;
}

#define HS_HIP(h, expr)                                                                                \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(h, HS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

template <typename T>
int dev_alloc(hs_engine *h, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) ? count * sizeof(T) : sizeof(T));
    if (e != hipSuccess) return fail(h, HS_E_HIP, "hipMalloc(%zu B): %s", count * sizeof(T), hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    // debug: HS_POISON_ALLOC=<byte> fills every device allocation with that byte -- nothing may depend on what hipMalloc hands out
    // (memory of an engine destroyed earlier in the process): tests/test_gpu_sharded.py::test_nothing_depends_on_what_the_allocator_hands_out
    static const char *poison = getenv("HS_POISON_ALLOC");
    if (poison && *poison) (void)hipMemset(q, (int)strtol(poison, nullptr, 0) & 0xff, count * sizeof(T) ? count * sizeof(T) : sizeof(T));
    return HS_OK;
}

// Uncached device memory (hipExtMallocWithFlags(hipDeviceMallocUncached): the buffers other ranks' kernels write into while this
// rank's kernel runs) comes from a process-wide pool and is never handed back to the runtime while the process lives.  Measured
// (round 6, tests/test_gpu_sharded.py in one process): after ~50 engines that each freed five such buffers, ordinary hipMalloc'ed
// arrays of a LATER engine read back as zeros although its kernels had written them -- an address range that changes its memory
// type between allocations is not safe here; with the pool (or without uncached memory) the same sequence is clean.
struct UncachedPool {
    std::mutex m;
    std::multimap<size_t, void *> idle;      // capacity in bytes -> buffer
};
UncachedPool &uncached_pool() { static UncachedPool *p = new UncachedPool; return *p; }
int uncached_alloc(hs_engine *h, void **out, size_t bytes) {
    const size_t want = ((bytes ? bytes : 1) + 65535) & ~(size_t)65535;
    {
        UncachedPool &P = uncached_pool();
        std::lock_guard<std::mutex> g(P.m);
        auto it = P.idle.lower_bound(want);
        if (it != P.idle.end() && it->first <= 2 * want) {
            *out = it->second;
            h->uncached.emplace_back(it->second, it->first);
            P.idle.erase(it);
            return HS_OK;
        }
    }
    void *q = nullptr;
    const hipError_t e = hipExtMallocWithFlags(&q, want, hipDeviceMallocUncached);
    if (e != hipSuccess) return fail(h, HS_E_HIP, "hipExtMallocWithFlags(%zu B, uncached): %s", want, hipGetErrorString(e));
    h->uncached.emplace_back(q, want);
    *out = q;
    return HS_OK;
}
void uncached_release(hs_engine *h) {
    UncachedPool &P = uncached_pool();
    std::lock_guard<std::mutex> g(P.m);
    for (auto &b : h->uncached) P.idle.emplace(b.second, b.first);
    h->uncached.clear();
}

template <typename T>
int upload(hs_engine *h, const T **dst, const T *src, size_t n, T dflt) {
    T *d = nullptr;
    int rc = dev_alloc(h, &d, n);
    if (rc) return rc;
    std::vector<T> tmp;
    if (!src) { tmp.assign(n, dflt); src = tmp.data(); }
    hipError_t e = hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(h, HS_E_HIP, "hipMemcpy H2D: %s", hipGetErrorString(e));
    *dst = d;
    return HS_OK;
}

template <int C>
void launch_run(hs_engine *h, int64_t end_ns, int mode, int flags) {
    if (h->n_pass > 0) {      // tandem queues: the general-path instantiation, one pass (hs_station.hpp `trk`)
        hipLaunchKernelGGL((hs_station_run<C, true>), dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot,
                           h->cands, h->cfg.n_lp, end_ns, mode, flags);
        return;
    }
    if constexpr (C == 1) {
        if (!h->any_profile && !(flags & 512)) {     // producer / consumer wavefronts (debug flag 512: the one-role kernel)
            if (h->uni_grid && (flags & (1 << 20)) == 0)   // uniform entity kinds: compile-time predicates (hs_station.hpp HSG)
                hipLaunchKernelGGL((hs_station_run<1, false, true, true>), dim3(h->n_blocks), dim3(2 * kBlock), 0, h->stream, h->P,
                                   h->X, h->L, h->tot, h->cands, h->cfg.n_lp, end_ns, mode, flags);
            else
                hipLaunchKernelGGL((hs_station_run<1, false, true>), dim3(h->n_blocks), dim3(2 * kBlock), 0, h->stream, h->P, h->X,
                                   h->L, h->tot, h->cands, h->cfg.n_lp, end_ns, mode, flags);
            return;
        }
    }
    if (h->any_profile)
        hipLaunchKernelGGL((hs_station_run<C, true>), dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot,
                           h->cands, h->cfg.n_lp, end_ns, mode, flags);
    else
        hipLaunchKernelGGL((hs_station_run<C, false>), dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot,
                           h->cands, h->cfg.n_lp, end_ns, mode, flags);
}

// K lanes per LP when the grid is uniform and leaves lanes idle (hs_kernels_wide.hpp); debug flag 1 << 22 keeps the one-lane kernel,
// bits 24..27 force K = 1 << (value - 1)
int wide_lanes(const hs_engine *h) {
    if (h->is_net) return 0;      // (a network's stations run on the network engines, whatever their egress arrays said before hs_engine_set_network)
    if (h->C != 1 || !h->uni_grid || h->any_profile || h->cfg.mode != HS_MODE_SINGLE || !h->fresh || h->wide_ctl == nullptr) return 0;
    if (h->flags & ((1 << 22) | 1 | 512 | (1 << 20))) return 0;
    const int forced = (h->flags >> 24) & 0xf;
    if (forced) return forced == 7 ? 64 : forced == 8 ? 65 : (1 << (forced - 1)) <= 16 ? 1 << (forced - 1) : 16;   // 7 / 8: a wavefront per LP, 16 / 8 LPs per workgroup
    const bool wave_ok = h->cfg.horizon_ns < (1ll << 39);      // (the speculated whole-ns arrival steps: hs_kernels_wave.hpp)
    // Measured on MI355X (tools/wide_timing.py, profiles/r03_wide_timing.log; 60 s of the headline grid, kernel ms):
    //   n_lp      one lane   K = 4    K = 8    K = 16
    //    1 024     0.372     0.118    0.087    0.093
    //    4 096     0.378     0.138    0.129    0.177
    //    8 192     0.385     0.166    0.187    0.315
    //   16 384     0.388     0.231    0.344    0.579
    //   32 768     0.403     0.422    0.610    1.059
    // Fewer lanes per LP = less redundant work in the serial arrival chain, more = shorter steps: K = 8 while the device has SIMDs
    // to spare, K = 4 up to a quarter of the size at which a lane per LP fills the machine.
    if (h->device_lanes == 0) {                                                  // (asked once: it is a ~10 us host call, twice per step)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->cfg.device) != hipSuccess) return 0;
        h->device_lanes = (long long)prop.multiProcessorCount * 4 * 64;          // one wavefront per SIMD
    }
    const long long lanes = h->device_lanes;
    const long long n = h->cfg.n_lp;
    // Round 5: one wavefront per LP (hs_kernels_wave.hpp).  Measured (tools/wide_timing.py, profiles/r05_wide_timing.log; kernel ms incl.
    // hs_station_wide_finish): 1 024 LPs 0.056 (8 LPs per workgroup) / 0.068 (16) against K = 8: 0.087; 8 192: 0.106 against K = 4: 0.168;
    // 16 384: 0.17 against K = 4: 0.23; 32 768: 0.31 against one lane per LP: 0.40; 65 536: ~0.6 against 0.42 -- one lane per LP from there.
    // Later in round 5 (T without selects, bootstrap and fold once per LP on wavefront 0): 8 192: 0.089; 32 768: 0.27; 40 960: 0.34 against
    // 0.40; 45 056: 0.37 against 0.41; 49 152: 0.405 against 0.410 (the crossover); 65 536: 0.53 against 0.43.
    if (wave_ok && n * 16 <= lanes) return 65;  // <= 4 096 LPs: 8 LPs per workgroup (more workgroups than CUs)
    if (wave_ok && n * 3 <= lanes * 2) return 64;   // <= 43 690 LPs: 16 LPs per workgroup (a whole line of every record row)
    if (n * 16 <= lanes) return 8;              // <= 4 096 LPs on 256 CUs
    if (n * 4 <= lanes) return 4;               // <= 16 384 LPs
    return 0;
}
template <int K>
void launch_wide(hs_engine *h, int64_t end_ns) {
    const int n = h->cfg.n_lp, G = 64 / K, nb = (n + G - 1) / G;
    hipLaunchKernelGGL(hs_station_wide<K>, dim3(nb), dim3(kWideBlock), 0, h->stream, h->P, h->X, h->L, h->tot, h->cands, h->wide_ctl,
                       h->wide_bail, n, end_ns, h->flags);
    hipLaunchKernelGGL(hs_station_wide_finish, dim3(1), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot, h->cands, nb, h->wide_ctl,
                       h->wide_bail, n, end_ns, (const WavePart *)nullptr, (long long)INT64_MIN);
}
template <int NW>
void launch_wave(hs_engine *h, int64_t end_ns) {
    const int n = h->cfg.n_lp, nb = (n + NW - 1) / NW;
    const bool fresh = h->reset_pending;                      // the bootstrap inside the kernel, no hs_station_reset launch
    h->reset_pending = false;
    if (fresh)
        hipLaunchKernelGGL((hs_station_wave<NW, true>), dim3(nb), dim3(NW * 64), 0, h->stream, h->P, h->X, h->L, h->tot, h->cands, h->wide_ctl,
                           h->wide_bail, h->wave_parts, n, end_ns, h->flags, h->cfg.start_ns);
    else
        hipLaunchKernelGGL((hs_station_wave<NW, false>), dim3(nb), dim3(NW * 64), 0, h->stream, h->P, h->X, h->L, h->tot, h->cands, h->wide_ctl,
                           h->wide_bail, h->wave_parts, n, end_ns, h->flags, h->cfg.start_ns);
    hipLaunchKernelGGL(hs_station_wide_finish, dim3(1), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot, h->cands, nb, h->wide_ctl,
                       h->wide_bail, n, end_ns, (const WavePart *)h->wave_parts, fresh ? (long long)h->cfg.start_ns : (long long)INT64_MIN);
}

int ensure_reset(hs_engine *h);
int launch_run_dispatch(hs_engine *h, int64_t end_ns) {
    if (h->exact_only) return HS_OK;     // (the single-heap loop has run the whole window: launch_prologue)
    const int K = wide_lanes(h);
    // (only hs_station_wave performs a deferred bootstrap itself; ADVICE r5: a reset launch that fails is the run's failure)
    if (K != 64 && K != 65) { const int rcr = ensure_reset(h); if (rcr) return rcr; }
    h->fresh = false;
    switch (K) {
        case 4: launch_wide<4>(h, end_ns); return HS_OK;
        case 8: launch_wide<8>(h, end_ns); return HS_OK;
        case 16: launch_wide<16>(h, end_ns); return HS_OK;
        case 64: launch_wave<16>(h, end_ns); return HS_OK;
        case 65: launch_wave<8>(h, end_ns); return HS_OK;
        default: break;
    }
    // Tandem queues: one launch per pass, upstream Servers first; the last one elects the event beyond end_ns among ALL LPs
    // (an internal mode 2 = "no election" for the others).  Everything else: one launch.
    const int passes = h->n_pass > 0 ? h->n_pass : 1;
    for (int p = 0; p < passes; ++p) {
        const int mode = (p == passes - 1) ? h->cfg.mode : 2;
        const int flags = h->n_pass > 0 ? (h->flags | ((p + 1) << 28)) : h->flags;
        switch (h->C) {
            case 1: launch_run<1>(h, end_ns, mode, flags); break;
            case 2: launch_run<2>(h, end_ns, mode, flags); break;
            case 4: launch_run<4>(h, end_ns, mode, flags); break;
            case 8: launch_run<8>(h, end_ns, mode, flags); break;
            case 16: launch_run<16>(h, end_ns, mode, flags); break;
            default: launch_run<32>(h, end_ns, mode, flags); break;     // (departure slots beyond 16 live in scratch: correct, not fast)
        }
    }
    return HS_OK;
}

template <int C>
void launch_net(hs_engine *h, int64_t wend, int win, int flags) {
    hipLaunchKernelGGL(hs_net_window<C>, dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L,
                       h->tot, h->cands, h->cfg.n_lp, wend, win, flags, h->SC);
}
void launch_net_dispatch(hs_engine *h, int64_t wend, int win, int flags) {
    switch (h->C) {
        case 1: launch_net<1>(h, wend, win, flags); break;
        case 2: launch_net<2>(h, wend, win, flags); break;
        default: launch_net<4>(h, wend, win, flags); break;
    }
}

// windows: finish the timestamp group the last run_until's election stopped inside (hs_kernels.hpp hs_net_resume)
void launch_net_resume(hs_engine *h, const NetState &NX, int send_idx) {
    switch (h->C) {
        case 1: hipLaunchKernelGGL(hs_net_resume<1>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, NX, h->L, h->tot, h->cfg.n_lp, send_idx, h->SC); break;
        case 2: hipLaunchKernelGGL(hs_net_resume<2>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, NX, h->L, h->tot, h->cfg.n_lp, send_idx, h->SC); break;
        default: hipLaunchKernelGGL(hs_net_resume<4>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, NX, h->L, h->tot, h->cfg.n_lp, send_idx, h->SC); break;
    }
    h->launches++;
}

template <int C>
hipError_t launch_async(hs_engine *h, int64_t end_ns, NetState NX) {
    int n = h->cfg.n_lp, flags = h->flags & (1 | 64 | 128 | 1024 | 0xff00 | (1 << 21)), lanes = h->async_lanes;
    const int per_block = (kBlock / 64) * lanes;
    int max_iters = h->round_iters;
    void *args[] = {&h->P, &h->NP, &h->X, &NX, &h->L, &h->tot, &n, &end_ns, &flags, &h->SC, &lanes, &max_iters};
    const void *fn = h->net_pf ? (const void *)hs_net_async<C, true> : (const void *)hs_net_async<C, false>;
    if constexpr (C == 1) {    // uniform entity kinds: the specialised instantiation (debug flag 1 << 20 keeps the generic one)
        if (h->net_uni && !h->net_pf && !h->net_global && (h->flags & ((1 << 20) | 1 | 32)) == 0 && h->round_iters == 0 && lanes == 64)
            fn = (const void *)hs_net_async<1, false, true>;
    }
    // LIVE exchange: the ranks' kernels wait for one another, and several of them may share this device (shards of one process, or
    // of several processes) -- a plain launch each: what makes them co-resident is that together they have no more workgroups than
    // the device has CUs (hs_engine_shard_live_run checks its own share)
    if (h->SC.live) return hipLaunchKernel(fn, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), args, 0, h->stream);
    return hipLaunchCooperativeKernel(fn, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), args, 0, h->stream);
}
// one SEGMENT of a network that does not fit one cooperative launch: stations [lp0, lp0 + blocks x 256) for `iters` iterations of
// the generic asynchronous kernel (the segment's first station travels in bits 8.. of `lanes`)
template <int C>
hipError_t launch_async_segment(hs_engine *h, int64_t end_ns, NetState NX, int lp0, int blocks, int iters) {
    int n = h->cfg.n_lp, flags = h->flags & (1 | 64 | 128 | 1024 | 0xff00 | (1 << 21)), lanes = 64 | ((lp0 / kBlock) << 8), max_iters = iters;
    void *args[] = {&h->P, &h->NP, &h->X, &NX, &h->L, &h->tot, &n, &end_ns, &flags, &h->SC, &lanes, &max_iters};
    const void *fn = h->net_pf ? (const void *)hs_net_async<C, true> : (const void *)hs_net_async<C, false>;
    return hipLaunchCooperativeKernel(fn, dim3((unsigned)blocks), dim3(kBlock), args, 0, h->stream);
}
template <int C, bool PF>
int async_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hs_net_async<C, PF>, kBlock, 0) != hipSuccess) return 0;
    return nb;
}

// The whole run in one launch of hs_net_async + the final launch of hs_net_window (election of the one event beyond
// end_ns).  Returns 1 if it ran, 0 if the network has to use the windowed engine, < 0 on error.
int try_run_net_whole(hs_engine *h, int64_t end_ns);
int ensure_async_fit(hs_engine *h) {
    if (h->async_fit < 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->cfg.device) != hipSuccess) return 0;
        const int per_cu = h->net_pf ? (h->C == 1 ? async_blocks_per_cu<1, true>() : h->C == 2 ? async_blocks_per_cu<2, true>() : async_blocks_per_cu<4, true>())
                                     : (h->C == 1 ? async_blocks_per_cu<1, false>() : h->C == 2 ? async_blocks_per_cu<2, false>() : async_blocks_per_cu<4, false>());
        const long long resident = prop.cooperativeLaunch ? (long long)per_cu * prop.multiProcessorCount : 0;   // workgroups
        h->async_resident_blocks = resident;
        h->async_fit = 0;
        for (int lanes = (h->flags & 32) ? 16 : 64; lanes <= 64; lanes *= 2) {
            const int per_block = (kBlock / 64) * lanes;
            if (((long long)h->cfg.n_lp + per_block - 1) / per_block <= resident) { h->async_lanes = lanes; h->async_fit = 1; break; }
        }
    }
    return h->async_fit;
}
// A network with MORE stations than one cooperative launch holds (round 5, VERDICT r4 missing 5: at 65 537 stations hs_engine_run_until
// used to fall to one launch per smallest link latency, 60 000 launches per 60 s).  The asynchronous protocol does not care who runs
// when: a station only ever advances below the bounds its incoming links carry, and a neighbour that is not running simply does
// not raise its bound.  So contiguous SEGMENTS of stations take turns on the device -- every launch advances one segment by
// `kSegmentIters` iterations against the bounds and queues the others left in global memory -- until no station has work at or
// before end_ns (Totals::not_done, read once per sweep).  The same bits as one launch (tests/test_gpu_ring.py); what the
// partitioned run does across GPUs (sharded.py), without outboxes: all segments share this engine's link queues.
constexpr int kSegmentIters = 64;
int run_net_segments(hs_engine *h, int64_t end_ns, NetState NX, long long resident_blocks) {
    const int n = h->cfg.n_lp;
    const long long total_blocks = ((long long)n + kBlock - 1) / kBlock;
    const int nseg = (int)((total_blocks + resident_blocks - 1) / resident_blocks);
    const long long per_seg = (total_blocks + nseg - 1) / nseg;              // balanced: no short last segment
    for (int sweep = 0; sweep < (1 << 20); ++sweep) {
        HS_HIP(h, hipMemsetAsync(&h->tot->not_done, 0, sizeof(unsigned long long), h->stream));
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const long long b0 = (long long)sgm * per_seg, b1 = std::min<long long>(total_blocks, b0 + per_seg);
            if (b1 <= b0) continue;
            const hipError_t e = h->C == 1 ? launch_async_segment<1>(h, end_ns, NX, (int)(b0 * kBlock), (int)(b1 - b0), kSegmentIters)
                               : h->C == 2 ? launch_async_segment<2>(h, end_ns, NX, (int)(b0 * kBlock), (int)(b1 - b0), kSegmentIters)
                                           : launch_async_segment<4>(h, end_ns, NX, (int)(b0 * kBlock), (int)(b1 - b0), kSegmentIters);
            if (e != hipSuccess) return fail(h, HS_E_HIP, "cooperative launch of a network segment failed: %s", hipGetErrorString(e));
            h->launches++;
        }
        unsigned long long left[2] = {0, 0};
        int ov = 0;
        HS_HIP(h, hipMemcpyAsync(&left[0], &h->tot->not_done, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HS_HIP(h, hipMemcpyAsync(&ov, &h->tot->overflow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HS_HIP(h, hipStreamSynchronize(h->stream));
        if (left[0] == 0ull || ov != 0) return HS_OK;
    }
    return fail(h, HS_E_HIP, "the segmented asynchronous run did not finish");
}

int try_run_net_whole(hs_engine *h, int64_t end_ns) {
    if (!h->async_ok || (h->flags & 16)) return 0;
    if (!ensure_async_fit(h)) {
        // more stations than one cooperative launch holds: segments take turns (debug flag 1 << 23: the window protocol instead)
        if ((h->flags & (1 << 23)) || h->async_resident_blocks < 1) return 0;
        NetState NXs = h->NX;
        NXs.aq_on = 1;
        if (h->net_resume) launch_net_resume(h, NXs, 0);
        const int rc = run_net_segments(h, end_ns, NXs, h->async_resident_blocks);
        if (rc) return rc;
        const NetState keep = h->NX;
        h->NX = NXs;
        launch_net_dispatch(h, end_ns, 1, (h->flags & 1) | 2 | 8);       // FINAL: leftover queue entries, overshoot
        h->NX = keep;
        HS_HIP(h, hipGetLastError());
        h->launches += 1;
        h->net_ran = true;
        return 1;
    }
    NetState NX = h->NX;
    NX.aq_on = 1;
    if (h->net_resume) launch_net_resume(h, NX, 0);
    hipError_t e = h->C == 1 ? launch_async<1>(h, end_ns, NX) : h->C == 2 ? launch_async<2>(h, end_ns, NX) : launch_async<4>(h, end_ns, NX);
    if (e != hipSuccess) {      // not co-resident after all
        (void)hipGetLastError();
        h->async_fit = 0;
        if (h->net_resume) return fail(h, HS_E_HIP, "the cooperative launch of a resumed window failed: %s", hipGetErrorString(e));
        return 0;               // ... windows
    }
    const NetState keep = h->NX;
    h->NX = NX;
    launch_net_dispatch(h, end_ns, 1, (h->flags & 1) | 2 | 8);       // FINAL: leftover queue entries, overshoot
    h->NX = keep;
    HS_HIP(h, hipGetLastError());
    h->launches += 2;
    h->net_ran = true;
    return 1;
}

// EXECUTE / EXCHANGE / ADVANCE (parallel/coordinator.py:87-124) as a stream of window launches
int run_net_async(hs_engine *h, int64_t end_ns) {
    { const int rcr = ensure_reset(h); if (rcr) return rcr; }     // (never a network run on a deferred bootstrap)
    {
        const int whole = try_run_net_whole(h, end_ns);
        if (whole != 0) return whole < 0 ? whole : HS_OK;
    }
    const int64_t W = h->window_ns;
    int64_t t0 = h->cfg.start_ns;
    int win = 0;
    if (h->net_resume) {
        // windows: the conservative windows continue behind the last end (everything at or before it has happened); what the
        // finished group sends waits in the incoming bags the first window merges (parity 1)
        launch_net_resume(h, h->NX, 1);
        t0 = h->net_resume_from + 1;
    }
    for (;;) {
        int64_t wend = t0 + W - 1;
        if (wend >= end_ns || wend < t0) wend = end_ns;
        launch_net_dispatch(h, wend, win, h->flags & 1);
        ++win;
        if (wend >= end_ns) break;
        t0 = wend + 1;
    }
    launch_net_dispatch(h, end_ns, win, (h->flags & 1) | 2);   // FINAL: merge the last window's messages, overshoot
    HS_HIP(h, hipGetLastError());
    h->launches += win + 1;
    h->net_ran = true;
    return HS_OK;
}

// the reset kernel itself (the Simulation.__init__ bootstrap on the device)
int launch_reset(hs_engine *h);
int ensure_reset(hs_engine *h) {
    if (!h->reset_pending) return HS_OK;
    h->reset_pending = false;
    return launch_reset(h);
}
int do_reset_async(hs_engine *h) {
    // An engine whose next run is hs_station_wave (wide_lanes() == 64 / 65 once `fresh`): that kernel performs the bootstrap itself
    // (FRESH instantiation) and hs_station_wide_finish starts the totals -- no reset launch, no state round trip through HBM.
    // Debug flag 1 << 29 keeps the reset kernel.
    {
        const bool was_fresh = h->fresh;
        h->fresh = true;
        const int K = (h->flags & (1 << 29)) ? 0 : wide_lanes(h);
        h->fresh = was_fresh;
        if ((K == 64 || K == 65) && !h->exact) {
            h->reset_pending = true;
            h->initialised = true;
            h->net_ran = false;
            h->net_last_end = INT64_MIN;
            h->tot_seen_valid = false;
            h->fresh = true;
            h->window_ends.clear();
            h->pending_async = false;
            return HS_OK;
        }
    }
    h->reset_pending = false;
    return launch_reset(h);
}
int launch_reset(hs_engine *h) {
    if (h->n_tab_rows > 0 && !h->tables_built) {      // the tick tables (hs_tables.hpp): once, BEFORE the bootstrap reads tick 0
        HS_HIP(h, tick_tables_launch(h->stream, h->tab_rows, h->n_tab_rows, h->cfg.start_ns, h->cfg.horizon_ns, h->tab_cap,
                                     h->tab_times, h->tab_count, h->tab_status, h->lane_budget, false));
        h->tables_built = true;
    }
    if (h->any_profile)
        hipLaunchKernelGGL(hs_station_reset<true>, dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->tot,
                           h->cfg.n_lp, h->C, h->cfg.start_ns, h->NX, h->is_net ? h->NP.n_links : 0);
    else
        hipLaunchKernelGGL(hs_station_reset<false>, dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->tot,
                           h->cfg.n_lp, h->C, h->cfg.start_ns, h->NX, h->is_net ? h->NP.n_links : 0);
    HS_HIP(h, hipGetLastError());
    if (h->exact) {            // the prologue starts over: empty heap, both counters at 0
        if (h->XI.per_lp) HS_HIP(h, hipMemsetAsync(h->xs, 0, ((size_t)h->cfg.n_lp + 1) * sizeof(XState), h->stream));
        HS_HIP(h, hipMemcpyAsync(h->xs, &h->xs_host, sizeof(XState), hipMemcpyHostToDevice, h->stream));
        HS_HIP(h, hipMemsetAsync(h->xs_host.qhead, 0xff, (size_t)h->cfg.n_lp * sizeof(int32_t), h->stream));
        HS_HIP(h, hipMemsetAsync(h->xs_host.qtail, 0xff, (size_t)h->cfg.n_lp * sizeof(int32_t), h->stream));
    }
    if (h->is_net && h->loss_host.send_log_n) HS_HIP(h, hipMemsetAsync(h->loss_host.send_log_n, 0, sizeof(unsigned long long), h->stream));
    h->initialised = true;
    h->net_ran = false;
    h->net_last_end = INT64_MIN;
    h->tot_seen_valid = false;
    h->fresh = true;
    h->window_ends.clear();
    h->pending_async = false;      // (a run enqueued before this reset has nothing left to finalise)
    return HS_OK;
}

bool lazy_active(const hs_engine *h) {
    // (a shard of a partitioned network is driven window by window / round by round from outside: it keeps the eager prologue)
    return h->lazy_prologue && !h->lazy_failed && !(h->is_net && h->net_global) && !h->exact_only && (h->flags & (1 << 16)) == 0;
}

// the prologue of a run (hs_exact.hpp); a no-op launch once it has handed over
int launch_prologue(hs_engine *h, int64_t end_ns) {
    if (!h->exact || (h->flags & 256)) return HS_OK;
    if (!h->exact_prologue && !h->exact_only) return HS_OK;
    if (h->lazy_prologue) h->P.sched_idx = lazy_active(h) ? nullptr : h->XI.sched_idx;   // (the prologue writes them)
    if (lazy_active(h)) return HS_OK;
    h->XI.no_handover = h->exact_only ? 1 : 0;
    hipLaunchKernelGGL(hs_exact_run, dim3(h->XI.per_lp ? (unsigned)((h->cfg.n_lp + 63) / 64) : 1u), dim3(64), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L, h->tot, h->xs, h->XI,
                       h->cfg.n_lp, h->C, h->is_net ? 1 : 0, h->is_net ? h->NP.n_links : 0, h->cfg.start_ns, end_ns);
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

// the ranks' exchange buffers as this process addresses them -> device arrays for hs_shard_push
int ipc_set_peers(hs_engine *h, const std::vector<int64_t *> &pin, const std::vector<int64_t *> &pbd) {
    const int world = h->SC.world;
    int rc;
    if ((rc = dev_alloc(h, &h->peer_inbox_dev, (size_t)world))) return rc;
    if ((rc = dev_alloc(h, &h->peer_bounds_dev, (size_t)world))) return rc;
    HS_HIP(h, hipMemcpy(h->peer_inbox_dev, pin.data(), (size_t)world * sizeof(int64_t *), hipMemcpyHostToDevice));
    HS_HIP(h, hipMemcpy(h->peer_bounds_dev, pbd.data(), (size_t)world * sizeof(int64_t *), hipMemcpyHostToDevice));
    h->ipc_ready = true;
    h->ipc_parity = 0;
    return HS_OK;
}

}  // namespace

extern "C" {

int hs_abi_version(void) { return HS_ABI_VERSION; }

int hs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *hs_last_error(const hs_engine *h) { return h ? h->error.c_str() : g_global_error.c_str(); }
const char *hs_last_global_error(void) { return g_global_error.c_str(); }

int hs_engine_create(const hs_config *cfg, hs_engine **out) {
    if (!cfg || !out) return fail(nullptr, HS_E_INVALID, "hs_engine_create: null argument");
    if (cfg->struct_size != sizeof(hs_config))
        return fail(nullptr, HS_E_INVALID, "hs_engine_create: hs_config size mismatch (ABI %d)", HS_ABI_VERSION);
    if (cfg->n_lp <= 0) return fail(nullptr, HS_E_INVALID, "hs_engine_create: n_lp must be > 0");
    if (cfg->mode != HS_MODE_SINGLE && cfg->mode != HS_MODE_REPLICAS)
        return fail(nullptr, HS_E_INVALID, "hs_engine_create: unknown mode %d", cfg->mode);
    if (cfg->horizon_ns < cfg->start_ns)
        return fail(nullptr, HS_E_INVALID, "hs_engine_create: horizon_ns precedes start_ns");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, HS_E_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, HS_E_INVALID, "device ordinal %d out of range (%d devices)", cfg->device, ndev);
    hs_engine *h = new (std::nothrow) hs_engine();
    if (!h) return fail(nullptr, HS_E_INVALID, "out of host memory");
    h->cfg = *cfg;
    hipError_t e = hipSetDevice(cfg->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_a);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_b);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_k0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_k1);
    if (e != hipSuccess) {
        int rc = fail(nullptr, HS_E_HIP, "device setup: %s", hipGetErrorString(e));
        delete h;
        return rc;
    }
    h->n_blocks = (cfg->n_lp + kBlock - 1) / kBlock;
    h->own_stream = h->stream;
    *out = h;
    return HS_OK;
}

int hs_engine_set_stations(hs_engine *h, const hs_stations *st) {
    if (!h || !st) return fail(h, HS_E_INVALID, "hs_engine_set_stations: null argument");
    if (h->have_stations) return fail(h, HS_E_STATE, "stations already set");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int n = h->cfg.n_lp;
    // ---- validation (mirrors the reference constructors' ValueErrors) and sizing
    int maxc = 1;
    double max_mean_records = 0.0;
    bool any_source = false;
    const double horizon_s = (double)(h->cfg.horizon_ns - h->cfg.start_ns) / 1e9;
    // several Sources feeding one Server: slots 1 .. kMaxXSrc of an LP (include/hs_engine.h `src_more_kind`)
    std::vector<uint8_t> xk((size_t)n * kMaxXSrc, (uint8_t)0);
    std::vector<double> xr((size_t)n * kMaxXSrc, 1.0), xsum((size_t)n, 0.0);
    std::vector<int64_t> xstop((size_t)n * kMaxXSrc, (int64_t)-1);
    int64_t n_xsrc_total = 0;
    for (int j = 0; j < kMaxXSrc && st->src_more_kind; ++j)
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)j * n + i;
            const int k = st->src_more_kind[o];
            if (k == HS_SRC_NONE) continue;
            if (k != HS_SRC_POISSON && k != HS_SRC_CONSTANT) return fail(h, HS_E_INVALID, "LP %d: unknown source kind %d in slot %d", i, k, j + 1);
            const bool prev = j == 0 ? (st->src_kind ? st->src_kind[i] : HS_SRC_POISSON) != HS_SRC_NONE : xk[(size_t)(j - 1) * n + i] != 0;
            if (!prev) return fail(h, HS_E_INVALID, "LP %d: source slots must be filled from 0", i);
            if ((st->svc_kind ? st->svc_kind[i] : HS_LAT_CONSTANT) == HS_LAT_NO_SERVER)
                return fail(h, HS_E_UNSUPPORTED, "LP %d: several Sources need a Server to feed", i);
            if (st->src_profile_kind && st->src_profile_kind[i] != 0)
                return fail(h, HS_E_UNSUPPORTED, "LP %d: a time-varying Source next to further Sources is not lowered", i);
            if (!st->src_more_rate) return fail(h, HS_E_INVALID, "src_more_rate is required with src_more_kind");
            const double r = st->src_more_rate[o];
            if (!(r > 0.0) || !std::isfinite(r)) return fail(h, HS_E_INVALID, "LP %d: source rate must be > 0 (got %g)", i, r);
            if (r > 1e8) return fail(h, HS_E_UNSUPPORTED, "LP %d: source rate %g above 1e8/s is not supported", i, r);
            xk[o] = (uint8_t)k; xr[o] = r; xsum[(size_t)i] += r;
            if (st->src_more_stop_after_ns) xstop[o] = st->src_more_stop_after_ns[o];
            ++n_xsrc_total;
        }
    h->any_xsrc = n_xsrc_total > 0;
    // every LP is Source.poisson -> Server(Exp, c = 1, unbounded queue), nothing stops: half of what the specialised network
    // kernel (NetStation<.., UNI>) assumes; hs_engine_set_network checks the routers and links
    h->uni_stations = true;
    for (int i = 0; i < n; ++i) {
        if ((st->src_kind ? st->src_kind[i] : HS_SRC_POISSON) != HS_SRC_POISSON || (st->concurrency ? st->concurrency[i] : 1) != 1 ||
            (st->svc_kind ? st->svc_kind[i] : HS_LAT_CONSTANT) != HS_LAT_EXPONENTIAL || (st->queue_cap ? st->queue_cap[i] : -1) >= 0 ||
            (st->src_stop_after_ns ? st->src_stop_after_ns[i] : -1) >= 0) { h->uni_stations = false; break; }
    }
    h->uni_grid = h->uni_stations;
    for (int i = 0; i < n && h->uni_grid; ++i) if ((st->egress ? st->egress[i] : HS_EGRESS_SINK) != HS_EGRESS_SINK) h->uni_grid = false;
    // (the uniform-kind kernels keep times as exact binary64 integers: whole ns in [0, 2^51), hs_device.hpp ns_from_seconds_d)
    h->f64_times = h->cfg.start_ns >= 0 && h->cfg.horizon_ns < (1ll << 51);
    for (int i = 0; i < n && h->f64_times; ++i)     // (one draw is at most 36.8 means / inter-arrival times: everything stays below 2^52 ns)
        if (!((st->svc_mean_s ? st->svc_mean_s[i] : 0.01) < 1e4) || !((st->src_rate ? st->src_rate[i] : 1.0) > 1e-3)) h->f64_times = false;
    if (!h->f64_times) h->uni_grid = false;
    if (h->any_xsrc) h->any_profile = true;                         // such LPs run on the general-path instantiation
    for (int i = 0; i < n; ++i) {
        const int sk = st->src_kind ? st->src_kind[i] : HS_SRC_POISSON;
        if (sk < 0 || sk > 2) return fail(h, HS_E_INVALID, "LP %d: unknown source kind %d", i, sk);
        if (sk != HS_SRC_NONE) {
            any_source = true;
            if (!st->src_rate) return fail(h, HS_E_INVALID, "src_rate is required when sources exist");
            const double r = st->src_rate[i];
            if (!(r > 0.0) || !std::isfinite(r))
                return fail(h, HS_E_INVALID, "LP %d: source rate must be > 0 (got %g)", i, r);
            if (r > 1e8) return fail(h, HS_E_UNSUPPORTED, "LP %d: source rate %g above 1e8/s is not supported", i, r);
            const double m = (r + xsum[(size_t)i]) * horizon_s;
            if (m > max_mean_records) max_mean_records = m;
        }
        const int c = st->concurrency ? st->concurrency[i] : 1;
        if (c < 1) return fail(h, HS_E_INVALID, "LP %d: max_concurrent must be >= 1, got %d", i, c);
        if (c > 32) return fail(h, HS_E_UNSUPPORTED, "LP %d: concurrency %d > 32 is not lowered yet", i, c);
        if (c > maxc) maxc = c;
        const int vk = st->svc_kind ? st->svc_kind[i] : HS_LAT_CONSTANT;
        if (vk != HS_LAT_EXPONENTIAL && vk != HS_LAT_CONSTANT && vk != HS_LAT_NO_SERVER)
            return fail(h, HS_E_UNSUPPORTED, "LP %d: service distribution kind %d is not lowered", i, vk);
        if (vk == HS_LAT_NO_SERVER && sk == HS_SRC_NONE)
            return fail(h, HS_E_INVALID, "LP %d: neither a Source nor a Server", i);
        const double mean = st->svc_mean_s ? st->svc_mean_s[i] : 0.01;
        if (!(mean >= 0.0) || !std::isfinite(mean)) return fail(h, HS_E_INVALID, "LP %d: bad service mean %g", i, mean);
        if (vk == HS_LAT_EXPONENTIAL && !(mean > 0.0))
            return fail(h, HS_E_INVALID, "LP %d: exponential service needs mean > 0", i);
        const int eg = st->egress ? st->egress[i] : HS_EGRESS_SINK;
        if (eg != HS_EGRESS_NONE && eg != HS_EGRESS_SINK && eg != HS_EGRESS_SERVER)
            return fail(h, HS_E_UNSUPPORTED, "LP %d: egress kind %d is not lowered", i, eg);
    }
    // Tandem queues: Server(downstream=<Server>) (components/server/server.py:271-272).  up[d] = the LP that forwards to LP d;
    // an LP's pass = its distance from the head of its chain (hs_station.hpp `trk`).
    std::vector<int32_t> tandem;                     // hs_tables.hpp TickTables::tandem: kMaxUp upstream rows, the pass, the downstream LP
    const size_t rowP = (size_t)kMaxUp * n, rowD = (size_t)(kMaxUp + 1) * n;
    bool fan_in = false;                             // more than kMaxUp Servers forward to one: no passes, the single-heap loop
    for (int i = 0; i < n; ++i) {
        if ((st->egress ? st->egress[i] : HS_EGRESS_SINK) != HS_EGRESS_SERVER) continue;
        if (tandem.empty()) { tandem.assign((size_t)(kMaxUp + 2) * n, -1); for (int k = 0; k < n; ++k) tandem[rowP + k] = 0; }
        if (!st->downstream_lp) return fail(h, HS_E_INVALID, "downstream_lp is required with HS_EGRESS_SERVER");
        const int d = st->downstream_lp[i];
        if (d < 0 || d >= n || d == i) return fail(h, HS_E_INVALID, "LP %d: downstream_lp %d is not another LP of this engine", i, d);
        if ((st->svc_kind ? st->svc_kind[i] : HS_LAT_CONSTANT) == HS_LAT_NO_SERVER || (st->svc_kind ? st->svc_kind[d] : HS_LAT_CONSTANT) == HS_LAT_NO_SERVER)
            return fail(h, HS_E_INVALID, "LP %d: HS_EGRESS_SERVER connects two Servers", i);
        int u = 0;
        while (u < kMaxUp && tandem[(size_t)u * n + d] >= 0) ++u;
        if (u == kMaxUp) fan_in = true;
        else tandem[(size_t)u * n + d] = i;
        tandem[rowD + i] = d;
    }
    if (!tandem.empty()) {
        if (h->cfg.mode != HS_MODE_SINGLE) return fail(h, HS_E_UNSUPPORTED, "tandem queues (HS_EGRESS_SERVER) need HS_MODE_SINGLE: the LPs of a chain are one Simulation");
        for (int i = 0; i < n; ++i) {                                   // no cycles of Servers (zero-length services would never end)
            int steps = 0;
            for (int d = tandem[rowD + i]; d >= 0; d = tandem[rowD + d])
                if (++steps > n) return fail(h, HS_E_UNSUPPORTED, "LP %d: a cycle of Servers (downstream of downstream ... of itself) is not lowered", i);
        }
        // an LP's pass = the longest chain of Servers above it (acyclic: relax until nothing moves)
        int max_pass = 0;
        for (int round = 0; round < n; ++round) {
            bool changed = false;
            for (int i = 0; i < n; ++i) {
                const int d = tandem[rowD + i];
                if (d >= 0 && tandem[rowP + d] < tandem[rowP + i] + 1) { tandem[rowP + d] = tandem[rowP + i] + 1; changed = true; }
            }
            if (!changed) break;
        }
        for (int i = 0; i < n; ++i) max_pass = std::max(max_pass, (int)tandem[rowP + i]);
        if (max_pass > 6) {
            if (!fan_in) return fail(h, HS_E_UNSUPPORTED, "more than 7 Servers in a row are not lowered");
            max_pass = 6;                                                // (fan-in beyond kMaxUp runs on the single heap anyway)
        }
        h->n_pass = max_pass + 1;
        h->tandem_fan_in = fan_in;
        // what a Server behind Servers can admit: its own Sources' ticks plus everything upstream (sizes the record logs)
        std::vector<double> flow((size_t)n, 0.0);
        for (int i = 0; i < n; ++i)
            flow[(size_t)i] = ((st->src_kind ? st->src_kind[i] : HS_SRC_POISSON) != HS_SRC_NONE ? st->src_rate[i] : 0.0) + xsum[(size_t)i];
        for (int round = 0; round < n; ++round) {                       // (acyclic: settles after as many rounds as the longest chain)
            std::vector<double> in((size_t)n, 0.0);
            for (int i = 0; i < n; ++i) { const int d = tandem[rowD + i]; if (d >= 0) in[(size_t)d] += flow[(size_t)i]; }
            bool changed = false;
            for (int i = 0; i < n; ++i) {
                const double f = ((st->src_kind ? st->src_kind[i] : HS_SRC_POISSON) != HS_SRC_NONE ? st->src_rate[i] : 0.0) + xsum[(size_t)i] + in[(size_t)i];
                if (f != flow[(size_t)i]) { flow[(size_t)i] = f; changed = true; }
            }
            if (!changed) break;
        }
        for (int i = 0; i < n; ++i) if (flow[(size_t)i] * horizon_s > max_mean_records) max_mean_records = flow[(size_t)i] * horizon_s;
        h->any_profile = true;                                       // the general-path instantiation
        h->uni_grid = false;
    }
    (void)any_source;
    // time-varying profiles (load/profile.py:52-113); src_rate of such a source is its PEAK rate (it sizes the logs)
    std::vector<uint8_t> pk((size_t)n, (uint8_t)0);
    std::vector<double> pp((size_t)n * 4, 0.0);
    for (int i = 0; i < n && st->src_profile_kind; ++i) {
        const int k = st->src_profile_kind[i];
        if (k == 0) continue;
        if (k != 1 && k != 2) return fail(h, HS_E_UNSUPPORTED, "LP %d: profile kind %d is not lowered", i, k);
        if (!st->src_profile_params) return fail(h, HS_E_INVALID, "src_profile_params is required with src_profile_kind");
        const double *q = st->src_profile_params + 4 * (size_t)i;
        for (int j = 0; j < 4; ++j) {
            if (!std::isfinite(q[j]) || q[j] < 0.0) return fail(h, HS_E_INVALID, "LP %d: bad profile parameter %g", i, q[j]);
            pp[(size_t)j * n + i] = q[j];
        }
        if (k == 1 && !(q[0] > 0.0)) return fail(h, HS_E_INVALID, "LP %d: LinearRampProfile needs duration_s > 0", i);
        pk[(size_t)i] = (uint8_t)k;
        h->any_profile = true;
        h->any_timevarying = true;
    }
    // Probes (instrumentation/probe.py:81-164): up to kMaxProbes per LP (slot 0 = probe_metric, slots 1.. = probe_metric_more);
    // rate = 1.0 / interval as the reference computes it
    std::vector<uint8_t> pm((size_t)n * kMaxProbes, (uint8_t)255);
    std::vector<double> prate((size_t)n * kMaxProbes, 1.0);
    double min_interval = 0.0;
    int64_t n_prb_total = 0;
    for (int j = 0; j < kMaxProbes; ++j) {
        const uint8_t *pmj = j == 0 ? st->probe_metric : (st->probe_metric_more ? st->probe_metric_more + (size_t)(j - 1) * n : nullptr);
        const double *pij = j == 0 ? st->probe_interval_s : (st->probe_interval_more ? st->probe_interval_more + (size_t)(j - 1) * n : nullptr);
        for (int i = 0; i < n && pmj; ++i) {
            const int m = pmj[i];
            if (m == 255) continue;
            if (m < 0 || m > 6) return fail(h, HS_E_UNSUPPORTED, "LP %d: probe metric %d is not lowered", i, m);
            if (j > 0 && pm[(size_t)(j - 1) * n + i] == 255) return fail(h, HS_E_INVALID, "LP %d: probe slots must be filled from 0", i);
            if (!pij) return fail(h, HS_E_INVALID, "probe_interval_s is required with probe_metric");
            const double iv = pij[i];
            if (!(iv > 0.0) || !std::isfinite(iv)) return fail(h, HS_E_INVALID, "Probe interval must be positive.");   // probe.py:29-30
            pm[(size_t)j * n + i] = (uint8_t)m;
            prate[(size_t)j * n + i] = 1.0 / iv;
            if (min_interval == 0.0 || iv < min_interval) min_interval = iv;
            h->any_probe = true;
            if (j + 1 > h->n_probe_slots) h->n_probe_slots = j + 1;
            ++n_prb_total;
        }
    }
    if (h->any_probe) h->any_profile = true;                        // probes run on the general-path instantiation
    // Requests injected with Simulation.schedule(): validated here, run by the general-path instantiation too
    int64_t n_sched = 0, max_sched = 0;
    if (st->sched_off) {
        if (st->sched_off[0] != 0) return fail(h, HS_E_INVALID, "sched_off[0] must be 0");
        for (int i = 0; i < n; ++i) {
            const int64_t a = st->sched_off[i], b = st->sched_off[i + 1];
            if (b < a) return fail(h, HS_E_INVALID, "sched_off must not decrease (LP %d)", i);
            if (b > a && !st->sched_time_ns) return fail(h, HS_E_INVALID, "sched_time_ns is required with sched_off");
            if (b > a && (st->svc_kind ? st->svc_kind[i] : HS_LAT_CONSTANT) == HS_LAT_NO_SERVER)
                return fail(h, HS_E_UNSUPPORTED, "LP %d: scheduled Requests need a Server to receive them", i);
            for (int64_t k = a; k < b; ++k) {
                if (st->sched_time_ns[k] < h->cfg.start_ns)
                    return fail(h, HS_E_INVALID, "LP %d: scheduled time %lld ns lies before start_ns", i, (long long)st->sched_time_ns[k]);
                if (k > a && st->sched_time_ns[k] < st->sched_time_ns[k - 1])
                    return fail(h, HS_E_INVALID, "LP %d: scheduled times must be ascending", i);
            }
            if (b - a > max_sched) max_sched = b - a;
        }
        n_sched = st->sched_off[n];
        if (n_sched > 0) { h->any_profile = true; h->any_sched = true; }
    }
    h->C = maxc <= 1 ? 1 : maxc <= 2 ? 2 : maxc <= 4 ? 4 : maxc <= 8 ? 8 : maxc <= 16 ? 16 : 32;
    // (C == 1 reads a sink record's created_at from the admission log: the m-th completion is the m-th admission.  A Server that can
    // receive TWO Requests in one nanosecond while idle -- a forward beside its own Source's Request, two injected Requests, two
    // Sources in lock step -- delivers both and REJECTS the second at the worker (server.py:223-234), after which that shortcut is
    // off by one: explicit column, C >= 2.  Found by the tandem + probes sweep on a chain WITHOUT tandem queues: two Requests
    // injected at one instant into an idle Server.)
    if ((!tandem.empty() || n_sched > 0 || h->any_xsrc) && h->C == 1) h->C = 2;
    int64_t cap = h->cfg.log_capacity;
    if (cap <= 0) {
        const double c = max_mean_records + 10.0 * std::sqrt(max_mean_records + 1.0) + 64.0 + (double)max_sched;
        cap = ((int64_t)c + 15) & ~(int64_t)15;
    }
    const double log_bytes = (double)n * (double)cap * 8.0 * (h->C > 1 ? 3.0 : 2.0);
    if (log_bytes > 200e9)
        return fail(h, HS_E_INVALID, "record logs would need %.1f GB (n_lp=%d, capacity=%lld)", log_bytes / 1e9, n,
                    (long long)cap);
    h->L.cap = cap;
    int rc;
    std::vector<uint64_t> dflt_base((size_t)n);
    for (int i = 0; i < n; ++i) dflt_base[(size_t)i] = h->cfg.lp_base + (uint64_t)i;
#define UP(field, src, T, d) if ((rc = upload<T>(h, &h->P.field, src, (size_t)n, d))) return rc
    UP(src_kind, st->src_kind, uint8_t, (uint8_t)HS_SRC_POISSON);
    UP(src_rate, st->src_rate, double, 1.0);
    UP(src_stop, st->src_stop_after_ns, int64_t, (int64_t)-1);
    UP(conc, st->concurrency, int32_t, 1);
    UP(svc_kind, st->svc_kind, uint8_t, (uint8_t)HS_LAT_CONSTANT);
    UP(svc_mean, st->svc_mean_s, double, 0.01);
    UP(qcap, st->queue_cap, int64_t, (int64_t)-1);
    UP(egress, st->egress, uint8_t, (uint8_t)HS_EGRESS_SINK);
    UP(seed, st->seed, uint64_t, h->cfg.seed);
    if ((rc = upload<uint64_t>(h, &h->P.stream_base, st->stream_base ? st->stream_base : dflt_base.data(), (size_t)n, 0)))
        return rc;
#undef UP
    if ((rc = upload<uint8_t>(h, &h->P.prof_kind, pk.data(), (size_t)n, 0))) return rc;
    if ((rc = upload<double>(h, &h->P.prof_p, pp.data(), (size_t)n * 4, 0.0))) return rc;
    if ((rc = upload<uint8_t>(h, &h->P.probe_metric, pm.data(), (size_t)n * kMaxProbes, 255))) return rc;
    if ((rc = upload<double>(h, &h->P.probe_rate, prate.data(), (size_t)n * kMaxProbes, 1.0))) return rc;
    h->P.tabs = nullptr;
    const bool sched_single = h->any_sched && h->cfg.mode == HS_MODE_SINGLE;      // (the election's tie check: TickTables::standin_sched)
    if (h->any_timevarying || h->any_probe || !tandem.empty() || h->any_xsrc || sched_single) {
        // Tick tables (hs_tables.hpp): one row per time-varying Source (Poisson ones draw from their own arrival stream;
        // deterministic ones with equal parameters share a row) and one per distinct Probe interval (a Probe's tick times are a
        // property of (interval, start) alone).
        std::vector<TickRow> rows;
        std::vector<int32_t> srow((size_t)n, -1), prow((size_t)n * kMaxProbes, -1);
        for (int i = 0; i < n; ++i) {
            if (pk[(size_t)i] == 0) continue;
            TickRow r{};
            r.kind = pk[(size_t)i];
            r.poisson = (st->src_kind ? st->src_kind[i] : HS_SRC_POISSON) == HS_SRC_POISSON ? 1u : 0u;
            r.p0 = pp[(size_t)i]; r.p1 = pp[(size_t)n + i]; r.p2 = pp[(size_t)2 * n + i]; r.p3 = pp[(size_t)3 * n + i];
            r.owner = i;
            if (r.poisson) {
                r.seed = st->seed ? st->seed[i] : h->cfg.seed;
                r.sid = stream_id(st->stream_base ? st->stream_base[i] : dflt_base[(size_t)i], kStreamArrival);
            }
            int32_t found = -1;
            if (!r.poisson)
                for (size_t q = 0; q < rows.size() && found < 0; ++q)
                    if (!rows[q].poisson && rows[q].kind == r.kind && rows[q].p0 == r.p0 && rows[q].p1 == r.p1 && rows[q].p2 == r.p2 &&
                        rows[q].p3 == r.p3) found = (int32_t)q;
            if (found < 0) { found = (int32_t)rows.size(); rows.push_back(r); }
            srow[(size_t)i] = found;
        }
        {
            std::vector<std::pair<double, int32_t>> seen;          // (rate, row) of the probe rows so far
            for (int j = 0; j < kMaxProbes; ++j)
                for (int i = 0; i < n; ++i) {
                    const size_t o = (size_t)j * n + i;
                    if (pm[o] == 255) continue;
                    int32_t found = -1;
                    for (const auto &pr : seen) if (pr.first == prate[o]) { found = pr.second; break; }
                    if (found < 0) {
                        TickRow r{};
                        r.kind = kProfGeneralConstant; r.poisson = 0; r.p0 = prate[o]; r.owner = i;
                        found = (int32_t)rows.size();
                        rows.push_back(r);
                        seen.emplace_back(prate[o], found);
                    }
                    prow[o] = found;
                }
        }
        // capacity: what a Source's logs are sized for (its peak rate) / the fastest Probe's ticks, + the two beyond the horizon
        int64_t tcap = h->any_timevarying ? cap + 2 : 0;
        if (h->any_probe) tcap = std::max<int64_t>(tcap, (int64_t)(horizon_s / min_interval) + 8 + 2);
        h->tab_cap = tcap;
        h->n_tab_rows = (int)rows.size();
        if ((double)rows.size() * (double)tcap * 8.0 > 100e9)
            return fail(h, HS_E_INVALID, "tick tables would need %.1f GB", (double)rows.size() * (double)tcap * 8.0 / 1e9);
        if (!rows.empty()) {
            if ((rc = dev_alloc(h, &h->tab_rows, rows.size()))) return rc;
            HS_HIP(h, hipMemcpy(h->tab_rows, rows.data(), rows.size() * sizeof(TickRow), hipMemcpyHostToDevice));
            if ((rc = dev_alloc(h, &h->tab_times, rows.size() * (size_t)tcap))) return rc;
            if ((rc = dev_alloc(h, &h->tab_count, rows.size()))) return rc;
            if ((rc = dev_alloc(h, &h->tab_status, 2))) return rc;
            HS_HIP(h, hipMemset(h->tab_status, 0, 2 * sizeof(unsigned long long)));
        }
        TickTables tt{};
        tt.times = h->tab_times; tt.cap = tcap; tt.t_start = h->cfg.start_ns;
        if (!tandem.empty()) {                                      // tandem queues: hs_tables.hpp TickTables::tandem
            if ((double)n * (double)cap * 32.0 > 100e9)
                return fail(h, HS_E_INVALID, "the forward logs of the tandem queues would need %.1f GB", (double)n * (double)cap * 32.0 / 1e9);
            if ((rc = upload<int32_t>(h, &tt.tandem, tandem.data(), tandem.size(), -1))) return rc;
            if ((rc = dev_alloc(h, &tt.inj_i, (size_t)kMaxUp * (size_t)n))) return rc;
            HS_HIP(h, hipMemset(tt.inj_i, 0, (size_t)kMaxUp * (size_t)n * sizeof(int64_t)));
            for (int64_t **col : {&tt.fw_rc, &tt.fw_rrc, &tt.fw_rdr, &tt.fw_dep}) if ((rc = dev_alloc(h, col, (size_t)n * (size_t)cap))) return rc;
            for (int64_t **col : {&tt.q_rrc, &tt.q_rdr, &tt.q_pay}) if ((rc = dev_alloc(h, col, (size_t)n * (size_t)kQCap))) return rc;
            if ((rc = dev_alloc(h, &tt.cand_key, (size_t)n * 4))) return rc;
            HS_HIP(h, hipMemset(tt.cand_key, 0, (size_t)n * 4 * sizeof(int64_t)));
        }
        if (tandem.empty() && !h->any_xsrc && sched_single) {
            // injected Requests only: the same tie check, every departure / injected Request a stand-in (TickTables::standin_sched)
            if ((rc = dev_alloc(h, &tt.cand_key, (size_t)n * 4))) return rc;
            HS_HIP(h, hipMemset(tt.cand_key, 0, (size_t)n * 4 * sizeof(int64_t)));
        }
        tt.standin_sched = sched_single ? 1 : 0;
        if (tandem.empty() && h->any_xsrc && h->cfg.mode == HS_MODE_SINGLE) {
            // several Sources per Server: a pending DEPARTURE's last election key is the construction rank of the Source its lineage
            // goes back to, which the engine does not carry -- cand_rank() uses the LP's first-listed Source.  When the election of
            // the event beyond end_ns comes down to that key for such a candidate (every other key ties with another LP's), the run
            // is repeated on the single heap (Totals::undecided bit 1, tandem_fallback) instead of guessing.  Found by
            // tools/gpu_random_sweep.py, multi_source case 22522: two lock-step constant Sources in different LPs, the other LP's
            // first-listed Source a Poisson one constructed earlier.
            if ((rc = dev_alloc(h, &tt.cand_key, (size_t)n * 4))) return rc;
            HS_HIP(h, hipMemset(tt.cand_key, 0, (size_t)n * 4 * sizeof(int64_t)));
            // ... and the engine carries the lineage's Source with every pending departure (TickTables::rs_dep): the stand-in is then
            // only used for departures whose lineage starts at an injected Request
            if ((rc = dev_alloc(h, &tt.rs_dep, (size_t)n * (size_t)h->C))) return rc;
            if ((rc = dev_alloc(h, &tt.rs_q, (size_t)n * (size_t)kQCap))) return rc;
            HS_HIP(h, hipMemset(tt.rs_dep, 0xff, (size_t)n * (size_t)h->C));
            HS_HIP(h, hipMemset(tt.rs_q, 0xff, (size_t)n * (size_t)kQCap));
        }
        if ((rc = upload<int32_t>(h, &tt.src_row, srow.data(), srow.size(), -1))) return rc;
        if ((rc = upload<int32_t>(h, &tt.probe_row, prow.data(), prow.size(), -1))) return rc;
        if ((rc = upload<TickTables>(h, &h->P.tabs, &tt, 1, TickTables{}))) return rc;
    }
    h->P.xsrc_kind = nullptr; h->P.xsrc_rate = nullptr; h->P.xsrc_stop = nullptr;
    if (h->any_xsrc) {
        if ((rc = upload<uint8_t>(h, &h->P.xsrc_kind, xk.data(), xk.size(), 0))) return rc;
        if ((rc = upload<double>(h, &h->P.xsrc_rate, xr.data(), xr.size(), 1.0))) return rc;
        if ((rc = upload<int64_t>(h, &h->P.xsrc_stop, xstop.data(), xstop.size(), (int64_t)-1))) return rc;
    }
    h->P.sched_off = nullptr; h->P.sched_t = nullptr;
    if (n_sched > 0) {
        if ((rc = upload<int64_t>(h, &h->P.sched_off, st->sched_off, (size_t)n + 1, 0))) return rc;
        if ((rc = upload<int64_t>(h, &h->P.sched_t, st->sched_time_ns, (size_t)n_sched, 0))) return rc;
    }
    h->P.tie_rank = nullptr;
    // the Sources in `sources=[...]` order: (LP, slot) pairs; default = LP-major, slot-minor
    std::vector<int32_t> so;
    std::vector<uint8_t> sslot;
    {
        auto has_src = [&](int lp, int slot) {
            return slot == 0 ? (st->src_kind ? st->src_kind[lp] : HS_SRC_POISSON) != HS_SRC_NONE : xk[(size_t)(slot - 1) * n + lp] != 0;
        };
        int64_t n_src_total = n_xsrc_total;
        for (int i = 0; i < n; ++i) if (has_src(i, 0)) ++n_src_total;
        if (st->source_order) {
            std::vector<uint8_t> taken((size_t)n * (kMaxXSrc + 1), (uint8_t)0);
            for (int64_t k = 0; k < n_src_total; ++k) {
                const int lp = st->source_order[k], slot = st->source_slot_order ? st->source_slot_order[k] : 0;
                if (lp < 0 || lp >= n || slot < 0 || slot > kMaxXSrc || !has_src(lp, slot) || taken[(size_t)slot * n + lp])
                    return fail(h, HS_E_INVALID, "source_order / source_slot_order must list every Source exactly once");
                taken[(size_t)slot * n + lp] = 1;
                so.push_back(lp); sslot.push_back((uint8_t)slot);
            }
        } else {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= kMaxXSrc; ++j) if (has_src(i, j)) { so.push_back(i); sslot.push_back((uint8_t)j); }
        }
    }
    h->h_src_lp = so; h->h_src_slot = sslot;      // (setup_exact_plain: the single-heap machinery built on demand)
    // the Probes in `probes=[...]` order: (LP, slot) pairs; default = LP-major, slot-minor
    std::vector<int32_t> po;
    std::vector<uint8_t> pslot;
    {
        std::vector<uint8_t> taken((size_t)n * kMaxProbes, (uint8_t)0);
        if (st->probe_order) {
            for (int64_t k = 0; k < n_prb_total; ++k) {
                const int lp = st->probe_order[k], slot = st->probe_slot_order ? st->probe_slot_order[k] : 0;
                if (lp < 0 || lp >= n || slot < 0 || slot >= kMaxProbes || pm[(size_t)slot * n + lp] == 255 || taken[(size_t)slot * n + lp])
                    return fail(h, HS_E_INVALID, "probe_order / probe_slot_order must list every probe exactly once");
                taken[(size_t)slot * n + lp] = 1;
                po.push_back(lp); pslot.push_back((uint8_t)slot);
            }
        } else {
            for (int i = 0; i < n && n_prb_total > 0; ++i)
                for (int j = 0; j < kMaxProbes; ++j) if (pm[(size_t)j * n + i] != 255) { po.push_back(i); pslot.push_back((uint8_t)j); }
        }
    }
    if (((double)so.size() + (double)n + (double)po.size() + 1.0) * 8.0 >= 2147483647.0)
        return fail(h, HS_E_UNSUPPORTED, "too many Sources / stations / Probes for the 32-bit construction ranks of the election key");
    if (st->source_order || st->probe_order || !tandem.empty()) {   // cross-LP ties go to the entity the reference constructed first (cand_rank, hs_station.hpp)
        std::vector<int32_t> tr((size_t)n * (kMaxXSrc + 2) + 1 + (size_t)n * kMaxProbes, -1);
        int32_t *sr = tr.data() + n;
        for (size_t q = 0; q < so.size(); ++q) {
            sr[(size_t)sslot[q] * n + (size_t)so[q]] = (int32_t)q;               // a tick: its own Source's position
            if (tr[(size_t)so[q]] < 0) tr[(size_t)so[q]] = (int32_t)q;           // anything else: the LP's first-listed Source
        }
        if (!tandem.empty())                                                     // a Server behind a Server: what reaches it descends
            for (int i = 0; i < n; ++i) {                                        // from the Sources of its chain's head
                int head = i;                                                    // (its first-listed upstream, all the way up)
                while (tandem[(size_t)head] >= 0) head = tandem[(size_t)head];
                if (head != i && tr[(size_t)i] < 0 && tr[(size_t)head] >= 0) tr[(size_t)i] = tr[(size_t)head];
            }
        for (int i = 0; i < n; ++i) if (tr[(size_t)i] < 0) tr[(size_t)i] = (int32_t)so.size() + i;   // sourceless LPs after them
        tr[(size_t)n * (kMaxXSrc + 2)] = (int32_t)so.size() + n;                 // Probes behind all of them ...
        int32_t *pr = tr.data() + (size_t)n * (kMaxXSrc + 2) + 1;                // ... each by its own position in `probes=[...]`
        for (size_t q = 0; q < po.size(); ++q) pr[(size_t)pslot[q] * n + (size_t)po[q]] = (int32_t)so.size() + n + (int32_t)q;
        if (!tandem.empty())     // ... and where every other key ties between two Servers of a chain, the upstream one's event was created first
            for (size_t q = 0; q < tr.size(); ++q) {                             // (ranks only compare: room for the pass below them)
                const size_t lp = q < (size_t)n * (kMaxXSrc + 2) ? q % (size_t)n : (q - (size_t)n * (kMaxXSrc + 2) - 1) % (size_t)n;
                // (3 bits for the pass: clamped -- a chain deeper than 7 behind a fan-in beyond kMaxUp runs on the single heap, which
                //  does not read this key; unclamped, a pass >= 8 spilled into the construction rank above it: ADVICE r3)
                if (q == (size_t)n * (kMaxXSrc + 2)) { tr[q] = tr[q] * 8; continue; }
                const int32_t ps = tandem[(size_t)kMaxUp * n + lp];
                tr[q] = tr[q] * 8 + (ps > 7 ? 7 : ps < 0 ? 0 : ps);
            }
        if ((rc = upload<int32_t>(h, &h->P.tie_rank, tr.data(), tr.size(), 0))) return rc;
    }
    h->P.sched_idx = nullptr;
    if (h->cfg.mode == HS_MODE_REPLICAS && (n_sched > 0 || h->any_probe || h->any_xsrc)) {
        {   // every LP's Sources in its own construction order: slots, 255-terminated
            std::vector<uint8_t> lso((size_t)n * (kMaxXSrc + 1), (uint8_t)255);
            std::vector<int> cnt((size_t)n, 0);
            for (size_t q = 0; q < so.size(); ++q) lso[(size_t)cnt[(size_t)so[q]]++ * n + (size_t)so[q]] = sslot[q];
            if ((rc = upload<uint8_t>(h, &h->XI.lp_src_slots, lso.data(), lso.size(), 255))) return rc;
        }
        // One prologue per LP (every LP is its own Simulation): its Events sorted by construction rank inside the LP
        std::vector<int64_t> se((size_t)n_sched), sr((size_t)n_sched);
        int64_t span = 0;
        for (int i = 0; i < n && n_sched > 0; ++i) {
            const int64_t a = st->sched_off[i], b = st->sched_off[i + 1];
            for (int64_t k = a; k < b; ++k) se[(size_t)k] = k;
            if (st->sched_rank) {
                std::sort(se.begin() + a, se.begin() + b, [&](int64_t x, int64_t y) { return st->sched_rank[x] < st->sched_rank[y]; });
                for (int64_t k = a; k < b; ++k) {
                    const int64_t r = st->sched_rank[se[(size_t)k]];
                    if (r < 0 || (k > a && r == st->sched_rank[se[(size_t)k - 1]]))
                        return fail(h, HS_E_INVALID, "LP %d: sched_rank must hold distinct positions >= 0", i);
                }
            }
            for (int64_t k = a; k < b; ++k) {
                sr[(size_t)k] = st->sched_rank ? st->sched_rank[se[(size_t)k]] : k - a;
                if (sr[(size_t)k] + 1 > span) span = sr[(size_t)k] + 1;
            }
        }
        if (span > max_sched + (1 << 16)) return fail(h, HS_E_INVALID, "sched_rank positions are implausibly sparse");
        if ((rc = upload<int64_t>(h, &h->XI.sched_rank, sr.data(), sr.size(), 0))) return rc;
        if ((rc = upload<int64_t>(h, &h->XI.sched_entry, se.data(), se.size(), 0))) return rc;
        h->XI.n_src = 0; h->XI.n_probe = 0; h->XI.n_sched = n_sched; h->XI.per_lp = 1;
        if ((rc = dev_alloc(h, &h->XI.sched_idx, (size_t)n_sched))) return rc;
        HS_HIP(h, hipMemset(h->XI.sched_idx, 0, (size_t)(n_sched > 0 ? n_sched : 1) * sizeof(uint32_t)));
        h->P.sched_idx = h->XI.sched_idx;
        const int64_t n_init_lp = 1 + kMaxXSrc + kMaxProbes + span;
        h->XI.init_cap_lp = n_init_lp;
        h->XI.heap_cap_lp = n_init_lp + h->C + 32;
        h->XI.pool_cap_lp = 2 * n_init_lp + 64;
        h->xs_host = XState{};
        if ((rc = dev_alloc(h, &h->xs_host.heap, (size_t)n * (size_t)h->XI.heap_cap_lp))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.qhead, (size_t)n))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.qtail, (size_t)n))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.pnext, (size_t)n * (size_t)h->XI.pool_cap_lp))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.pidx, (size_t)n * (size_t)h->XI.pool_cap_lp))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.init_t, (size_t)n * (size_t)n_init_lp))) return rc;
        if ((rc = dev_alloc(h, &h->xs, (size_t)n + 1))) return rc;
        h->exact = true;
        h->exact_prologue = true;
    }
    if (h->cfg.mode == HS_MODE_SINGLE && (n_sched > 0 || h->any_probe || h->any_xsrc || !tandem.empty())) {
        // The prologue (hs_exact.hpp): the reference's pre-run events in the order it constructs them.  (Tandem queues: the same
        // loop as the engine's exact path for whole runs -- `exact_only`.)
        h->exact_prologue = n_sched > 0 || h->any_probe || h->any_xsrc;
        // (more upstream Servers than the passes merge: one heap from the start.  Tandem queues next to pre-run events start on the
        // passes like any lazy_prologue engine and move to the single heap -- which IS the prologue's loop -- on the first hazard)
        h->exact_only = !tandem.empty() && h->tandem_fan_in;
        std::vector<int32_t> sl((size_t)n_sched);
        std::vector<int64_t> se((size_t)n_sched);
        std::vector<int32_t> lp_of((size_t)n_sched);
        for (int i = 0; i < n && n_sched > 0; ++i)
            for (int64_t k = st->sched_off[i]; k < st->sched_off[i + 1]; ++k) lp_of[(size_t)k] = i;
        std::vector<int64_t> sr((size_t)n_sched);
        for (int64_t j = 0; j < n_sched; ++j) se[(size_t)j] = j;
        if (st->sched_rank) {
            std::sort(se.begin(), se.end(), [&](int64_t a, int64_t b) { return st->sched_rank[a] < st->sched_rank[b]; });
            for (int64_t j = 0; j < n_sched; ++j) {
                const int64_t r = st->sched_rank[se[(size_t)j]];
                if (r < 0 || (j > 0 && r == st->sched_rank[se[(size_t)j - 1]]))
                    return fail(h, HS_E_INVALID, "sched_rank must hold distinct positions >= 0");
            }
        }
        for (int64_t j = 0; j < n_sched; ++j) {
            sl[(size_t)j] = lp_of[(size_t)se[(size_t)j]];
            sr[(size_t)j] = st->sched_rank ? st->sched_rank[se[(size_t)j]] : j;
        }
        if ((rc = upload<int64_t>(h, &h->XI.sched_rank, sr.data(), sr.size(), 0))) return rc;
        if ((rc = upload<int32_t>(h, &h->XI.src_lp, so.data(), so.size(), 0))) return rc;
        if ((rc = upload<uint8_t>(h, &h->XI.src_slot, sslot.data(), sslot.size(), 0))) return rc;
        if ((rc = upload<int32_t>(h, &h->XI.probe_lp, po.data(), po.size(), 0))) return rc;
        if ((rc = upload<uint8_t>(h, &h->XI.probe_slot, pslot.data(), pslot.size(), 0))) return rc;
        if ((rc = upload<int32_t>(h, &h->XI.sched_lp, sl.data(), sl.size(), 0))) return rc;
        if ((rc = upload<int64_t>(h, &h->XI.sched_entry, se.data(), se.size(), 0))) return rc;
        h->XI.n_src = (int32_t)so.size(); h->XI.n_probe = (int32_t)po.size(); h->XI.n_sched = n_sched;
        if ((rc = dev_alloc(h, &h->XI.sched_idx, (size_t)n_sched))) return rc;
        HS_HIP(h, hipMemset(h->XI.sched_idx, 0, (size_t)(n_sched > 0 ? n_sched : 1) * sizeof(uint32_t)));
        h->P.sched_idx = h->XI.sched_idx;
        int64_t rank_span = 0;                                  // positions 0 .. rank_span-1 (cancelled Events leave gaps)
        for (int64_t j = 0; j < n_sched; ++j) if (sr[(size_t)j] + 1 > rank_span) rank_span = sr[(size_t)j] + 1;
        if (rank_span > n_sched + (1 << 20)) return fail(h, HS_E_INVALID, "sched_rank positions are implausibly sparse");
        const int64_t n_init = (int64_t)so.size() + (int64_t)po.size() + rank_span;
        h->n_init = n_init;
        h->lazy_prologue = h->exact_prologue;
        h->xs_host = XState{};
        h->xs_host.heap_cap = n_init + (int64_t)n * (h->C + 16) + 1024;
        h->xs_host.pool_cap = 2 * n_init + 16 * (int64_t)n + 1024;
        if (!tandem.empty() || h->any_xsrc || h->any_sched)       // a whole run (tandem queues; several Sources per Server after an undecided election): one list cell per admitted Request
            h->xs_host.pool_cap = std::min<int64_t>(h->xs_host.pool_cap + (int64_t)n * cap, (int64_t)1 << 30);
        if ((rc = dev_alloc(h, &h->xs_host.heap, (size_t)h->xs_host.heap_cap))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.qhead, (size_t)n))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.qtail, (size_t)n))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.pnext, (size_t)h->xs_host.pool_cap))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.pidx, (size_t)h->xs_host.pool_cap))) return rc;
        if ((rc = dev_alloc(h, &h->xs_host.init_t, (size_t)n_init))) return rc;
        if ((rc = dev_alloc(h, &h->xs, 1))) return rc;
        h->exact = true;
    }
    const size_t N = (size_t)n, NC = (size_t)n * (size_t)h->C;
#define AL(field, count) if ((rc = dev_alloc(h, &h->X.field, count))) return rc
    AL(A, N); AL(seqA, N); AL(crtA, N); AL(arr_k, N); AL(arr_time, N); AL(svc_k, N);
    AL(D, NC); AL(seqD, NC); AL(crtD, NC); AL(svc_s, NC); AL(crt, NC);
    AL(seq, N); AL(buf, N); AL(active, N);
    AL(generated, N); AL(accepted, N); AL(dropped, N); AL(completed, N); AL(rejected, N); AL(started, N);
    AL(received, N); AL(sink_w, N); AL(total_service, N); AL(q, N); AL(grp_time, N); AL(last_time, N);
    AL(events, N); AL(ev_kind, N * 11);
    AL(dpA, N); AL(rcA, N); AL(dpD, NC); AL(rcD, NC); AL(wkD, NC); AL(qdep, N * kQCap); AL(qrc, N * kQCap);   // lineage (hs_station.hpp)
    if (h->any_profile) {      // the general-path instantiation of the run kernel loads / stores the probe state of every LP
        AL(PA, N * kMaxProbes); AL(seqP, N * kMaxProbes); AL(crtP, N * kMaxProbes); AL(p_arr, N * kMaxProbes);
        AL(p_n, N * kMaxProbes); AL(ev_probe, N * 2); AL(sched_i, N); AL(rcP, N * kMaxProbes);
    }
    if (h->any_xsrc) {
        AL(XA, N * kMaxXSrc); AL(crtX, N * kMaxXSrc); AL(x_arr, N * kMaxXSrc); AL(x_n, N * kMaxXSrc); AL(seqX, N * kMaxXSrc);
        AL(x_k, N * kMaxXSrc); AL(dpX, N * kMaxXSrc); AL(rcX, N * kMaxXSrc);
    }
    if (h->any_probe) {
        h->L.pcap = (int64_t)(horizon_s / min_interval) + 8;
        if ((double)h->L.pcap * (double)n * 16.0 > 50e9) return fail(h, HS_E_INVALID, "probe logs would need %.1f GB", (double)h->L.pcap * n * 16.0 / 1e9);
        if ((rc = dev_alloc(h, &h->L.probe_t, N * (size_t)h->L.pcap * (size_t)h->n_probe_slots))) return rc;
        if ((rc = dev_alloc(h, &h->L.probe_v, N * (size_t)h->L.pcap * (size_t)h->n_probe_slots))) return rc;
    }
#undef AL
    if ((rc = dev_alloc(h, &h->L.adm, N * (size_t)cap))) return rc;
    if ((rc = dev_alloc(h, &h->L.sink_t, N * (size_t)cap))) return rc;
    if (h->C > 1) {
        // explicit created_at column: completions leave in a different order than admissions
        if ((rc = dev_alloc(h, &h->L.sink_created_own, N * (size_t)cap))) return rc;
    }
    h->L.sink_created = (h->C > 1) ? h->L.sink_created_own : h->L.adm;
    if ((rc = dev_alloc(h, &h->tot, 1))) return rc;
    HS_HIP(h, hipMemset(h->tot, 0, sizeof(Totals)));
    if ((rc = dev_alloc(h, &h->cands, std::max<size_t>((size_t)h->n_blocks, (size_t)(n + 1) / 2)))) return rc;   // (the wide kernel: one per >= 2 LPs)
    if (h->C == 1 && h->uni_grid && !h->any_profile && h->cfg.mode == HS_MODE_SINGLE) {
        if ((rc = dev_alloc(h, &h->wide_ctl, 1))) return rc;
        if ((rc = dev_alloc(h, &h->wide_bail, (size_t)n))) return rc;
        if ((rc = dev_alloc(h, &h->wave_parts, ((size_t)n + 7) / 8))) return rc;
        HS_HIP(h, hipMemset(h->wide_ctl, 0, sizeof(WideCtl)));
    }
    HS_HIP(h, hipMemset(h->tot, 0, sizeof(Totals)));
    // (the uploads and fills above went through the NULL stream; the engine's launches use a non-blocking stream, which does not
    //  wait for it: everything is in place before the caller can enqueue a run)
    HS_HIP(h, hipDeviceSynchronize());
    h->have_stations = true;
    return HS_OK;
}

int hs_engine_set_network(hs_engine *h, const hs_network *net) {
    if (!h || !net) return fail(h, HS_E_INVALID, "hs_engine_set_network: null argument");
    if (!h->have_stations) return fail(h, HS_E_STATE, "set stations before the network");
    if (h->is_net) return fail(h, HS_E_STATE, "network already set");
    if (h->initialised) return fail(h, HS_E_STATE, "set the network before the first run");
    if (h->cfg.mode != HS_MODE_SINGLE) return fail(h, HS_E_INVALID, "a network of stations is one Simulation: HS_MODE_SINGLE");
    if (h->C > 4) return fail(h, HS_E_UNSUPPORTED, "networked stations support concurrency <= 4 for now");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    { const int rcr = ensure_reset(h); if (rcr) return rcr; }     // (a bootstrap deferred for a station engine this no longer is)
    const int n = h->cfg.n_lp, nl = net->n_links;
    if (nl < 0) return fail(h, HS_E_INVALID, "n_links < 0");
    // A shard of a larger network: link endpoints are network-wide station indices, this engine owns
    // [lp_base, lp_base + n_lp).  Otherwise the engine holds the whole network and endpoints are its own indices.
    const bool global = net->n_global_lp > 0;
    const int64_t lo = global ? (int64_t)h->cfg.lp_base : 0;
    const int64_t n_all = global ? net->n_global_lp : n;
    if (global && (lo + n > n_all)) return fail(h, HS_E_INVALID, "shard [%lld, %lld) exceeds the %lld stations of the network",
                                                (long long)lo, (long long)(lo + n), (long long)n_all);
    if (global && !net->link_gid) return fail(h, HS_E_INVALID, "a shard needs link_gid (network-wide link ids)");
    if (!net->egress_kind) return fail(h, HS_E_INVALID, "egress_kind is required");
    if (nl > 0 && (!net->link_dst || !net->link_lat_min_s || !net->link_src))
        return fail(h, HS_E_INVALID, "link_dst, link_src and link_lat_min_s are required");
    // lookahead W = min over links of int(to_seconds(from_seconds(lat_min)) * 1e9) -- the same truncations the
    // device applies (core/temporal.py:62,66)
    int64_t W = INT64_MAX;
    for (int l = 0; l < nl; ++l) {
        const double lm = net->link_lat_min_s[l];
        if (!(lm > 0.0) || !std::isfinite(lm))
            return fail(h, HS_E_INVALID, "link %d: min latency must be > 0 (conservative windows need lookahead), got %g", l, lm);
        const double lc = (double)(int64_t)(lm * 1e9) / 1e9;
        const int64_t w = (int64_t)(lc * 1e9);
        if (w <= 0) return fail(h, HS_E_INVALID, "link %d: min latency %g s truncates to 0 ns", l, lm);
        if (w < W) W = w;
        if (net->link_dst[l] < 0 || net->link_dst[l] >= n_all || net->link_src[l] < 0 || net->link_src[l] >= n_all)
            return fail(h, HS_E_INVALID, "link %d: endpoint out of range", l);
        const bool src_here = net->link_src[l] >= lo && net->link_src[l] < lo + n;
        const bool dst_here = net->link_dst[l] >= lo && net->link_dst[l] < lo + n;
        if (!src_here && !dst_here) return fail(h, HS_E_INVALID, "link %d touches no station of this shard", l);
        const int jk = net->link_jitter_kind ? net->link_jitter_kind[l] : HS_LAT_CONSTANT;
        if (jk == HS_LAT_EXPONENTIAL && !(net->link_jitter_mean_s && net->link_jitter_mean_s[l] > 0.0))
            return fail(h, HS_E_INVALID, "link %d: exponential jitter needs mean > 0", l);
        if (jk != HS_LAT_EXPONENTIAL && jk != HS_LAT_CONSTANT)
            return fail(h, HS_E_UNSUPPORTED, "link %d: jitter kind %d is not lowered", l, jk);
        if (jk == HS_LAT_CONSTANT && net->link_jitter_mean_s && !(net->link_jitter_mean_s[l] >= 0.0 && net->link_jitter_mean_s[l] < 1e6))
            return fail(h, HS_E_INVALID, "link %d: constant jitter %g s", l, net->link_jitter_mean_s[l]);
    }
    std::vector<int32_t> rt0((size_t)n, -1), rt1((size_t)n, -1), rt2((size_t)n, -1), rt3((size_t)n, -1), lof((size_t)n, -1);
    std::vector<uint8_t> rtk((size_t)n, (uint8_t)2);
    std::vector<uint8_t> link_used((size_t)(nl > 0 ? nl : 1), 0);
    auto use_link = [&](int lp, int l) -> int {
        if (l < 0 || l >= nl) return fail(h, HS_E_INVALID, "LP %d: link index %d out of range", lp, l);
        if (net->link_src[l] != lo + lp) return fail(h, HS_E_INVALID, "LP %d uses link %d whose source is station %d", lp, l, net->link_src[l]);
        if (link_used[(size_t)l]) return fail(h, HS_E_INVALID, "link %d is referenced twice", l);
        link_used[(size_t)l] = 1;
        return HS_OK;
    };
    for (int i = 0; i < n; ++i) {
        const int ek = net->egress_kind[i];
        int rc2;
        if (ek == HS_EGRESS_ROUTER) {
            if (!net->router_target0 || !net->router_target1) return fail(h, HS_E_INVALID, "router targets are required");
            const int k = net->router_n_targets ? net->router_n_targets[i] : 2;
            if (k < 1 || k > 4) return fail(h, HS_E_UNSUPPORTED, "LP %d: RandomRouter with %d targets (1..4 are lowered)", i, k);
            if ((k > 2 && !net->router_target2) || (k > 3 && !net->router_target3))
                return fail(h, HS_E_INVALID, "router_target2 / router_target3 are required for routers with that many targets");
            rtk[(size_t)i] = (uint8_t)k;
            const int32_t tg[4] = {net->router_target0[i], net->router_target1[i], k > 2 ? net->router_target2[i] : -1,
                                   k > 3 ? net->router_target3[i] : -1};
            int n_link = 0;
            for (int q = 0; q < k; ++q) {
                if (tg[q] < -1) return fail(h, HS_E_INVALID, "LP %d: bad router target", i);
                if (tg[q] >= 0) { if ((rc2 = use_link(i, tg[q]))) return rc2; ++n_link; }
            }
            if (n_link > 2) return fail(h, HS_E_UNSUPPORTED, "LP %d: a router with more than two NetworkLink targets is not lowered", i);
            rt0[(size_t)i] = tg[0]; rt1[(size_t)i] = tg[1]; rt2[(size_t)i] = tg[2]; rt3[(size_t)i] = tg[3];
        } else if (ek == HS_EGRESS_LINK) {
            if (!net->link_of) return fail(h, HS_E_INVALID, "link_of is required");
            lof[(size_t)i] = net->link_of[i];
            if ((rc2 = use_link(i, lof[(size_t)i]))) return rc2;
        } else if (ek != HS_EGRESS_NONE && ek != HS_EGRESS_SINK) {
            return fail(h, HS_E_UNSUPPORTED, "LP %d: egress kind %d is not lowered", i, ek);
        }
    }
    if (nl == 0) W = h->cfg.horizon_ns - h->cfg.start_ns + 1;   // no links: one window
    h->window_ns = W;
    int rc;
    std::vector<uint64_t> rbase((size_t)n), lbase((size_t)(nl > 0 ? nl : 1));
    std::vector<uint64_t> sbase((size_t)n);
    HS_HIP(h, hipMemcpy(sbase.data(), h->P.stream_base, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) rbase[(size_t)i] = net->router_stream_base ? net->router_stream_base[i] : sbase[(size_t)i];
    for (int l = 0; l < nl; ++l) {
        const int64_t sl = net->link_src[l] - lo;   // incoming links of a shard (remote source) are never drawn from here
        lbase[(size_t)l] = net->link_stream_base ? net->link_stream_base[l]
                                                 : (sl >= 0 && sl < n ? sbase[(size_t)sl] : 0);
    }
    const size_t NL = (size_t)(nl > 0 ? nl : 1);
    std::vector<uint8_t> jk(NL, (uint8_t)HS_LAT_CONSTANT);
    std::vector<double> jm(NL, 0.0), lmin(NL, 1.0), lloss(NL, 0.0);
    std::vector<int32_t> ldst(NL, 0);
    for (int l = 0; l < nl; ++l) {
        jk[(size_t)l] = net->link_jitter_kind ? net->link_jitter_kind[l] : (uint8_t)HS_LAT_CONSTANT;
        jm[(size_t)l] = net->link_jitter_mean_s ? net->link_jitter_mean_s[l] : 0.0;
        lmin[(size_t)l] = net->link_lat_min_s[l];
        ldst[(size_t)l] = net->link_dst[l];
        if (net->link_loss_rate) {
            const double pl = net->link_loss_rate[l];
            if (!(pl >= 0.0 && pl <= 1.0))                       // components/network/link.py:71-72
                return fail(h, HS_E_INVALID, "link %d: packet_loss_rate must be in [0, 1], got %g", l, pl);
            lloss[(size_t)l] = pl;
        }
    }
    if ((rc = upload<uint8_t>(h, &h->NP.egress, net->egress_kind, (size_t)n, 0))) return rc;
    if ((rc = upload<int32_t>(h, &h->NP.rt0, rt0.data(), (size_t)n, -1))) return rc;
    if ((rc = upload<int32_t>(h, &h->NP.rt1, rt1.data(), (size_t)n, -1))) return rc;
    if ((rc = upload<int32_t>(h, &h->NP.rt2, rt2.data(), (size_t)n, -1))) return rc;
    if ((rc = upload<int32_t>(h, &h->NP.rt3, rt3.data(), (size_t)n, -1))) return rc;
    if ((rc = upload<uint8_t>(h, &h->NP.rt_cnt, rtk.data(), (size_t)n, 2))) return rc;
    if ((rc = upload<int32_t>(h, &h->NP.link_of, lof.data(), (size_t)n, -1))) return rc;
    if ((rc = upload<uint64_t>(h, &h->NP.route_base, rbase.data(), (size_t)n, 0))) return rc;
    h->NP.n_links = nl;
    if ((rc = upload<int32_t>(h, &h->NP.link_dst, ldst.data(), NL, 0))) return rc;
    if ((rc = upload<double>(h, &h->NP.link_lat_min, lmin.data(), NL, 0.0))) return rc;
    if ((rc = upload<uint8_t>(h, &h->NP.link_jit_kind, jk.data(), NL, 1))) return rc;
    if ((rc = upload<double>(h, &h->NP.link_jit_mean, jm.data(), NL, 0.0))) return rc;
    if ((rc = upload<uint64_t>(h, &h->NP.link_base, lbase.data(), NL, 0))) return rc;
    {   // table-decided losses (hs_network.link_drop_capacity): the bit tables, all clear, and the send log
        std::vector<int64_t> doff(NL + 1, 0);
        int64_t total = 0;
        for (int l = 0; l < nl; ++l) {
            const int64_t c = net->link_drop_capacity ? net->link_drop_capacity[l] : 0;
            if (c < 0) return fail(h, HS_E_INVALID, "link %d: link_drop_capacity < 0", l);
            if (c > 0 && lloss[(size_t)l] != 0.0)
                return fail(h, HS_E_INVALID, "link %d: a link with a loss table must have link_loss_rate 0", l);
            doff[(size_t)l] = total;
            if (c > 0) { lloss[(size_t)l] = kLossTable; total += (c + 31) / 32 * 32; }
        }
        for (size_t l = (size_t)(nl > 0 ? nl : 0); l <= NL; ++l) doff[l] = total;
        h->drop_off_host = doff;
#ifdef HS_NO_LOSS_TABLES
        if (total > 0) return fail(h, HS_E_UNSUPPORTED, "built without loss tables");
#else
        h->NP.loss_tables = nullptr;
        if (total > 0) {
            LossTables lt{};
            uint32_t *bits = nullptr;
            if ((rc = upload<int64_t>(h, &lt.drop_off, doff.data(), NL + 1, 0))) return rc;
            if ((rc = dev_alloc(h, &bits, (size_t)(total / 32)))) return rc;
            HS_HIP(h, hipMemset(bits, 0, (size_t)(total / 32) * sizeof(uint32_t)));
            lt.drop_bits = bits;
            h->drop_bits_dev = bits;
            if ((rc = dev_alloc(h, &lt.send_log, (size_t)total * 3))) return rc;
            if ((rc = dev_alloc(h, &lt.send_log_n, 1))) return rc;
            HS_HIP(h, hipMemset(lt.send_log_n, 0, sizeof(unsigned long long)));
            lt.send_log_cap = total;
            lt.overflow_word = &h->tot->overflow;
            h->loss_host = lt;
            if ((rc = upload<LossTables>(h, &h->NP.loss_tables, &lt, 1, lt))) return rc;
        }
#endif
    }
    if ((rc = upload<double>(h, &h->NP.link_loss, lloss.data(), NL, 0.0))) return rc;
    std::vector<int32_t> in_deg_h;
    {   // incoming links per LP (CSR) and the transit floor of every link, for the asynchronous engine
        std::vector<int32_t> in_off((size_t)n + 1, 0), in_links(NL, 0);
        std::vector<int64_t> lat_ns(NL, 1);
        {   // (a shard: only the links that END here, indexed by the local station)
            auto here = [&](int l) { return net->link_dst[l] >= lo && net->link_dst[l] < lo + n; };
            for (int l = 0; l < nl; ++l) if (here(l)) in_off[(size_t)(net->link_dst[l] - lo) + 1]++;
            for (int i = 0; i < n; ++i) in_off[(size_t)i + 1] += in_off[(size_t)i];
            std::vector<int32_t> cur(in_off.begin(), in_off.end() - 1);
            for (int l = 0; l < nl; ++l) if (here(l)) in_links[(size_t)cur[(size_t)(net->link_dst[l] - lo)]++] = l;
        }
        for (int l = 0; l < nl; ++l) {
            const double lc = (double)(int64_t)(net->link_lat_min_s[l] * 1e9) / 1e9;   // ConstantLatency.get_latency().to_seconds()
            lat_ns[(size_t)l] = (int64_t)(lc * 1e9);
        }
        in_deg_h.resize((size_t)n);
        for (int i = 0; i < n; ++i) in_deg_h[(size_t)i] = in_off[(size_t)i + 1] - in_off[(size_t)i];
        if ((rc = upload<int32_t>(h, &h->NP.in_off, in_off.data(), (size_t)n + 1, 0))) return rc;
        if ((rc = upload<int32_t>(h, &h->NP.in_links, in_links.data(), NL, 0))) return rc;
        if ((rc = upload<int64_t>(h, &h->NP.link_lat_ns, lat_ns.data(), NL, 1))) return rc;
    }
    h->NP.link_gid = nullptr;
    if (net->link_gid && nl > 0) {
        std::vector<int32_t> gid((size_t)nl);
        int64_t gmax = -1;
        for (int l = 0; l < nl; ++l) {
            if (net->link_gid[l] < 0 || net->link_gid[l] > 0x7fffffffll) return fail(h, HS_E_INVALID, "link %d: bad link_gid", l);
            gid[(size_t)l] = (int32_t)net->link_gid[l];
            if (net->link_gid[l] > gmax) gmax = net->link_gid[l];
        }
        h->n_gid = net->n_global_links > gmax + 1 ? net->n_global_links : gmax + 1;
        std::vector<int32_t> g2l((size_t)h->n_gid, -1);
        for (int l = 0; l < nl; ++l) {
            if (g2l[(size_t)gid[(size_t)l]] >= 0) return fail(h, HS_E_INVALID, "link_gid %d appears twice", gid[(size_t)l]);
            g2l[(size_t)gid[(size_t)l]] = l;
        }
        if ((rc = upload<int32_t>(h, &h->NP.link_gid, gid.data(), (size_t)nl, 0))) return rc;
        const int32_t *g2l_dev = nullptr;
        if ((rc = upload<int32_t>(h, &g2l_dev, g2l.data(), (size_t)h->n_gid, -1))) return rc;
        h->h_gid2local = g2l;
        h->gid2local = const_cast<int32_t *>(g2l_dev);
    }
    h->net_global = global;
    h->n_global_lp = (int32_t)n_all;
    h->SC = ShardCtl{};
    h->SC.lp_base = lo;
    h->h_link_dst.assign(ldst.begin(), ldst.end());
    h->h_link_src.assign(net->link_src, net->link_src + nl);
    const int bag = net->bag_capacity > 0 ? net->bag_capacity : 16;
    h->NX.bag_cap = bag;
    const size_t N = (size_t)n, NB = (size_t)n * (size_t)bag;
#define ALN(field, count) if ((rc = dev_alloc(h, &h->NX.field, count))) return rc
    ALN(route_k, N); ALN(routed, N); ALN(link_k, NL); ALN(link_in, NL); ALN(link_sent, NL); ALN(link_packets, NL); ALN(next_time, N);
    ALN(pend_pay, N);
    ALN(bag_cnt, N); ALN(bag_t, NB); ALN(bag_ts, NB); ALN(bag_cr, NB); ALN(bag_link, NB); ALN(bag_lin, NB);
    ALN(in_cnt, 2 * N); ALN(in_t, 2 * NB); ALN(in_ts, 2 * NB); ALN(in_cr, 2 * NB); ALN(in_link, 2 * NB); ALN(in_lin, 2 * NB);
    if ((rc = dev_alloc(h, &h->X.enqpay, N * (size_t)kEnqPay))) return rc;    // hs_net_async's ENQ payloads (general path)
    int aqc = 1;
    while (aqc < bag) aqc <<= 1;      // a power of two: queue slots are addressed with a mask, not a 64-bit modulo
#ifndef HS_AQ_MIN
#define HS_AQ_MIN 64
#endif
    if (aqc < HS_AQ_MIN) aqc = HS_AQ_MIN;   // pre-sent messages (hs_netstation.hpp `early_upto`) sit in the queue for a whole backlog
    if (global && aqc < 256) aqc = 256; // a shard's incoming cross links are filled a whole exchange round at a time
    h->NX.aq_cap = aqc;
    h->NX.aq_on = 0;
    h->NX.pk_base = h->cfg.start_ns;
    if (nl > 0) {
        const size_t NQ = NL * (size_t)aqc;
        if (global) {
            // a shard: the link queues other ranks may write into while this rank's kernel runs (LIVE exchange) -- uncached device
            // memory, exportable with hipIpcGetMemHandle: a peer's stores arrive in HBM past this device's L2
            void *a = nullptr, *b = nullptr, *c = nullptr;
            if ((rc = uncached_alloc(h, &a, NQ * 4 * sizeof(int64_t)))) return rc;
            if ((rc = uncached_alloc(h, &b, NL * sizeof(unsigned long long)))) return rc;
            if ((rc = uncached_alloc(h, &c, NL * sizeof(int64_t)))) return rc;
            h->NX.aq_rec = (int64_t *)a; h->NX.aq_head = (unsigned long long *)b; h->NX.aq_ea = (int64_t *)c;
            HS_HIP(h, hipMemset(a, 0, NQ * 4 * sizeof(int64_t)));
            HS_HIP(h, hipMemset(b, 0, NL * sizeof(unsigned long long)));
            HS_HIP(h, hipMemset(c, 0, NL * sizeof(int64_t)));
            ALN(aq_tail, NL);
        } else {
            ALN(aq_rec, NQ * 4); ALN(aq_tail, NL); ALN(aq_head, NL); ALN(aq_ea, NL);
        }
        ALN(early_upto, (size_t)n); ALN(d_pre, (size_t)n);
        // the whole network in one cooperative launch (shards: hs_engine_shard_round); with probes, time-varying profiles
        // or scheduled Requests the PF instantiation of the kernel (a profile's next arrival and the next scheduled Request
        // are part of next_admission() / next_time(), which is all the bounds are made of)
        // (the links' packed (bound, tail) words hold 44 bits of nanoseconds: 4.9 hours of simulated time)
        const bool fits = h->cfg.horizon_ns - h->cfg.start_ns < (int64_t)kPkNever - 2 && aqc < (1 << (kPkTailBits - 2));
        h->async_ok = !global && fits;
        h->net_pf = h->any_probe || h->any_timevarying || h->any_sched || h->any_xsrc;
        // the network's entity kinds are uniform: the specialised instantiation (hs_netstation.hpp HSU)
        h->net_uni = h->uni_stations && h->f64_times && !global && !h->net_pf && h->C == 1;
        for (int i = 0; i < n && h->net_uni; ++i) {
            if (net->egress_kind[i] != HS_EGRESS_ROUTER) { h->net_uni = false; break; }
            int n_link = 0, l1 = -1;
            const int32_t tg[4] = {rt0[(size_t)i], rt1[(size_t)i], rt2[(size_t)i], rt3[(size_t)i]};
            for (int q = 0; q < (int)rtk[(size_t)i]; ++q) if (tg[q] >= 0) { ++n_link; l1 = tg[q]; }
            if (n_link != 1 || jk[(size_t)l1] != HS_LAT_EXPONENTIAL || lloss[(size_t)l1] != 0.0) h->net_uni = false;
            if (h->net_uni && !(net->link_lat_min_s[l1] < 1e4 && (net->link_jitter_mean_s ? net->link_jitter_mean_s[l1] : 0.0) < 1e4)) h->net_uni = false;
            if (h->net_uni && in_deg_h[(size_t)i] != 1) h->net_uni = false;        // exactly one incoming link per station
        }
    }
#undef ALN
    if (!h->L.sink_created_own) {   // not every completion reaches the Sink any more: explicit created_at column
        if ((rc = dev_alloc(h, &h->L.sink_created_own, N * (size_t)h->L.cap))) return rc;
    }
    h->L.sink_created = h->L.sink_created_own;
#ifndef HS_LOGS_ROW_MAJOR   // (scratch build: the [cap][n_lp] logs the network engines had until round 5)
    h->L.lp_major = 1;          // an LP's records contiguous (hs_station.hpp RecordLogs::lp_major); nothing has been logged yet
#endif
    if (h->exact && h->cfg.mode == HS_MODE_SINGLE && !global) {
        // a whole run on the single heap (an election the lineage key does not decide: tandem_fallback) takes one list cell per
        // admitted Request, like tandem queues
        const int64_t want = std::min<int64_t>(h->xs_host.pool_cap + (int64_t)n * h->L.cap, (int64_t)1 << 28);
        if (want > h->xs_host.pool_cap) {
            if ((rc = dev_alloc(h, &h->xs_host.pnext, (size_t)want))) return rc;
            if ((rc = dev_alloc(h, &h->xs_host.pidx, (size_t)want))) return rc;
            h->xs_host.pool_cap = want;
        }
    }
    HS_HIP(h, hipDeviceSynchronize());      // (null-stream uploads and fills, as in hs_engine_set_stations)
    h->is_net = true;
    return HS_OK;
}

int hs_engine_set_stream(hs_engine *h, void *hip_stream, int external) {
    if (!h) return fail(h, HS_E_INVALID, "null handle");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    // external != 0: run on the caller's stream; a NULL handle then means the device's default (null) stream,
    // which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams
    h->stream = external ? (hipStream_t)hip_stream : h->own_stream;
    h->external_stream = external != 0;
    return HS_OK;
}

int hs_engine_shard_attach(hs_engine *h, const hs_shard *sh) {
    if (!h || !sh) return fail(h, HS_E_INVALID, "hs_engine_shard_attach: null argument");
    if (!h->is_net || !h->net_global) return fail(h, HS_E_STATE, "set a network with n_global_lp > 0 before attaching the shard");
    if (h->initialised) return fail(h, HS_E_STATE, "attach the shard before the first run");
    if (sh->world < 1 || sh->rank < 0 || sh->rank >= sh->world || !sh->shard_lo)
        return fail(h, HS_E_INVALID, "bad rank / world / shard_lo");
    if (sh->shard_lo[sh->rank] != (int64_t)h->cfg.lp_base || sh->shard_lo[sh->rank + 1] != (int64_t)h->cfg.lp_base + h->cfg.n_lp)
        return fail(h, HS_E_INVALID, "shard_lo[rank] does not match this engine's [lp_base, lp_base + n_lp)");
    if (sh->shard_lo[0] != 0 || sh->shard_lo[sh->world] != h->n_global_lp)
        return fail(h, HS_E_INVALID, "shard_lo must cover [0, n_global_lp)");
    if (!sh->outbox_dev || !sh->inbox_dev || !sh->gvt_dev || !sh->cand_dev || sh->msg_capacity < 1)
        return fail(h, HS_E_INVALID, "exchange buffers are required");
    if (sh->window_ns < 1 || sh->window_ns > h->window_ns)
        return fail(h, HS_E_INVALID, "window_ns must be the minimum lookahead over ALL shards (got %lld, this shard's links allow %lld)",
                    (long long)sh->window_ns, (long long)h->window_ns);
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int nl = h->NP.n_links;
    std::vector<int32_t> lrank((size_t)(nl > 0 ? nl : 1), sh->rank);
    for (int l = 0; l < nl; ++l) {
        const int64_t d = h->h_link_dst[(size_t)l];
        int r = 0;
        while (r + 1 < sh->world && d >= sh->shard_lo[r + 1]) ++r;
        lrank[(size_t)l] = r;
    }
    int rc;
    if ((rc = upload<int32_t>(h, &h->SC.link_rank, lrank.data(), lrank.size(), 0))) return rc;
    if ((rc = dev_alloc(h, &h->SC.wend_slots, 2))) return rc;
    h->SC.gvt_in = sh->gvt_dev; h->SC.gvt_out = sh->gvt_dev;     // re-pointed per launch (parity)
    h->SC.outbox = sh->outbox_dev; h->SC.cand_out = sh->cand_dev;
    h->SC.msg_cap = sh->msg_capacity; h->SC.row = 1 + kMsgWords * sh->msg_capacity;
    h->SC.rank = sh->rank; h->SC.world = sh->world;
    h->SC.W = sh->window_ns;
    h->SC.lp_base = (int64_t)h->cfg.lp_base;
    h->inbox = sh->inbox_dev;
    h->window_ns = sh->window_ns;
    h->shard_gvt = sh->gvt_dev;
    return HS_OK;
}

// Sharded run, driven by the host one window at a time (all calls only enqueue work on the engine's stream):
//   begin;  for k = 0, 1, ...: window(k); <all-to-all outbox -> inbox>; inject(k); <all-reduce(min) gvt[k & 1]>;
//   every so often progress() (synchronises) until the returned window end reaches end_ns;
//   final(k); <all-gather cand>; overshoot(lp) on the winner.
int hs_engine_shard_begin(hs_engine *h, int64_t end_ns) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_begin: no shard attached");
    if (end_ns > h->cfg.horizon_ns) return fail(h, HS_E_INVALID, "end_ns beyond the configured horizon");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    int rc = do_reset_async(h);
    if (rc) return rc;
    rc = ensure_reset(h);
    if (rc) return rc;
    h->SC.end_ns = end_ns;
    const int64_t init_slots[2] = {h->cfg.start_ns - 1, h->cfg.start_ns - 1};
    const int64_t init_gvt[2] = {kInfNs, h->cfg.start_ns};       // window 0 reads gvt[1], accumulates into gvt[0]
    HS_HIP(h, hipMemcpyAsync(h->SC.wend_slots, init_slots, sizeof init_slots, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemcpyAsync(h->shard_gvt, init_gvt, sizeof init_gvt, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemsetAsync(h->SC.outbox, 0, (size_t)h->SC.world * h->SC.row * 8, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));                   // the host arrays above are stack memory
    h->launches = 1;
    h->net_ran = true;
    h->ipc_parity = 0;            // (every rank starts a run on the same half of the double-buffered exchange buffers)
    return HS_OK;
}

int hs_engine_shard_window(hs_engine *h, int64_t k) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_window: no shard attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    h->SC.gvt_in = h->shard_gvt + ((k + 1) & 1);
    h->SC.gvt_out = h->shard_gvt + (k & 1);
    launch_net_dispatch(h, 0, (int)(k & 0x3fffffff), h->flags & 1);
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

int hs_engine_shard_inject(hs_engine *h, int64_t k) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_inject: no shard attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int64_t items = (int64_t)h->SC.world * h->SC.msg_cap;
    const unsigned blocks = (unsigned)((items + 255) / 256);
    hipLaunchKernelGGL(hs_shard_inject, dim3(blocks ? blocks : 1), dim3(256), 0, h->stream, h->NX, h->inbox, h->SC.outbox,
                       h->SC.world, h->SC.msg_cap, h->SC.row, h->cfg.n_lp, h->SC.lp_base, (int)(k & 1), h->gid2local,
                       h->n_gid, h->shard_gvt + ((k + 1) & 1), h->tot);
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

int hs_engine_shard_progress(hs_engine *h, int64_t k_last, int64_t *wend_out) {
    if (!h || !h->SC.wend_slots || !wend_out) return fail(h, HS_E_STATE, "hs_engine_shard_progress: no shard attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipMemcpyAsync(wend_out, h->SC.wend_slots + (k_last & 1), 8, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 4) return fail(h, HS_E_INVALID, "a message arrived for a station or link this shard does not own");
    if (t.overflow & 32) return fail(h, HS_E_OVERFLOW, "a link's loss table (hs_network.link_drop_capacity) is shorter than the packets that entered the link");
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a message bag or an exchange row overflowed; raise bag_capacity / msg_capacity");
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

// Asynchronous rounds instead of windows (same attach / begin / final / overshoot):
//   begin;  loop: round; <all-to-all outbox -> inbox>; <all-reduce(MAX) bounds>; inject_async;
//   every so often async_done() (synchronises) until no rank has work left;  final(0); <all-gather cand>; overshoot.
int hs_engine_shard_async_setup(hs_engine *h, int32_t n_cross, const int64_t *cross_gid, int64_t *bounds_dev,
                                int32_t max_iters) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_async_setup: no shard attached");
    if (n_cross < 0 || (n_cross > 0 && !cross_gid) || !bounds_dev || max_iters < 1)
        return fail(h, HS_E_INVALID, "hs_engine_shard_async_setup: bad argument");
    if (!h->NX.aq_tail) return fail(h, HS_E_STATE, "the network has no links");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    if (!ensure_async_fit(h)) return fail(h, HS_E_UNSUPPORTED, "the shard's stations are not co-resident on this device (asynchronous rounds need a cooperative launch)");
    if (h->cfg.horizon_ns - h->cfg.start_ns >= (int64_t)kPkNever - 2)
        return fail(h, HS_E_UNSUPPORTED, "asynchronous rounds hold 44 bits of nanoseconds per link bound (4.9 h of simulated time); use rounds = False");
    const int64_t lo = (int64_t)h->cfg.lp_base, hi = lo + h->cfg.n_lp;
    std::vector<int32_t> loc((size_t)(n_cross > 0 ? n_cross : 1), -1);
    std::vector<uint8_t> role((size_t)(n_cross > 0 ? n_cross : 1), 0);
    int out_here = 0;
    for (int i = 0; i < n_cross; ++i) {
        const int64_t g = cross_gid[i];
        if (g < 0 || g >= h->n_gid) return fail(h, HS_E_INVALID, "cross link id %lld out of range", (long long)g);
        const int l = h->h_gid2local[(size_t)g];
        if (l < 0) continue;
        loc[(size_t)i] = l;
        const bool s_here = h->h_link_src[(size_t)l] >= lo && h->h_link_src[(size_t)l] < hi;
        const bool d_here = h->h_link_dst[(size_t)l] >= lo && h->h_link_dst[(size_t)l] < hi;
        if (s_here && d_here) return fail(h, HS_E_INVALID, "link %lld does not cross a shard boundary", (long long)g);
        role[(size_t)i] = (uint8_t)((s_here ? 1 : 0) | (d_here ? 2 : 0));
        out_here += s_here ? 1 : 0;
    }
    // a round may append 2 x group_cap x iterations x C messages per outgoing cross link: to one outbox row on this side, to
    // one link queue on the other.  The iterations per round are clamped to what those hold.
    // (a group of a pre-sending station may append two: a departure that was not pre-sent + the next request's pre-send)
    const long long per_iter = 2ll * kAsyncGroupCap * h->C;
    long long fit = (h->NX.aq_cap / 2) / per_iter;
    if (out_here > 0) { const long long f2 = h->SC.msg_cap / (per_iter * out_here); if (f2 < fit) fit = f2; }
    if (fit < 1)
        return fail(h, HS_E_INVALID, "msg_capacity %d is too small for %d outgoing cross links x %lld messages per iteration; "
                    "raise msg_capacity", h->SC.msg_cap, out_here, per_iter);
    if (max_iters > fit) max_iters = (int32_t)fit;
    int rc;
    if ((rc = upload<int32_t>(h, &h->cross_local, loc.data(), loc.size(), -1))) return rc;
    if ((rc = upload<uint8_t>(h, &h->cross_role, role.data(), role.size(), 0))) return rc;
    h->n_cross = n_cross;
    h->cross_bounds = bounds_dev;
    h->round_iters_cfg = max_iters;
    h->shard_async = true;
    return HS_OK;
}

// ---- device-side exchange between asynchronous rounds (csrc/hs_kernels.hpp hs_shard_push) -----------------------------------
int hs_engine_shard_ipc_export(hs_engine *h, void *handles_out) {
    if (!h || !h->shard_async || !handles_out) return fail(h, HS_E_STATE, "hs_engine_shard_ipc_export: call hs_engine_shard_async_setup first");
    static_assert(sizeof(hipIpcMemHandle_t) == HS_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    if (!h->ipc_inbox) {
        const size_t nin = (size_t)2 * h->SC.world * h->SC.row, nb = (size_t)2 * h->SC.world * ((size_t)h->n_cross + 1);
        void *a = nullptr, *b = nullptr;
        // uncached: a peer's stores arrive in HBM past this device's L2, the readers use system-scope loads
        { const int rca = uncached_alloc(h, &a, nin * 8); if (rca) return rca; }
        { const int rcb = uncached_alloc(h, &b, nb * 8); if (rcb) return rcb; }
        HS_HIP(h, hipMemset(a, 0, nin * 8));
        {   // bounds start at "nothing known" (INT64_MIN), like the vector they replace
            std::vector<int64_t> init(nb, INT64_MIN);
            HS_HIP(h, hipMemcpy(b, init.data(), nb * 8, hipMemcpyHostToDevice));
        }
        HS_HIP(h, hipDeviceSynchronize());  // (the peers write into these from their own streams)
        h->ipc_inbox = (int64_t *)a; h->ipc_bounds = (int64_t *)b;
    }
    hipIpcMemHandle_t hh[2];
    HS_HIP(h, hipIpcGetMemHandle(&hh[0], h->ipc_inbox));
    HS_HIP(h, hipIpcGetMemHandle(&hh[1], h->ipc_bounds));
    memcpy(handles_out, hh, sizeof hh);
    return HS_OK;
}

int hs_engine_shard_ipc_attach(hs_engine *h, const void *all_handles) {
    if (!h || !h->ipc_inbox || !all_handles) return fail(h, HS_E_STATE, "hs_engine_shard_ipc_attach: call hs_engine_shard_ipc_export first");
    if (h->ipc_ready) return fail(h, HS_E_STATE, "hs_engine_shard_ipc_attach: already attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int world = h->SC.world;
    const hipIpcMemHandle_t *hh = (const hipIpcMemHandle_t *)all_handles;
    std::vector<int64_t *> pin((size_t)world), pbd((size_t)world);
    for (int r = 0; r < world; ++r) {
        if (r == h->SC.rank) { pin[(size_t)r] = h->ipc_inbox; pbd[(size_t)r] = h->ipc_bounds; continue; }
        void *a = nullptr, *b = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&a, hh[2 * r], hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) { h->ipc_opened.push_back(a); e = hipIpcOpenMemHandle(&b, hh[2 * r + 1], hipIpcMemLazyEnablePeerAccess); }
        if (e != hipSuccess) return fail(h, HS_E_HIP, "hipIpcOpenMemHandle (rank %d's exchange buffers): %s", r, hipGetErrorString(e));
        h->ipc_opened.push_back(b);
        pin[(size_t)r] = (int64_t *)a; pbd[(size_t)r] = (int64_t *)b;
    }
    return ipc_set_peers(h, pin, pbd);
}

int hs_engine_shard_peers_local(hs_engine *h, int64_t *const *inbox_ptrs, int64_t *const *bounds_ptrs) {
    if (!h || !h->ipc_inbox || !inbox_ptrs || !bounds_ptrs) return fail(h, HS_E_STATE, "hs_engine_shard_peers_local: call hs_engine_shard_ipc_export first");
    if (h->ipc_ready) return fail(h, HS_E_STATE, "hs_engine_shard_peers_local: already attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int world = h->SC.world;
    std::vector<int64_t *> pin(inbox_ptrs, inbox_ptrs + world), pbd(bounds_ptrs, bounds_ptrs + world);
    if (pin[(size_t)h->SC.rank] != h->ipc_inbox || pbd[(size_t)h->SC.rank] != h->ipc_bounds)
        return fail(h, HS_E_INVALID, "hs_engine_shard_peers_local: entry [rank] must be this engine's own buffers (hs_engine_shard_ipc_buffers)");
    return ipc_set_peers(h, pin, pbd);
}

int hs_engine_shard_ipc_buffers(hs_engine *h, int64_t **inbox_out, int64_t **bounds_out) {
    if (!h || !h->ipc_inbox || !inbox_out || !bounds_out) return fail(h, HS_E_STATE, "hs_engine_shard_ipc_buffers: call hs_engine_shard_ipc_export first");
    *inbox_out = h->ipc_inbox; *bounds_out = h->ipc_bounds;
    return HS_OK;
}

int hs_engine_shard_push(hs_engine *h) {
    if (!h || !h->ipc_ready) return fail(h, HS_E_STATE, "hs_engine_shard_push: call hs_engine_shard_ipc_attach first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(hs_shard_push, dim3((unsigned)h->SC.world), dim3(256), 0, h->stream, h->SC.outbox, h->SC.row, h->SC.msg_cap,
                       h->SC.world, h->SC.rank, h->ipc_parity, h->peer_inbox_dev, h->peer_bounds_dev, h->cross_bounds, h->n_cross, h->tot);
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

int hs_engine_shard_inject_ipc(hs_engine *h) {
    if (!h || !h->ipc_ready) return fail(h, HS_E_STATE, "hs_engine_shard_inject_ipc: call hs_engine_shard_ipc_attach first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(hs_shard_combine_bounds, dim3((unsigned)((h->n_cross + 1 + 255) / 256)), dim3(256), 0, h->stream, h->ipc_bounds,
                       h->SC.world, h->ipc_parity, h->n_cross, h->cross_bounds);
    hipLaunchKernelGGL(hs_shard_fetch_inbox, dim3((unsigned)h->SC.world), dim3(256), 0, h->stream, h->ipc_inbox, h->SC.world, h->ipc_parity,
                       h->SC.row, h->SC.msg_cap, const_cast<int64_t *>(h->inbox));
    HS_HIP(h, hipGetLastError());
    h->ipc_parity ^= 1;
    h->launches += 2;
    return hs_engine_shard_inject_async(h);
}

// ---- LIVE exchange (ShardCtl::live): no launch boundary per round ----------------------------------------------------------------
static int live_set_peers(hs_engine *h, const std::vector<int64_t *> &rec, const std::vector<int64_t *> &ea,
                          const std::vector<unsigned long long *> &head, const int32_t *peer_link) {
    const int world = h->SC.world;
    int rc;
    if ((rc = dev_alloc(h, &h->live_rec_dev, (size_t)world))) return rc;
    if ((rc = dev_alloc(h, &h->live_ea_dev, (size_t)world))) return rc;
    if ((rc = dev_alloc(h, &h->live_head_dev, (size_t)world))) return rc;
    HS_HIP(h, hipMemcpy(h->live_rec_dev, rec.data(), (size_t)world * sizeof(void *), hipMemcpyHostToDevice));
    HS_HIP(h, hipMemcpy(h->live_ea_dev, ea.data(), (size_t)world * sizeof(void *), hipMemcpyHostToDevice));
    HS_HIP(h, hipMemcpy(h->live_head_dev, head.data(), (size_t)world * sizeof(void *), hipMemcpyHostToDevice));
    const size_t NL = (size_t)(h->NP.n_links > 0 ? h->NP.n_links : 1);
    {
        const int32_t *pl = nullptr;
        if ((rc = upload<int32_t>(h, &pl, peer_link, NL, -1))) return rc;
        h->live_link_dev = const_cast<int32_t *>(pl);
    }
    // every link that leaves the shard needs a place in its destination's table
    const int64_t lo = (int64_t)h->cfg.lp_base, hi = lo + h->cfg.n_lp;
    for (int l = 0; l < h->NP.n_links; ++l) {
        const bool s_here = h->h_link_src[(size_t)l] >= lo && h->h_link_src[(size_t)l] < hi;
        const bool d_here = h->h_link_dst[(size_t)l] >= lo && h->h_link_dst[(size_t)l] < hi;
        if (s_here && !d_here && peer_link[l] < 0) return fail(h, HS_E_INVALID, "link %d leaves the shard but has no index at its destination rank", l);
    }
    if (!ensure_async_fit(h)) return fail(h, HS_E_UNSUPPORTED, "the shard's stations are not co-resident on this device");
    h->SC.peer_rec = h->live_rec_dev; h->SC.peer_ea = h->live_ea_dev; h->SC.peer_head = h->live_head_dev; h->SC.peer_link = h->live_link_dev;
    h->SC.live = 0;                       // (set for the duration of a live launch only: rounds and windows keep their outbox rows)
    h->live_ready = true;
    return HS_OK;
}

int hs_engine_shard_live_export(hs_engine *h, void *handles_out) {
    if (!h || !h->SC.wend_slots || !h->NX.aq_rec || !handles_out) return fail(h, HS_E_STATE, "hs_engine_shard_live_export: attach the shard first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    hipIpcMemHandle_t hh[3];
    HS_HIP(h, hipIpcGetMemHandle(&hh[0], h->NX.aq_rec));
    HS_HIP(h, hipIpcGetMemHandle(&hh[1], h->NX.aq_ea));
    HS_HIP(h, hipIpcGetMemHandle(&hh[2], h->NX.aq_head));
    memcpy(handles_out, hh, sizeof hh);
    return HS_OK;
}

int hs_engine_shard_live_attach(hs_engine *h, const void *all_handles, const int32_t *peer_link) {
    if (!h || !h->SC.wend_slots || !h->NX.aq_rec || !all_handles || !peer_link) return fail(h, HS_E_STATE, "hs_engine_shard_live_attach: attach the shard first");
    if (h->live_ready) return fail(h, HS_E_STATE, "hs_engine_shard_live_attach: already attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int world = h->SC.world;
    const hipIpcMemHandle_t *hh = (const hipIpcMemHandle_t *)all_handles;
    std::vector<int64_t *> rec((size_t)world), ea((size_t)world);
    std::vector<unsigned long long *> head((size_t)world);
    for (int r = 0; r < world; ++r) {
        if (r == h->SC.rank) { rec[(size_t)r] = h->NX.aq_rec; ea[(size_t)r] = h->NX.aq_ea; head[(size_t)r] = h->NX.aq_head; continue; }
        void *p[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < 3; ++k) {
            const hipError_t e = hipIpcOpenMemHandle(&p[k], hh[3 * r + k], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(h, HS_E_HIP, "hipIpcOpenMemHandle (rank %d's link queues): %s", r, hipGetErrorString(e));
            h->ipc_opened.push_back(p[k]);
        }
        rec[(size_t)r] = (int64_t *)p[0]; ea[(size_t)r] = (int64_t *)p[1]; head[(size_t)r] = (unsigned long long *)p[2];
    }
    return live_set_peers(h, rec, ea, head, peer_link);
}

// enqueue the whole run of this rank (hs_engine_shard_begin set its end): ONE launch of the asynchronous engine, which exchanges
// messages and bounds with the other ranks' launches while it runs.  Every rank must have passed hs_engine_shard_begin (a barrier
// of the caller's) before any rank gets here.
int hs_engine_shard_live_run(hs_engine *h) {
    if (!h || !h->live_ready) return fail(h, HS_E_STATE, "hs_engine_shard_live_run: call hs_engine_shard_live_attach first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    if (!h->live_stream) {
        HS_HIP(h, hipStreamCreateWithFlags(&h->live_stream, hipStreamNonBlocking));
        HS_HIP(h, hipEventCreateWithFlags(&h->live_ev, hipEventDisableTiming));
    }
    // behind everything enqueued on the engine's stream so far (the reset), on a stream of its own: the launch waits for the peers'
    // launches, which -- shards of one process -- may sit on the very stream this engine shares with them
    HS_HIP(h, hipEventRecord(h->live_ev, h->stream));
    HS_HIP(h, hipStreamWaitEvent(h->live_stream, h->live_ev, 0));
    NetState NX = h->NX;
    NX.aq_on = 1;
    h->round_iters = 0;
    h->SC.live = 1;
    const int64_t end_ns = h->SC.end_ns;
    const hipStream_t keep = h->stream;
    h->stream = h->live_stream;
    const hipError_t e = h->C == 1 ? launch_async<1>(h, end_ns, NX) : h->C == 2 ? launch_async<2>(h, end_ns, NX) : launch_async<4>(h, end_ns, NX);
    h->stream = keep;
    h->SC.live = 0;
    if (e != hipSuccess) return fail(h, HS_E_HIP, "launch of the live asynchronous engine failed: %s", hipGetErrorString(e));
    h->launches += 1;
    return HS_OK;
}

int hs_engine_shard_live_wait(hs_engine *h) {
    if (!h || !h->live_ready) return fail(h, HS_E_STATE, "hs_engine_shard_live_wait: not attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    if (h->live_stream) HS_HIP(h, hipStreamSynchronize(h->live_stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 8) return fail(h, HS_E_HIP, "the asynchronous engine gave up waiting for a neighbour (bounded spin exhausted): did every rank launch?");
    if (t.overflow & 32) return fail(h, HS_E_OVERFLOW, "a link's loss table (hs_network.link_drop_capacity) is shorter than the packets that entered the link");
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a message bag or a link queue overflowed; raise bag_capacity");
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_shard_round(hs_engine *h) {
    if (!h || !h->shard_async) return fail(h, HS_E_STATE, "hs_engine_shard_round: call hs_engine_shard_async_setup first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    NetState NX = h->NX;
    NX.aq_on = 1;
    h->round_iters = h->round_iters_cfg;
    const int64_t end_ns = h->SC.end_ns;
    hipError_t e = h->C == 1 ? launch_async<1>(h, end_ns, NX) : h->C == 2 ? launch_async<2>(h, end_ns, NX) : launch_async<4>(h, end_ns, NX);
    h->round_iters = 0;
    if (e != hipSuccess) return fail(h, HS_E_HIP, "cooperative launch failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(hs_shard_bounds_out, dim3((unsigned)((h->n_cross + 1 + 255) / 256)), dim3(256), 0, h->stream, h->NX.aq_ea,
                       h->NX.pk_base, h->cross_local, h->cross_role, h->n_cross, h->cross_bounds, h->tot);
    HS_HIP(h, hipGetLastError());
    h->launches += 2;
    return HS_OK;
}

int hs_engine_shard_inject_async(hs_engine *h) {
    if (!h || !h->shard_async) return fail(h, HS_E_STATE, "hs_engine_shard_inject_async: call hs_engine_shard_async_setup first");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(hs_shard_inject_async, dim3(1), dim3(256), 0, h->stream, h->NX, h->inbox, h->SC.outbox, h->SC.world,
                       h->SC.msg_cap, h->SC.row, h->cfg.n_lp, h->SC.lp_base, h->gid2local, h->n_gid, h->cross_local,
                       h->cross_role, h->n_cross, h->cross_bounds, h->tot);
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

// synchronises; *any_not_done = the all-reduced "some rank still has work" flag of the last exchanged round
int hs_engine_shard_async_done(hs_engine *h, int32_t *any_not_done) {
    if (!h || !h->shard_async || !any_not_done) return fail(h, HS_E_STATE, "hs_engine_shard_async_done: not set up");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    int64_t v = 0;
    HS_HIP(h, hipMemcpyAsync(&v, h->cross_bounds + h->n_cross, 8, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    *any_not_done = v != 0;
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 4) return fail(h, HS_E_INVALID, "a message arrived for a station or link this shard does not own");
    if (t.overflow & 8) return fail(h, HS_E_HIP, "the asynchronous engine gave up waiting for a neighbour (bounded spin exhausted)");
    if (t.overflow & 32) return fail(h, HS_E_OVERFLOW, "a link's loss table (hs_network.link_drop_capacity) is shorter than the packets that entered the link");
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a message bag, a link queue or an exchange row overflowed; raise bag_capacity / msg_capacity");
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_shard_final(hs_engine *h, int64_t k) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_final: no shard attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    h->SC.gvt_in = h->shard_gvt + ((k + 1) & 1);
    h->SC.gvt_out = h->shard_gvt + (k & 1);
    launch_net_dispatch(h, h->SC.end_ns, (int)(k & 0x3fffffff), (h->flags & 1) | 2 | 4 | ((h->shard_async || h->live_ready) ? 8 : 0));
    HS_HIP(h, hipGetLastError());
    h->launches++;
    h->final_win = k;
    return HS_OK;
}

int hs_engine_shard_overshoot(hs_engine *h, int32_t lp) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_overshoot: no shard attached");
    if (lp < 0 || lp >= h->cfg.n_lp) return fail(h, HS_E_INVALID, "LP index %d out of range", lp);
    HS_HIP(h, hipSetDevice(h->cfg.device));
    const int win = (int)(h->final_win & 0x3fffffff);
    switch (h->C) {
        case 1: hipLaunchKernelGGL(hs_shard_overshoot<1>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L, h->tot, h->cfg.n_lp, lp, win, h->SC); break;
        case 2: hipLaunchKernelGGL(hs_shard_overshoot<2>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L, h->tot, h->cfg.n_lp, lp, win, h->SC); break;
        default: hipLaunchKernelGGL(hs_shard_overshoot<4>, dim3(1), dim3(64), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L, h->tot, h->cfg.n_lp, lp, win, h->SC); break;
    }
    HS_HIP(h, hipGetLastError());
    h->launches++;
    return HS_OK;
}

static int results_final(hs_engine *h);
int hs_engine_get_net_stats(hs_engine *h, const hs_net_stats *o) {
    if (!h || !o) return fail(h, HS_E_INVALID, "hs_engine_get_net_stats: null argument");
    if (!h->is_net) return fail(h, HS_E_STATE, "no network set");
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->cfg.n_lp, nl = (size_t)h->NP.n_links;
    if (o->routed) HS_HIP(h, hipMemcpy(o->routed, h->NX.routed, n * 8, hipMemcpyDeviceToHost));
    if (o->link_entered && nl) HS_HIP(h, hipMemcpy(o->link_entered, h->NX.link_in, nl * 8, hipMemcpyDeviceToHost));
    if (o->link_packets_sent && nl) HS_HIP(h, hipMemcpy(o->link_packets_sent, h->NX.link_packets, nl * 8, hipMemcpyDeviceToHost));
    if (o->link_packets_dropped && nl) {                     // entered - not lost
        std::vector<int64_t> in(nl), sent(nl);
        HS_HIP(h, hipMemcpy(in.data(), h->NX.link_in, nl * 8, hipMemcpyDeviceToHost));
        HS_HIP(h, hipMemcpy(sent.data(), h->NX.link_sent, nl * 8, hipMemcpyDeviceToHost));
        for (size_t l = 0; l < nl; ++l) o->link_packets_dropped[l] = in[l] - sent[l];
    }
    return HS_OK;
}

int hs_engine_set_link_drops(hs_engine *h, int32_t link, const uint32_t *bits, int64_t n_bits) {
    if (!h || !h->is_net) return fail(h, HS_E_STATE, "hs_engine_set_link_drops: no network set");
    if (link < 0 || link >= h->NP.n_links || !h->drop_bits_dev) return fail(h, HS_E_INVALID, "link %d has no loss table", (int)link);
    const int64_t b0 = h->drop_off_host[(size_t)link], cap = h->drop_off_host[(size_t)link + 1] - b0;
    if (cap <= 0) return fail(h, HS_E_INVALID, "link %d has no loss table", (int)link);
    if (n_bits < 0 || n_bits > cap || (n_bits > 0 && !bits)) return fail(h, HS_E_INVALID, "link %d: %lld bits for a table of %lld", (int)link, (long long)n_bits, (long long)cap);
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    HS_HIP(h, hipMemset(h->drop_bits_dev + b0 / 32, 0, (size_t)(cap / 32) * sizeof(uint32_t)));
    if (n_bits > 0) HS_HIP(h, hipMemcpy(h->drop_bits_dev + b0 / 32, bits, (size_t)((n_bits + 31) / 32) * sizeof(uint32_t), hipMemcpyHostToDevice));
    HS_HIP(h, hipDeviceSynchronize());
    return HS_OK;
}

int64_t hs_engine_read_send_log(hs_engine *h, int64_t *out_triples, int64_t capacity) {
    if (!h || !h->is_net) return fail(h, HS_E_STATE, "hs_engine_read_send_log: no network set");
    if (!h->loss_host.send_log_n) return 0;
    { const int rcf = results_final(h); if (rcf) return rcf; }
    unsigned long long n = 0;
    HS_HIP(h, hipMemcpy(&n, h->loss_host.send_log_n, sizeof n, hipMemcpyDeviceToHost));
    const int64_t have = (int64_t)(n < (unsigned long long)h->loss_host.send_log_cap ? n : (unsigned long long)h->loss_host.send_log_cap);
    const int64_t take = have < capacity ? have : capacity;
    if (take > 0 && out_triples) HS_HIP(h, hipMemcpy(out_triples, h->loss_host.send_log, (size_t)take * 3 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return (int64_t)n;
}

int hs_engine_reset(hs_engine *h) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "hs_engine_reset: stations not set");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    if (h->net_on_heap) { h->exact_only = false; h->net_on_heap = false; }      // (the next run starts on the parallel engines again)
    int rc = do_reset_async(h);
    if (rc) return rc;
    HS_HIP(h, hipStreamSynchronize(h->stream));
    return HS_OK;
}

// Windows over a network: what a run_until with a later end does.  0: repeat the run from the start; 1: continue from the state the
// last run left; 2: nothing moves.
static int net_window_state(hs_engine *h, int64_t end_ns, int &state, const Totals *seen) {
    state = 0;
    if (h->net_global || (h->flags & (1 << 24)) || h->NX.pend_pay == nullptr) return HS_OK;
    Totals t;
    int phase = h->xs_phase_seen;
    if (seen) t = *seen;
    else {
        HS_HIP(h, hipStreamSynchronize(h->stream));
        HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
        if (h->exact && h->xs && !h->XI.per_lp) HS_HIP(h, hipMemcpy(&phase, &h->xs->phase, sizeof phase, hipMemcpyDeviceToHost));
    }
    // An engine with a prologue (hs_exact.hpp) continues while the prologue is skipped (lazy_prologue: a hazard repeats ALL windows
    // behind it, prologue_fallback) or once it has handed over to the parallel engines (phase 2: its launches are no-ops from then on);
    // in between -- the single lane still holds the heap at a window end -- the run is repeated
    if (h->exact && !lazy_active(h) && phase != 2) return HS_OK;
    // (undecided bit 2 -- a pre-run root beside another root of its nanosecond -- only matters while the prologue is skipped)
    const int und = lazy_active(h) ? t.undecided : (t.undecided & ~4);
    if (t.overflow != 0 || t.qoverflow != 0 || und != 0 || t.no_resume != 0) return HS_OK;
    state = t.cur_time > end_ns ? 2 : 1;
    return HS_OK;
}

int hs_engine_run_until_async(hs_engine *h, int64_t end_ns) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "hs_engine_run_until: stations not set");
    if (end_ns > h->cfg.horizon_ns)
        return fail(h, HS_E_INVALID, "end_ns %lld beyond the configured horizon %lld", (long long)end_ns,
                    (long long)h->cfg.horizon_ns);
    HS_HIP(h, hipSetDevice(h->cfg.device));
    h->launches = 0;
    const bool seen_valid = h->tot_seen_valid && !h->pending_async;
    h->tot_seen_valid = false;
    h->net_resume = false;
    HS_HIP(h, hipEventRecord(h->ev_a, h->stream));
    if (!h->initialised) { int rc = do_reset_async(h); if (rc) return rc; h->launches++; }
    HS_HIP(h, hipEventRecord(h->ev_k0, h->stream));
    if (h->is_net) {
        if (h->net_global) return fail(h, HS_E_STATE, "a shard of a partitioned network is driven with hs_engine_shard_*");
        h->net_window_path = 0;
        if (h->exact_only) {
            // the single-heap loop (hs_exact.hpp, no hand-over) holds the whole network since an election its lineage key did not
            // decide (tandem_fallback): `_execute_until` again is that loop again -- the heap keeps every pending event
            h->window_ends.push_back(end_ns);
            if (end_ns > h->net_last_end) h->net_last_end = end_ns;
            h->net_window_path = 1;
            int rc = launch_prologue(h, end_ns);
            if (rc) return rc;
            h->net_ran = true;
            HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
            HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
            h->pending_async = true;
            return HS_OK;
        }
        if (h->net_ran) {
            // Windows over a network (core/simulation.py:527-541 `_run_window` = `_execute_until` again).  The reference pops events in
            // one global order whatever the window ends are, and every call stops behind the first event beyond its end -- so after
            // windows e_1 <= ... <= e_k the state is that of ONE run to e_k.  An end at or before the last one moves nothing: the
            // reference's loop condition `current_time <= end` is already false (the event beyond the earlier end was processed).
            if (end_ns <= h->net_last_end) {
                h->net_window_path = 2;
                HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
                HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
                h->pending_async = true;
                return HS_OK;
            }
            // Round 6: a later end CONTINUES from the state the last run left -- every station's rows, the bags (the final launch took
            // the link queues' leftovers into them), the links' bounds (lower bounds whatever the end was: NetStation::pre_send) and
            // the group the election stopped inside (hs_net_resume) -- O(window) as the reference's `_run_window`, not O(prefix).
            // Not for shards; a prologue that still holds the single-lane heap (net_window_state), a state the asynchronous kernel
            // cannot take back (a bag larger than its LDS column, an overflow) and debug flag 1 << 24 repeat the run from the start
            // as rounds 4-5 did.
            int state = 0;
            { const int rc = net_window_state(h, end_ns, state, seen_valid ? &h->tot_seen : nullptr); if (rc) return rc; }
            h->net_window_path = state == 0 ? 3 : state;
            if (state == 2) {            // the event beyond the last end lies beyond this one too: `current_time <= end` is false
                h->net_last_end = end_ns;
                HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
                HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
                h->pending_async = true;
                return HS_OK;
            }
            if (state == 1) { h->net_resume = true; h->net_resume_from = h->net_last_end; }
            else {
                int rc = do_reset_async(h);
                if (rc) return rc;
                h->launches++;
            }
        }
        h->net_last_end = end_ns;
        if (h->exact) h->window_ends.push_back(end_ns);       // (a skipped prologue's hazards and an undecided election repeat the run)
        int rc = launch_prologue(h, end_ns);
        if (rc) return rc;
        rc = run_net_async(h, end_ns);
        h->net_resume = false;
        if (rc) return rc;
    } else {
        if (h->n_pass > 0 && ((h->flags & (1 << 17)) || ((h->flags & (1 << 16)) && h->exact_prologue)) && h->exact && h->window_ends.empty())
            h->exact_only = true;   // debug: single heap from the start (1 << 16: wherever a prologue exists -- with tandem queues that is this loop)
        if (h->n_pass > 0 || lazy_active(h) || ((h->any_xsrc || h->any_sched) && h->exact)) h->window_ends.push_back(end_ns);
        int rc = launch_prologue(h, end_ns);
        if (rc) return rc;
        rc = launch_run_dispatch(h, end_ns);
        if (rc) return rc;
        HS_HIP(h, hipGetLastError());
        h->launches++;
    }
    HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
    HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
    h->pending_async = true;
    return HS_OK;
}

// A station engine that skipped the prologue (lazy_prologue): did the run meet what only the prologue orders exactly?
static bool lazy_hazard(hs_engine *h, bool &hazard) {
    Totals t;
    if (hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost) != hipSuccess) return false;
    unsigned long long total = 0;
    for (int k = 0; k < HS_EV_KINDS; ++k) total += t.ev[k];
    // (the election of the event beyond end_ns ranks a pre-run event first: true once the run has created n_init events)
    hazard = (t.undecided & 4) != 0 || total < 2ull * (unsigned long long)h->n_init;
    return true;
}
// The single-heap machinery (hs_exact.hpp) for an engine whose model has no pre-run events beyond its Sources' first ticks -- built
// on demand, the first time an election needs the reference's sort-index ledger (tandem_fallback).  Mirrors the HS_MODE_SINGLE block
// of hs_engine_set_stations with no Probes and no scheduled Requests.
int setup_exact_plain(hs_engine *h) {
    const int n = h->cfg.n_lp;
    int rc;
    const int32_t zero32 = 0; const uint8_t zero8 = 0; const int64_t zero64 = 0;
    if ((rc = upload<int32_t>(h, &h->XI.src_lp, h->h_src_lp.data(), h->h_src_lp.size(), 0))) return rc;
    if ((rc = upload<uint8_t>(h, &h->XI.src_slot, h->h_src_slot.data(), h->h_src_slot.size(), 0))) return rc;
    if ((rc = upload<int32_t>(h, &h->XI.probe_lp, &zero32, 0, 0))) return rc;
    if ((rc = upload<uint8_t>(h, &h->XI.probe_slot, &zero8, 0, 0))) return rc;
    if ((rc = upload<int32_t>(h, &h->XI.sched_lp, &zero32, 0, 0))) return rc;
    if ((rc = upload<int64_t>(h, &h->XI.sched_entry, &zero64, 0, 0))) return rc;
    if ((rc = upload<int64_t>(h, &h->XI.sched_rank, &zero64, 0, 0))) return rc;
    h->XI.n_src = (int32_t)h->h_src_lp.size(); h->XI.n_probe = 0; h->XI.n_sched = 0; h->XI.per_lp = 0;
    if ((rc = dev_alloc(h, &h->XI.sched_idx, (size_t)1))) return rc;
    HS_HIP(h, hipMemset(h->XI.sched_idx, 0, sizeof(uint32_t)));
    const int64_t n_init = (int64_t)h->h_src_lp.size();
    h->n_init = n_init;
    h->xs_host = XState{};
    h->xs_host.heap_cap = n_init + (int64_t)n * (h->C + 16) + 1024;
    h->xs_host.pool_cap = std::min<int64_t>(2 * n_init + 16 * (int64_t)n + 1024 + (int64_t)n * h->L.cap, (int64_t)1 << 28);
    if ((rc = dev_alloc(h, &h->xs_host.heap, (size_t)h->xs_host.heap_cap))) return rc;
    if ((rc = dev_alloc(h, &h->xs_host.qhead, (size_t)n))) return rc;
    if ((rc = dev_alloc(h, &h->xs_host.qtail, (size_t)n))) return rc;
    if ((rc = dev_alloc(h, &h->xs_host.pnext, (size_t)h->xs_host.pool_cap))) return rc;
    if ((rc = dev_alloc(h, &h->xs_host.pidx, (size_t)h->xs_host.pool_cap))) return rc;
    if ((rc = dev_alloc(h, &h->xs_host.init_t, (size_t)std::max<int64_t>(n_init, 1)))) return rc;
    if ((rc = dev_alloc(h, &h->xs, 1))) return rc;
    HS_HIP(h, hipDeviceSynchronize());
    h->exact = true;
    return HS_OK;
}

constexpr unsigned long long kSingleHeapReplayMax = 4000000ull;   // events a network's undecided election may cost on the single lane (~2 us each)
// Tandem queues: the passes met an order between two LPs' events that their lineage key does not decide (Totals::undecided).
// The run since the last reset is repeated, window by window, on the single-heap loop -- the reference's own algorithm.
int tandem_fallback(hs_engine *h) {
    // (also engines with several Sources per Server whose election rested on a departure's construction rank: set_stations)
    // (round 6: and station NETWORKS held by one engine whose election of the one event beyond end_time rested on a stand-in rank --
    //  hs_net_window's tie check: until now refused by name; the single heap IS the reference's sort-index ledger)
    const bool net_election = h->is_net && !h->net_global && h->cfg.mode == HS_MODE_SINGLE;
    const bool station_election = (h->any_xsrc || h->any_sched) && !h->is_net && h->cfg.mode == HS_MODE_SINGLE;
    if ((h->n_pass == 0 && !station_election && !net_election) || h->exact_only) return HS_OK;
    if (!h->exact && !net_election) return HS_OK;
    int und = 0;
    HS_HIP(h, hipMemcpy(&und, &h->tot->undecided, sizeof und, hipMemcpyDeviceToHost));
    if (h->n_pass == 0) {                     // several Sources per Server, no tandem queues: only the election's tie (value 2) moves the
        und &= 2;                             // run to the single heap; a skipped prologue's hazards (value 4) are prologue_fallback's
        if (!und) return HS_OK;
    }
    if (net_election && und) {
        // one lane replays the whole run: only while that is seconds (lock-step constants live in small models); beyond it the run
        // stays refused by name (hs_engine_run_until)
        Totals t;
        HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
        unsigned long long total = 0;
        for (int k = 0; k < HS_EV_KINDS; ++k) total += t.ev[k];
        if (total > kSingleHeapReplayMax) return HS_OK;
        if (!h->exact) {                       // a model without Probes / scheduled Requests / further Sources: no machinery yet
            if (h->net_last_end == INT64_MIN) return HS_OK;
            h->window_ends.assign(1, h->net_last_end);      // (plain networks keep no list of ends: the state is that of ONE run to the last one)
            const int rcx = setup_exact_plain(h);
            if (rcx) return rcx;
        }
    }
    if (!und && lazy_active(h)) {             // (pre-run events next to tandem queues: lazy_prologue's short-run rule)
        bool hazard = false;
        if (!lazy_hazard(h, hazard)) return fail(h, HS_E_HIP, "reading the totals failed");
        und = hazard ? 4 : 0;
    }
    if (!und) return HS_OK;
    if (h->window_ends.empty()) return HS_OK;     // (nothing ran since the last reset: no run to repeat -- ADVICE r4)
    const std::vector<int64_t> ends = h->window_ends;
    h->exact_only = true;
    h->net_on_heap = net_election;
    int rc = do_reset_async(h);
    if (rc) return rc;
    h->window_ends = ends;
    for (int64_t e : ends) {
        rc = launch_prologue(h, e);
        if (rc) return rc;
    }
    HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
    HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    return HS_OK;
}

int prologue_fallback(hs_engine *h) {
    if (!lazy_active(h) || h->window_ends.empty()) return HS_OK;
    bool hazard = false;
    if (!lazy_hazard(h, hazard)) return fail(h, HS_E_HIP, "reading the totals failed");
    if (!hazard) return HS_OK;
    const std::vector<int64_t> ends = h->window_ends;
    h->lazy_failed = true;
    int rc = do_reset_async(h);
    if (rc) return rc;
    int64_t prev = INT64_MIN;
    for (int64_t e : ends) {
        rc = launch_prologue(h, e);
        if (rc) return rc;
        if (h->is_net) {
            // (round 6: the windows of a network continue from one another -- hs_engine_run_until_async -- so does their repetition)
            if (prev != INT64_MIN) { h->net_resume = true; h->net_resume_from = prev; }
            rc = run_net_async(h, e);
            h->net_resume = false;
            if (rc) return rc;
        }
        else { rc = launch_run_dispatch(h, e); if (rc) return rc; h->launches++; }
        HS_HIP(h, hipGetLastError());
        prev = e;
    }
    // the reset above forgot the network's last window end; an earlier or equal end after this moves nothing
    if (h->is_net) h->net_last_end = ends.back();
    HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
    HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    return HS_OK;
}

int hs_engine_prologue_path(const hs_engine *h) {
    if (!h || !h->exact || (!h->exact_prologue && !h->exact_only)) return 0;
    return lazy_active(h) ? 1 : 2;
}
int hs_engine_window_path(const hs_engine *h) { return h ? h->net_window_path : 0; }
int hs_engine_tandem_path(const hs_engine *h) { return !h || h->n_pass == 0 ? 0 : h->exact_only ? 2 : 1; }

int hs_engine_synchronize(hs_engine *h) {
    if (!h) return fail(h, HS_E_INVALID, "null handle");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    { int rc = tandem_fallback(h); if (rc) return rc; }
    { int rc = prologue_fallback(h); if (rc) return rc; }
    h->pending_async = false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) h->last_run_ms = ms;
    if (hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1) == hipSuccess) h->last_kernel_ms = ms;
    return HS_OK;
}

// Every getter goes through here: a run enqueued with hs_engine_run_until_async is FINAL only behind hs_engine_synchronize -- that is
// where a run that skipped the prologue, or tandem passes that met an undecided tie, are repeated on the single heap (ADVICE r3:
// a getter that merely waited for the stream returned the results of the skipped path).
static int results_final(hs_engine *h) {
    { const int rcr = ensure_reset(h); if (rcr) return rcr; }      // (a reset whose bootstrap was deferred into a run that never came)
    if (h->pending_async) return hs_engine_synchronize(h);
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, HS_E_HIP, "device synchronisation failed");
    return HS_OK;
}

int hs_engine_run_until(hs_engine *h, int64_t end_ns) {
    int rc = hs_engine_run_until_async(h, end_ns);
    if (rc) return rc;
    rc = hs_engine_synchronize(h);
    if (rc) return rc;
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    h->tot_seen = t; h->tot_seen_valid = h->is_net;
    if (h->is_net && h->exact && h->xs && !h->XI.per_lp) HS_HIP(h, hipMemcpy(&h->xs_phase_seen, &h->xs->phase, sizeof(int), hipMemcpyDeviceToHost));
    if (h->n_tab_rows > 0) {
        unsigned long long st[2] = {0ull, 0ull};
        HS_HIP(h, hipMemcpy(st, h->tab_status, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] != 0ull)
            return fail(h, HS_E_UNSUPPORTED, "LP %lld: one arrival of its time-varying Source needs more than 64 x %lld adaptive-Simpson "
                        "intervals (the reference's own integrator needs minutes for such an arrival; hs_engine_set_profile_budget "
                        "raises the limit, tools/profile_cost.py shows the cost) -- refused instead of stalling the device",
                        (long long)st[0] - 2, (long long)h->lane_budget);
        if (st[1] != 0ull)
            return fail(h, HS_E_OVERFLOW, "LP %lld: the tick table of its Source / Probe overflowed (capacity %lld ticks); raise log_capacity",
                        (long long)st[1] - 2, (long long)h->tab_cap);
    }
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (h->is_net && (t.undecided & 2))
        return fail(h, HS_E_UNSUPPORTED, "the one event beyond end_time is a lock-step tie between two stations (same time, creation time and "
                    "lineage) that only the reference's sort-index ledger decides, and one of the two is a departure, a message or an injected "
                    "Request, whose construction rank the network engines do not carry: refused instead of guessing (constant arrivals, "
                    "services and link latencies in lock step; a different end_time or seed-free jitter avoids it; runs of up to "
                    "4 000 000 events are repeated on the single-heap loop and answered instead)");
    if (t.overflow & 16)
        return fail(h, HS_E_OVERFLOW, "the prologue (csrc/hs_exact.hpp) ran out of heap / payload-pool space");
    if (t.overflow & 8)
        return fail(h, HS_E_HIP, "the asynchronous network engine gave up waiting for a neighbour (bounded spin); "
                                 "set debug flag 16 to use the windowed engine");
    if (t.overflow & 32) return fail(h, HS_E_OVERFLOW, "a link's loss table (hs_network.link_drop_capacity) is shorter than the packets that entered the link");
    if (t.overflow & 2)
        return fail(h, HS_E_OVERFLOW, "a station's in-flight message bag overflowed (capacity %d); raise bag_capacity",
                    (int)h->NX.bag_cap);
    if (t.overflow)
        return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_bench_runs(hs_engine *h, int64_t end_ns, int32_t repeats, float *kernel_ms_out, float *total_ms_out) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "hs_engine_bench_runs: stations not set");
    if (repeats <= 0) return fail(h, HS_E_INVALID, "repeats must be > 0");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    std::vector<hipEvent_t> ev((size_t)repeats * 2 + 2);
    for (auto &e : ev) HS_HIP(h, hipEventCreate(&e));
    // Round 6: the kernel duration is SAMPLED on every eighth repeat instead of bracketed on every one.  An event pair is two barrier
    // packets on the stream; around a 90 us step (the strong shard) they were 13.7 us of every step the caller's clock saw (measured:
    // 0.1078 against 0.0941 ms per step).  A repeat without a pair reports the latest sample.  HS_BENCH_STEP_EVENTS=1 brackets every
    // repeat as before, =0 none (every repeat then reports the batch average).
    static const int step_events_mode = []() { const char *e = getenv("HS_BENCH_STEP_EVENTS"); return e ? (e[0] == '0' ? 0 : 1) : 2; }();
    auto step_events_at = [&](int r) { return step_events_mode == 1 || (step_events_mode == 2 && (r & 7) == 0); };
    for (int attempt = 0; attempt < 2; ++attempt) {
        HS_HIP(h, hipEventRecord(ev[(size_t)repeats * 2], h->stream));
        for (int r = 0; r < repeats; ++r) {
            int rc = do_reset_async(h);
            if (rc) return rc;
            if (step_events_at(r)) HS_HIP(h, hipEventRecord(ev[(size_t)2 * r], h->stream));
            { int rc1 = launch_prologue(h, end_ns); if (rc1) return rc1; }
            if (h->is_net) { h->launches = 0; int rc2 = run_net_async(h, end_ns); if (rc2) return rc2; }
            else { const int rc2 = launch_run_dispatch(h, end_ns); if (rc2) return rc2; }
            HS_HIP(h, hipGetLastError());
            if (step_events_at(r)) HS_HIP(h, hipEventRecord(ev[(size_t)2 * r + 1], h->stream));
        }
        HS_HIP(h, hipEventRecord(ev[(size_t)repeats * 2 + 1], h->stream));
        HS_HIP(h, hipStreamSynchronize(h->stream));
        if (!lazy_active(h)) break;          // (lazy_prologue: a run that needs the prologue is timed with it)
        bool hazard = false;
        if (!lazy_hazard(h, hazard)) return fail(h, HS_E_HIP, "reading the totals failed");
        if (!hazard) break;
        h->lazy_failed = true;
    }
    float tot_ms = 0.f;
    HS_HIP(h, hipEventElapsedTime(&tot_ms, ev[(size_t)repeats * 2], ev[(size_t)repeats * 2 + 1]));
    float sample_ms = tot_ms / (float)repeats;
    for (int r = 0; r < repeats; ++r) {
        if (step_events_at(r)) HS_HIP(h, hipEventElapsedTime(&sample_ms, ev[(size_t)2 * r], ev[(size_t)2 * r + 1]));
        if (kernel_ms_out) kernel_ms_out[r] = sample_ms;
        h->last_kernel_ms = sample_ms;
    }
    if (total_ms_out) *total_ms_out = tot_ms;
    h->last_run_ms = tot_ms / (float)repeats;
    if (!h->is_net) h->launches = 2;
    for (auto &e : ev) hipEventDestroy(e);
    Totals t;                                                       // a timed run that overflowed is not a result
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if ((h->n_pass > 0 || h->any_xsrc || h->any_sched) && !h->is_net && !h->exact_only && (t.undecided & 3))
        return fail(h, HS_E_UNSUPPORTED, "tandem queues / several Sources per Server: this configuration needs the single-heap path (lock-step ties); hs_engine_run_until "
                                         "switches to it, hs_engine_bench_runs does not");
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 8) return fail(h, HS_E_HIP, "the asynchronous network engine gave up waiting for a neighbour (bounded spin)");
    if (t.overflow & 32) return fail(h, HS_E_OVERFLOW, "a link's loss table (hs_network.link_drop_capacity) is shorter than the packets that entered the link");
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a station's in-flight message bag overflowed (capacity %d); raise bag_capacity", (int)h->NX.bag_cap);
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_set_profile_budget(hs_engine *h, int64_t intervals_per_lane) {
    if (!h) return fail(h, HS_E_INVALID, "hs_engine_set_profile_budget: null handle");
    if (intervals_per_lane < 1) return fail(h, HS_E_INVALID, "hs_engine_set_profile_budget: the budget must be >= 1");
    h->lane_budget = (long long)intervals_per_lane;
    h->tables_built = false;       // the next reset produces the tables again
    return HS_OK;
}

int hs_engine_get_summary(hs_engine *h, hs_summary *out) {
    if (!h || !out) return fail(h, HS_E_INVALID, "hs_engine_get_summary: null argument");
    if (!h->have_stations) return fail(h, HS_E_STATE, "stations not set");
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    int64_t total = 0;
    for (int k = 0; k < HS_EV_KINDS; ++k) { out->events_by_kind[k] = (int64_t)t.ev[k]; total += (int64_t)t.ev[k]; }
    out->events_processed = total;
    out->events_cancelled = 0;
    out->final_time_ns = (h->cfg.mode == HS_MODE_SINGLE) ? t.cur_time : t.final_time;
    out->requests_completed = (int64_t)t.completed;
    out->sink_records = (int64_t)t.received;
    out->last_run_ms = h->last_run_ms;
    out->kernel_ms = h->last_kernel_ms;
    out->launches = h->launches;
    out->window_ns = h->window_ns;
    out->overflow = t.overflow;
    return HS_OK;
}

int hs_engine_get_lp_stats(hs_engine *h, const hs_lp_stats *o) {
    if (!h || !o) return fail(h, HS_E_INVALID, "hs_engine_get_lp_stats: null argument");
    if (!h->have_stations) return fail(h, HS_E_STATE, "stations not set");
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->cfg.n_lp;
#define DL(dst, src, T) if (o->dst) HS_HIP(h, hipMemcpy(o->dst, h->X.src, n * sizeof(T), hipMemcpyDeviceToHost))
    DL(generated, generated, int64_t); DL(accepted, accepted, int64_t); DL(dropped, dropped, int64_t);
    DL(completed, completed, int64_t); DL(rejected, rejected, int64_t); DL(total_service_s, total_service, double);
    DL(sink_received, received, int64_t); DL(queue_depth, buf, int64_t); DL(active, active, int32_t);
    DL(events, events, int64_t); DL(final_time_ns, last_time, int64_t);
#undef DL
    return HS_OK;
}

int64_t hs_engine_read_sink(hs_engine *h, int32_t lp, int64_t *t_ns, int64_t *created_ns, int64_t cap) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "stations not set");
    if (lp < 0 || lp >= h->cfg.n_lp) return fail(h, HS_E_INVALID, "LP index %d out of range", lp);
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(h, HS_E_HIP, "device synchronisation failed");
    int64_t cnt = 0;
    if (hipMemcpy(&cnt, h->X.received + lp, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, HS_E_HIP, "memcpy");
    if (cnt > h->L.cap) cnt = h->L.cap;
    if (cnt > cap) cnt = cap;
    if (cnt > 0) {
        int64_t *tmp = nullptr;
        if (hipMalloc(&tmp, (size_t)cnt * 8) != hipSuccess) return fail(h, HS_E_HIP, "hipMalloc of the read-back staging buffer failed");
        const int64_t *cols[2] = {h->L.sink_t, h->L.sink_created};
        int64_t *dsts[2] = {t_ns, created_ns};
        for (int c = 0; c < 2; ++c) {
            if (!dsts[c]) continue;
            hipLaunchKernelGGL(hs_gather_one, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, cols[c], tmp,
                               h->cfg.n_lp, lp, cnt, h->L.lp_major ? h->L.cap : (int64_t)0);
            if (hipStreamSynchronize(h->stream) != hipSuccess ||
                hipMemcpy(dsts[c], tmp, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                hipFree(tmp);
                return fail(h, HS_E_HIP, "sink read-back failed");
            }
        }
        hipFree(tmp);
    }
    return cnt;
}

int64_t hs_engine_read_sinks(hs_engine *h, int64_t *counts, int64_t *t_ns, int64_t *created_ns, int64_t cap_total) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "stations not set");
    if (!counts) return fail(h, HS_E_INVALID, "counts is required");
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(h, HS_E_HIP, "device synchronisation failed");
    const size_t n = (size_t)h->cfg.n_lp;
    if (hipMemcpy(counts, h->X.received, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, HS_E_HIP, "memcpy");
    // exclusive offsets of each LP's run in the concatenated output; the transposition [cap][n_lp] -> per-LP runs
    // happens on the device (hs_gather_logs), then one bulk D2H per column
    const int64_t cap = h->L.cap;
    std::vector<int64_t> off(n);
    int64_t total = 0, maxc = 0;
    for (size_t i = 0; i < n; ++i) {
        const int64_t c = counts[i] > cap ? cap : counts[i];
        off[i] = total;
        total += c;
        maxc = c > maxc ? c : maxc;
    }
    if (total > cap_total) return fail(h, HS_E_INVALID, "output buffers too small for the sink records");
    if (total == 0 || (!t_ns && !created_ns)) return total;
    int64_t *d_off = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_off, n * 8) != hipSuccess || hipMalloc(&d_out, (size_t)total * 8) != hipSuccess) {
        if (d_off) hipFree(d_off);
        return fail(h, HS_E_HIP, "hipMalloc of the read-back staging buffers failed");
    }
    bool ok = hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess;
    const int64_t *cols[2] = {h->L.sink_t, h->L.sink_created};
    int64_t *dsts[2] = {t_ns, created_ns};
    for (int c = 0; c < 2 && ok; ++c) {
        if (!dsts[c]) continue;
        const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((maxc + 63) / 64));
        hipLaunchKernelGGL(hs_gather_logs, grid, dim3(256), 0, h->stream, cols[c], h->X.received, d_off, d_out, (int)n, cap, (int)h->L.lp_major);
        ok = hipStreamSynchronize(h->stream) == hipSuccess &&
             hipMemcpy(dsts[c], d_out, (size_t)total * 8, hipMemcpyDeviceToHost) == hipSuccess;
    }
    hipFree(d_off);
    hipFree(d_out);
    if (!ok) return fail(h, HS_E_HIP, "sink read-back failed");
    return total;
}

int64_t hs_engine_read_probe(hs_engine *h, int32_t lp, int64_t *t_ns, int64_t *values, int64_t cap) {
    return hs_engine_read_probe_slot(h, lp, 0, t_ns, values, cap);
}

int hs_engine_read_source_generated(hs_engine *h, int32_t slot, int64_t *out) {
    if (!h || !h->have_stations || !out) return fail(h, HS_E_STATE, "stations not set");
    if (slot < 0 || slot > kMaxXSrc) return fail(h, HS_E_INVALID, "source slot %d out of range", slot);
    const size_t n = (size_t)h->cfg.n_lp;
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (slot == 0) { HS_HIP(h, hipMemcpy(out, h->X.generated, n * 8, hipMemcpyDeviceToHost)); return HS_OK; }
    if (!h->any_xsrc) { for (size_t i = 0; i < n; ++i) out[i] = 0; return HS_OK; }
    HS_HIP(h, hipMemcpy(out, h->X.x_n + (size_t)(slot - 1) * n, n * 8, hipMemcpyDeviceToHost));
    return HS_OK;
}

int64_t hs_engine_read_probe_slot(hs_engine *h, int32_t lp, int32_t slot, int64_t *t_ns, int64_t *values, int64_t cap) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "stations not set");
    if (lp < 0 || lp >= h->cfg.n_lp) return fail(h, HS_E_INVALID, "LP index %d out of range", lp);
    if (slot < 0 || slot >= kMaxProbes) return fail(h, HS_E_INVALID, "probe slot %d out of range", slot);
    if (!h->any_probe || slot >= h->n_probe_slots) return 0;
    if (h) { const int rcf = results_final(h); if (rcf) return rcf; }
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(h, HS_E_HIP, "device synchronisation failed");
    int64_t cnt = 0;
    if (hipMemcpy(&cnt, h->X.p_n + (size_t)slot * h->cfg.n_lp + lp, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, HS_E_HIP, "memcpy");
    if (cnt > h->L.pcap) cnt = h->L.pcap;
    if (cnt > cap) cnt = cap;
    if (cnt > 0) {
        int64_t *tmp = nullptr;
        if (hipMalloc(&tmp, (size_t)cnt * 8) != hipSuccess) return fail(h, HS_E_HIP, "hipMalloc of the read-back staging buffer failed");
        const size_t so = (size_t)slot * (size_t)h->L.pcap * (size_t)h->cfg.n_lp;
        const int64_t *cols[2] = {h->L.probe_t + so, h->L.probe_v + so};
        int64_t *dsts[2] = {t_ns, values};
        for (int c = 0; c < 2; ++c) {
            if (!dsts[c]) continue;
            hipLaunchKernelGGL(hs_gather_one, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, cols[c], tmp,
                               h->cfg.n_lp, lp, cnt, (int64_t)0);
            if (hipStreamSynchronize(h->stream) != hipSuccess ||
                hipMemcpy(dsts[c], tmp, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                hipFree(tmp);
                return fail(h, HS_E_HIP, "probe read-back failed");
            }
        }
        hipFree(tmp);
    }
    return cnt;
}

void hs_engine_destroy(hs_engine *h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->ipc_opened) hipIpcCloseMemHandle(p);      // the peers' exchange buffers as mapped here
    if (h->live_stream) { hipStreamDestroy(h->live_stream); hipEventDestroy(h->live_ev); }
    for (void *p : h->allocs) hipFree(p);
    uncached_release(h);                       // (back to the process-wide pool, never to the runtime)
    if (h->ev_a) hipEventDestroy(h->ev_a);
    if (h->ev_b) hipEventDestroy(h->ev_b);
    if (h->ev_k0) hipEventDestroy(h->ev_k0);
    if (h->ev_k1) hipEventDestroy(h->ev_k1);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
}

int hs_debug_async_counters(hs_engine *h, unsigned long long out[4]) {
    if (!h || !out) return HS_E_INVALID;
    Totals t;
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost) != hipSuccess) return HS_E_HIP;
    for (int i = 0; i < 4; ++i) out[i] = t.dbg[i];
    return HS_OK;
}

int hs_debug_set_flags(hs_engine *h, int flags) {
    if (!h) return HS_E_INVALID;
    h->flags = flags;
    return HS_OK;
}

int hs_debug_draws(int32_t device, uint64_t seed, uint64_t sid, uint64_t k0, int64_t n, double rate, double *u,
                   double *e, int64_t *ns) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, HS_E_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (n <= 0 || !u || !e || !ns) return fail(nullptr, HS_E_INVALID, "hs_debug_draws: bad arguments");
    HS_HIP(nullptr, hipSetDevice(device));
    double *du = nullptr, *de = nullptr;
    int64_t *dn = nullptr;
    HS_HIP(nullptr, hipMalloc((void **)&du, (size_t)n * 8));
    HS_HIP(nullptr, hipMalloc((void **)&de, (size_t)n * 8));
    HS_HIP(nullptr, hipMalloc((void **)&dn, (size_t)n * 8));
    hipLaunchKernelGGL(hs_debug_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, seed, sid, k0, n, rate,
                       du, de, dn);
    HS_HIP(nullptr, hipGetLastError());
    HS_HIP(nullptr, hipMemcpy(u, du, (size_t)n * 8, hipMemcpyDeviceToHost));
    HS_HIP(nullptr, hipMemcpy(e, de, (size_t)n * 8, hipMemcpyDeviceToHost));
    HS_HIP(nullptr, hipMemcpy(ns, dn, (size_t)n * 8, hipMemcpyDeviceToHost));
    hipFree(du); hipFree(de); hipFree(dn);
    return HS_OK;
}

// The tick table of ONE stream (csrc/hs_tables.hpp) -- `lone` = 0: the cooperative kernel the engine uses; 1: the same chain on
// one lane with the sequential integrator (the device-side reference: tests/test_gpu_tables.py compares the two bit for bit).
// profile: {kind, p0, p1, p2, p3}; returns the number of entries written to out_times (<= cap), or < 0.
int64_t hs_debug_tick_table(int32_t device, const double *profile, int32_t poisson, uint64_t seed, uint64_t sid, int64_t start_ns,
                            int64_t horizon_ns, int64_t cap, int64_t lane_budget, int32_t lone, int64_t *out_times,
                            uint64_t *out_status) {
    if (!profile || !out_times || cap <= 0) return fail(nullptr, HS_E_INVALID, "hs_debug_tick_table: bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HS_E_NO_DEVICE, "no HIP device");
    TickRow r{};
    r.kind = (uint32_t)profile[0]; r.p0 = profile[1]; r.p1 = profile[2]; r.p2 = profile[3]; r.p3 = profile[4];
    r.poisson = poisson ? 1u : 0u; r.seed = seed; r.sid = sid; r.owner = 0;
    TickRow *drow = nullptr; int64_t *dt = nullptr, *dc = nullptr; unsigned long long *ds = nullptr;
    hipError_t e = hipMalloc((void **)&drow, sizeof r);
    if (e == hipSuccess) e = hipMalloc((void **)&dt, (size_t)cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&dc, 8);
    if (e == hipSuccess) e = hipMalloc((void **)&ds, 16);
    if (e == hipSuccess) e = hipMemcpy(drow, &r, sizeof r, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = tick_tables_launch(nullptr, drow, 1, start_ns, horizon_ns, cap, dt, dc, ds, lane_budget > 0 ? lane_budget : kDefaultLaneBudget, lone != 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    int64_t cnt = 0;
    unsigned long long st[2] = {0ull, 0ull};
    if (e == hipSuccess) e = hipMemcpy(&cnt, dc, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st, ds, 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_times, dt, (size_t)cap * 8, hipMemcpyDeviceToHost);
    (void)hipFree(drow); (void)hipFree(dt); (void)hipFree(dc); (void)hipFree(ds);
    if (e != hipSuccess) return fail(nullptr, HS_E_HIP, "hs_debug_tick_table: %s", hipGetErrorString(e));
    if (out_status) { out_status[0] = st[0]; out_status[1] = st[1]; }
    return cnt;
}

int hs_debug_const_div(int32_t device, double b, int64_t n, const double *a, double *q_fast, double *q_ieee,
                       double *q_ns) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, HS_E_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (n <= 0 || !a || !q_fast || !q_ieee || !q_ns) return fail(nullptr, HS_E_INVALID, "hs_debug_const_div: bad arguments");
    HS_HIP(nullptr, hipSetDevice(device));
    double *d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto &p : d) HS_HIP(nullptr, hipMalloc((void **)&p, (size_t)n * 8));
    HS_HIP(nullptr, hipMemcpy(d[0], a, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hs_debug_const_div_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, b, n, d[0], d[1],
                       d[2], d[3]);
    HS_HIP(nullptr, hipGetLastError());
    HS_HIP(nullptr, hipMemcpy(q_fast, d[1], (size_t)n * 8, hipMemcpyDeviceToHost));
    HS_HIP(nullptr, hipMemcpy(q_ieee, d[2], (size_t)n * 8, hipMemcpyDeviceToHost));
    HS_HIP(nullptr, hipMemcpy(q_ns, d[3], (size_t)n * 8, hipMemcpyDeviceToHost));
    for (auto &p : d) hipFree(p);
    return HS_OK;
}

}  // extern "C"
