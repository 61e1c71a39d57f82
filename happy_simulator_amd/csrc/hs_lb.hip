// hs_lb.hip -- load-balancer topologies on the GPU (the hs_lb_* part of include/hs_engine.h).
//
//     S x Source -> LoadBalancer(ConsistentHash) -> B x Server -> Sink(s)
//
// Replaces, for this entity set, `Simulation._execute_until` (happysimulator/core/simulation.py:449-505) together
// with LoadBalancer._forward_request / _handle_response (components/load_balancer/load_balancer.py:347-473) and
// ConsistentHash (components/load_balancer/strategies.py:336-433).  Every hop from a Source's tick to the backend's
// queue takes zero simulated time and the graph is feed-forward, so one run is a pipeline of passes, each with one
// logical process per lane (see the ABI comment in include/hs_engine.h):
//
//   hs_lbk_sources     S lanes   ticks -> (backend << TB | arrival ns, creation stamp) in [tick][source] logs
//   radix sort         HBM       by (backend, arrival ns)                      (hs_radix.hpp)
//   hs_lb_segments     n lanes   per-backend segment offsets in the sorted list
//   hs_lbk_backends<C> B lanes   Queue / Driver / Worker protocol over each backend's arrival list
//   radix sort         HBM       shared Sink: completions by completion ns
//   hs_lb_sink_finish  n lanes   ties of the merged Sink order + gather of created_at
//   hs_lb_finalize     1 block   election of the one event beyond end_ns (core/simulation.py:472) + totals
//
// Event accounting (reference-equivalent, SURVEY.md 3.2 + (f) N1): a tick at t <= end processes SourceEvent@Source,
// Request@LoadBalancer, Request@Server and `_lb_response`@LoadBalancer (the forwarded Event's completion hook fires as
// soon as the backend's non-generator enqueue handler returns: core/event.py:277-283) -- all at t -- then the
// backend's QUEUE_NOTIFY / QUEUE_POLL / QUEUE_DELIVER / Request@worker / ProcessContinuation / Request@Sink events as
// in hs_station.hpp.  Same-timestamp order inside a backend follows creation order exactly as there; an arrival's
// Request@Server is TWO generations below its SourceEvent (SourceEvent -> Request@LB -> Request@Server), which is
// modelled by the Q_PRE pseudo-events of the in-group FIFO.  There is no CPU fallback in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hs_engine.h"
#include "hs_ring.hpp"
#include "hs_device.hpp"
#include "hs_tables_api.hpp"
#include "hs_radix.hpp"

using namespace hs;

namespace {

constexpr int kLbBlock = 256;
constexpr int kLbQCap = 48;                  // in-group FIFO depth per backend (LDS)
constexpr uint32_t kStreamKey = 4;           // stream kind of a Source's client-id draws
constexpr uint64_t kCrtMask = (1ull << 56) - 1;

enum : uint32_t { LQ_NOTIFY = 2, LQ_POLL = 3, LQ_DELIVER = 4, LQ_PRE = 5, LQ_CONT = 6 };

struct LbTotals {
    unsigned long long ev[HS_EV_KINDS];
    unsigned long long completed, received;
    long long last_time;          // latest processed event time <= end
    long long final_time;         // Simulation._current_time after the run (the overshoot event's time)
    int qoverflow, bad_client;
    int probe_tie;                // a probe sample fell on the nanosecond of an event of its target (refused, never guessed)
    long long max_count;          // most Requests emitted by one source (rows of the arrival log in use)
    long long max_be;             // most Requests routed to one backend (rows of the wave-coalesced layout in use)
    int use_t;                    // this run uses the [row][backend] layout for the backend streams (max_be <= rows)
    int src_redo;                 // hs_lbk_sources_lean met what it leaves to hs_lbk_sources: the host repeats the run with that kernel
};

// The per-backend streams (arrival times in, service samples in, completion records out) in the layout the backend
// lanes want: [k][backend], so that the 64 lanes of a wavefront reading / appending their k-th elements touch 512
// contiguous bytes -- with dense per-backend segments every lane walks its own cache lines and the kernel is bound by the
// texture addresser.  The rows are sized at creation for 3x the mean load; a run whose busiest backend needs more falls
// back to the dense segments (device-side flag, same kernels, stride 1).
struct LbLayout {
    int64_t rows;
    uint64_t *tkey;               // [rows][B] sorted arrival keys of backend b, request k
    double *tsv;                  // [rows][B] service sample of that request (single-worker FIFO backends)
};

// a candidate for the one event beyond end_time.  The election's key is the reference's (time, _sort_index): of two events on one
// nanosecond the one created first; of two CREATED in one nanosecond the one fewer steps from the root of the group that created
// it, then the one whose root was created first (the heap is a FIFO inside a nanosecond; csrc/hs_station.hpp StationState
// lineage, tools/election_rules.py), then the construction order.
struct LbCand { long long t, t_created, rcrt; int depth, idx, valid, pad; double svc_s; };

struct LbSrc {                    // [S] each
    const uint8_t *kind; const double *rate; const int64_t *stop; const int64_t *n_clients; const uint64_t *base;
    const uint8_t *prof_kind;     // time-varying rate (Source.with_profile): 0 constant, 1 linear ramp, 2 spike; null = none anywhere
    const double *prof_p;         // [4][S]
    const int64_t *tab_times;     // tick tables of the time-varying Sources (hs_tables.hpp): tick d of Source s = tab_times[tab_row[s] * tab_cap + d]
    const int32_t *tab_row;       // [S] -1: constant rate
    int64_t tab_cap;
    int64_t *count;               // Requests emitted (ticks with a payload at t <= end)
    int64_t *generated;           // Source._generated_count
    LbCand *cand;                 // the pending SourceEvent beyond end
};

struct LbBe {                     // [B] each
    const int32_t *conc; const uint8_t *svc_kind; const double *svc_mean; const int64_t *qcap; const uint8_t *egress;
    const uint64_t *base;
    int64_t *accepted, *dropped, *completed, *rejected, *received, *depth;
    int32_t *active;
    double *total_service;
    LbCand *cand;                 // the earliest pending departure beyond end
};

__device__ __forceinline__ bool cand_before(const LbCand &a, const LbCand &b) {
    if (a.valid != b.valid) return a.valid > b.valid;
    if (!a.valid) return false;
    if (a.t != b.t) return a.t < b.t;
    if (a.t_created != b.t_created) return a.t_created < b.t_created;
    if (a.depth != b.depth) return a.depth < b.depth;
    if (a.rcrt != b.rcrt) return a.rcrt < b.rcrt;
    return a.idx < b.idx;
}
__device__ __forceinline__ LbCand lb_cand_none(int idx) {
    LbCand c;
    c.t = kInfNs; c.t_created = 0; c.rcrt = INT64_MIN; c.depth = 0; c.idx = idx; c.valid = 0; c.pad = 0; c.svc_s = 0.0;
    return c;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ LbCand cand_shfl_xor(const LbCand &c, int o) {
    LbCand d;
    d.t = __shfl_xor(c.t, o, 64); d.t_created = __shfl_xor(c.t_created, o, 64); d.rcrt = __shfl_xor(c.rcrt, o, 64);
    d.depth = __shfl_xor(c.depth, o, 64); d.pad = 0;
    d.idx = __shfl_xor(c.idx, o, 64); d.valid = __shfl_xor(c.valid, o, 64); d.svc_s = __shfl_xor(c.svc_s, o, 64);
    return d;
}
// earliest candidate of the workgroup -> out[blockIdx.x]   (every thread of the block must call this)
__device__ __forceinline__ void block_min_cand(LbCand c, LbCand *wc, LbCand *out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const LbCand d = cand_shfl_xor(c, o);
        if (cand_before(d, c)) c = d;
    }
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kLbBlock / 64; ++w) if (cand_before(wc[w], c)) c = wc[w];
        out[blockIdx.x] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// 1. Sources.  Source.handle_event (load/source.py:142-180) with a client-id request factory
//    (examples/visual/chash_example.py:69-88) and ConsistentHash.select as a table lookup.
// ---------------------------------------------------------------------------------------------
// The stream values of the Sources' ticks, produced by a kernel of their own (round 3): the inter-arrival increment E_d / rate and
// the backend of client-id draw d are pure functions of (seed, Source, d), so one thread per (pair of ticks, Source) computes them
// with the whole device -- hs_lbk_sources, one lane per Source and therefore half the SIMDs idle, keeps only the serial ns
// recursion and the log appends.  [tick][source] layout: the consumers' reads are coalesced.  `n_pre` ticks per Source are
// produced (mean + 5 sigma of the busiest Source); a Source that needs more computes them itself.
// Speculated arrival steps (round 4).  A tick is A' = trunc((A / 1e9 + inc) * 1e9) with three binary64 roundings: ten DEPENDENT fp64
// instructions, the serial chain of hs_lbk_sources.  With F = inc * 1e9:  the rounded product z the truncation sees satisfies
// |z - (A + F)| <= 3u (A + F) + O(u^2), u = 2^-53 (A / 1e9 is correctly rounded -- hs_device.hpp seconds_from_ns_d --, then one
// rounded sum and one rounded product), and F itself is known to u F from the rounded product RN(inc * 1e9).  So for horizons below
// 2^40 ns (18 minutes) every error is below 4e-4 ns, and whenever frac(RN(inc * 1e9)) lies in [2^-10, 1 - 2^-10] the tick is EXACTLY
// A' = A + floor(RN(inc * 1e9)): one dependent add.  The draws kernel stores that whole-nanosecond step; the (one in 500) increments
// too close to a whole number keep `inc` itself, marked by the sign bit, and take the ten-instruction step.  margin == 0: no
// speculation (time-varying Sources, horizons of 2^40 ns and more): the value is `inc`.
// Where the stream values of (tick d, Source s) live in hs_lb_source_draws' output: tiled so that what ONE wavefront of
// hs_lbk_sources loads for a chunk of 16 ticks (16 x 64 values) is contiguous -- with a plain [tick][source] array those 32 loads
// went to 32 rows 256 KB apart (a page each), and the chunk's load latency, not the ticks, set the kernel's time.
__device__ __forceinline__ size_t lb_draw_index(uint64_t d, int s, int S) {
    const size_t waves = ((size_t)S + 63) >> 6;
    return ((((size_t)(d >> 4) * waves + ((size_t)s >> 6)) << 4) + (size_t)(d & 15)) * 64 + ((size_t)s & 63);
}
__device__ __forceinline__ double lb_step_encode(double inc, double margin) {
    if (!(margin > 0.0)) return inc;
    const double F = __dmul_rn(inc, 1e9), fl = __builtin_floor(F), frac = __dsub_rn(F, fl);
    // (F < 2^39: with A < 2^40 the three roundings of the reference's step stay below 3u(A + F) + uF < 6.2e-4 ns < the 2^-10 margin --
    //  the old guard F < 4e15 let a Poisson Source slower than ~0.017/s step by > 2^41 ns, where that bound fails)
    const bool safe = frac >= margin && frac <= __dsub_rn(1.0, margin) && F < 549755813888.0;
    return safe ? fl : __longlong_as_double((long long)((uint64_t)__double_as_longlong(inc) | 0x8000000000000000ull));
}
// the next tick after the one at `arr_d` (whole ns as a double) for an encoded step
__device__ __forceinline__ double lb_step_apply(double arr_d, double v) {
    if (__double_as_longlong(v) >= 0) return __dadd_rn(arr_d, v);                                  // a whole-nanosecond step: exact
    const double inc = __longlong_as_double((long long)((uint64_t)__double_as_longlong(v) & 0x7fffffffffffffffull));
    return ns_from_seconds_d(__dadd_rn(seconds_from_ns_d(arr_d), inc));
}

__global__ void __launch_bounds__(256) hs_lb_source_draws(LbSrc P, int S, uint64_t seed, const int32_t *__restrict__ client_be,
                                                          int64_t n_table, int64_t n_pre, double *__restrict__ dinc,
                                                          int32_t *__restrict__ dbe, double margin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = i / S;
    const int s = (int)(i - pair * S);
    if (2 * pair >= n_pre) return;
    const uint32_t kind = P.kind[s];
    const double rate = P.rate[s];
    const double nclients = (double)P.n_clients[s];
    const uint64_t sa = stream_id(P.base[s], kStreamArrival), sk = stream_id(P.base[s], kStreamKey);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    double inc0, inc1;
    if (kind == HS_SRC_POISSON) {
        const U4 o = philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)sa, (uint32_t)(sa >> 32), k0, k1);
        inc0 = __ddiv_rn(exp1_from_uniform(res53(o.x, o.y)), rate);
        inc1 = __ddiv_rn(exp1_from_uniform(res53(o.z, o.w)), rate);
    } else { inc0 = __ddiv_rn(1.0, rate); inc1 = inc0; }
    const U4 q = philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)sk, (uint32_t)(sk >> 32), k0, k1);
    const int64_t c0 = __double2ll_rz(__dmul_rn(res53(q.x, q.y), nclients));
    const int64_t c1 = __double2ll_rz(__dmul_rn(res53(q.z, q.w), nclients));
    const size_t o0 = lb_draw_index((uint64_t)(2 * pair), s, S), o1 = lb_draw_index((uint64_t)(2 * pair + 1), s, S);
    dinc[o0] = lb_step_encode(inc0, margin); dbe[o0] = (c0 >= 0 && c0 < n_table) ? client_be[c0] : -1;
    if (2 * pair + 1 < n_pre) { dinc[o1] = lb_step_encode(inc1, margin); dbe[o1] = (c1 >= 0 && c1 < n_table) ? client_be[c1] : -1; }
}

// Round 4: the Sources' ticks with ~20 instructions each instead of ~110.  hs_lbk_sources carries, per tick, everything the
// reference's rare cases need (two ticks on one nanosecond and their lineage depth, stop_after, time travel, a full log, an invalid
// client id, the tick beyond end_time) as straight-line predicates -- on a LONE wavefront per SIMD every instruction costs ~8 cycles,
// and the kernel took 175 us at the configs[4] size whatever was done to its memory accesses.  This kernel runs the COMMON run only:
// constant-rate Sources on the exact-binary64 path with speculated whole-nanosecond steps (lb_step_encode), no stop_after, every
// value pre-drawn; per tick one dependent add, the int64 of the sum, two packed stores -- whole chunks of 16 ticks, also past
// end_time (the rows exist; TickValid reads LbSrc::count) -- and it raises LbTotals::src_redo where a Source leaves that run (two
// ticks on one nanosecond, more ticks than were pre-drawn, a client id outside the table); the host then repeats the run with hs_lbk_sources (hs_lb_run).  The tick
// beyond end_time and its lineage are read back from the Source's own log rows at the end.
constexpr int kLeanBlock = kLbBlock;            // (the same grid as hs_lbk_sources: LbSrc::cand has one slot per workgroup)
__global__ void __launch_bounds__(kLeanBlock) hs_lbk_sources_lean(LbSrc P, int S, int64_t start_ns, int64_t end_ns,
                                                                 uint64_t *__restrict__ keys, uint64_t *__restrict__ vals,
                                                                 int64_t rows, int tb, LbTotals *tot, const double *__restrict__ dinc,
                                                                 const int32_t *__restrict__ dbe, int lanes, int give_up) {
    constexpr int kChunk = 16;
    __shared__ LbCand wc[kLeanBlock / 64];
    __shared__ double s_inc[kChunk][kLeanBlock];
    __shared__ int32_t s_be[kChunk][kLeanBlock];
    const int s = ((blockIdx.x * kLeanBlock + threadIdx.x) >> 6) * lanes + (threadIdx.x & 63);
    const bool live = (threadIdx.x & 63) < lanes && s < S;
    uint32_t n = 0;                               // ticks <= end_ns
    bool redo = give_up != 0;                     // (debug flag 1024: the repeat path under test)
    LbCand c = lb_cand_none(s);
    int64_t last = INT64_MIN;
    if (live) {
        const double endd = (double)end_ns;
        double A = (double)start_ns;              // the tick before the next one (whole ns, exact)
        int64_t prev_i = start_ns;
        uint32_t off = (uint32_t)s * 8u;          // byte offset of the next row's slot (rows * S * 8 < 2^32: the host checks it)
        const uint32_t row_bytes = (uint32_t)S * 8u;
        int64_t done_rows = 0;
        double pinc[kChunk];
        int32_t pbe[kChunk];
        auto prefetch = [&](uint64_t d0) {
            if ((int64_t)(d0 + kChunk) > rows) return;
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                const size_t o = lb_draw_index(d0 + j, s, S);
                pinc[j] = dinc[o]; pbe[j] = dbe[o];
            }
        };
        prefetch(0);
        for (int64_t d0 = 0; d0 + kChunk <= rows; d0 += kChunk) {
            if (!__any(A <= endd)) break;         // (wavefront-uniform: every Source of the wavefront has its tick beyond end_ns)
#pragma unroll
            for (int j = 0; j < kChunk; ++j) { s_inc[j][threadIdx.x] = pinc[j]; s_be[j][threadIdx.x] = pbe[j]; }
            prefetch((uint64_t)d0 + kChunk);
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                const double v = s_inc[j][threadIdx.x];
                const int32_t be_j = s_be[j][threadIdx.x];
                double An = __dadd_rn(A, v);                                         // a whole-nanosecond step: exact
                if (__builtin_expect(__double_as_longlong(v) < 0, 0)) An = lb_step_apply(A, v);   // (one increment in ~500)
                redo = redo || An == A || (be_j < 0 && An <= endd);                  // two ticks on one nanosecond; a client id outside the table
                const int64_t a_i = i64_from_whole_d(An);
                *reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(keys) + off) = ((uint64_t)(uint32_t)be_j << tb) | (uint64_t)a_i;
                *reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(vals) + off) = (uint64_t)prev_i & kCrtMask;
                n += An <= endd ? 1u : 0u;
                prev_i = a_i; A = An; off += row_bytes;
            }
            done_rows = d0 + kChunk;
        }
        if (A <= endd) redo = true;               // more ticks than rows were pre-drawn
        if (!redo) {
            // tick n is the first beyond end_ns: row n + 1 holds it as "the tick before" (if that row was written: else it is A)
            __threadfence();
            auto row_val = [&](int64_t r) {
                return (int64_t)(__hip_atomic_load(&vals[(size_t)r * (size_t)S + (size_t)s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kCrtMask);
            };
            const int64_t a_n = (int64_t)n + 1 < done_rows ? row_val((int64_t)n + 1) : i64_from_whole_d(A);
            const int64_t a_n1 = n >= 1 ? row_val((int64_t)n) : start_ns;                       // the last tick <= end_ns (or the start)
            c.t = a_n; c.t_created = a_n1; c.valid = 1;
            c.depth = n >= 1 ? 1 : 0;
            c.rcrt = n >= 2 ? row_val((int64_t)n - 1) : n == 1 ? start_ns : INT64_MIN;
            if (n >= 1) last = a_n1;
        }
        P.count[s] = (int64_t)n;
        P.generated[s] = n;
    }
    block_min_cand(c, wc, P.cand);
    long long mc = live ? (long long)n : 0ll, ml = live ? (long long)last : INT64_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_xor(mc, o, 64), b2 = __shfl_xor(ml, o, 64);
        mc = a > mc ? a : mc; ml = b2 > ml ? b2 : ml;
    }
    const uint32_t st = wave_sum<uint32_t>(live ? n : 0u);
    if ((threadIdx.x & 63) == 0) {
        if (mc) atomicMax(&tot->max_count, mc);
        if (st) { atomicAdd(&tot->ev[HS_EV_SOURCE], (unsigned long long)st); atomicAdd(&tot->ev[HS_EV_LB], (unsigned long long)st);
                  atomicAdd(&tot->ev[HS_EV_LB_RESP], (unsigned long long)st); }
        if (ml != INT64_MIN) atomicMax(&tot->last_time, ml);
    }
    if (__any(redo) && (threadIdx.x & 63) == 0) atomicOr(&tot->src_redo, 2);
}

// PF: some Source has a time-varying profile -- its next arrival is the reference's numerical inversion (hs_profile.hpp), a
// separate instantiation so that the common one carries no scratch frame.
// F64: every time of the run is a whole number of ns below 2^51 (hs_lb::f64_times): the recursion runs on binary64 integers.
// Round 4: a chunk's stream values go through LDS (gfx9 counts loads and stores with ONE counter, and with the chunk in registers the
// compiler put an `s_waitcnt vmcnt(0)` in front of every tick's first use of a loaded value, i.e. behind the log stores of the tick
// before it; LDS reads have their own counter), the next chunk is in flight meanwhile, the common tick is straight-line code on a
// speculated whole-nanosecond step (lb_step_encode) and the totals take one atomic per wavefront.  Measured honestly: the kernel
// stayed at 170-180 us through all of it -- SQ counters say ~110 instructions per tick at ~8.4 cycles each on a LONE wavefront per
// SIMD (512 wavefronts of 64 Sources; the chain of a Source is serial, so more wavefronts do not shorten it).  What would: K lanes
// per Source on the whole-ns steps (an exact integer prefix sum, the marked increments resolved one by one) -- not built.
template <bool PF, bool F64>
__global__ void __launch_bounds__(kLbBlock) hs_lbk_sources(LbSrc P, int S, uint64_t seed, int64_t start_ns, int64_t end_ns,
                                                          const int32_t *__restrict__ client_be, int64_t n_table,
                                                          uint64_t *__restrict__ keys, uint64_t *__restrict__ vals,
                                                          int64_t cap, int tb, LbTotals *tot, const double *__restrict__ dinc,
                                                          const int32_t *__restrict__ dbe, int64_t n_pre, int lanes, double margin) {
    constexpr int kChunk = 16;
    __shared__ LbCand wc[kLbBlock / 64];
    __shared__ double s_inc[kChunk][kLbBlock];
    __shared__ int32_t s_be[kChunk][kLbBlock];
    const int f64_times = F64 ? 1 : 0;
    const int s = ((blockIdx.x * kLbBlock + threadIdx.x) >> 6) * lanes + (threadIdx.x & 63);     // `lanes` Sources per wavefront (lb_lanes())
    const bool live = (threadIdx.x & 63) < lanes && s < S;
    uint32_t n_tick = 0, n_req = 0;
    int bad = 0, over = 0;
    int64_t last = INT64_MIN;
    LbCand c = lb_cand_none(s);
    if (live) {
        const uint32_t kind = P.kind[s];
        const double rate = P.rate[s];
        const int64_t stop = P.stop[s];
        const double nclients = (double)P.n_clients[s];
        const uint64_t sa = stream_id(P.base[s], kStreamArrival), sk = stream_id(P.base[s], kStreamKey);
        const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        const double inc_const = __ddiv_rn(1.0, rate);       // constant source: target area 1.0 (providers/constant_arrival.py:23)
        const int64_t *tab = nullptr;                                 // a time-varying Source: its ticks come from the tick table
        if constexpr (PF) { if (P.prof_kind[s] != kProfConstant) tab = P.tab_times + (size_t)P.tab_row[s] * (size_t)P.tab_cap; }
        const bool timevarying = PF && tab != nullptr;
        // Tick d happens at A_d = from_seconds(to_seconds(A_{d-1}) + E_d / rate)  (load/arrival_time_provider.py:72-82;
        // A_{-1} = start: Source.start at Simulation.__init__, load/source.py:120-140) and, while the provider still
        // returns Requests, draws client id number d.  Both streams are therefore indexed by the tick number: the
        // expensive part (Philox, hs_log, the division, the client -> backend lookup) is produced eight ticks at a time
        // in straight-line code -- independent chains the SIMD can overlap -- and only the ns recursion is serial.
        // (f64_times: every time of the run is a whole number of ns below 2^51, so the recursion runs on binary64 integers --
        //  hs_device.hpp ns_from_seconds_d: 8 dependent fp64 instructions per tick instead of ~60 with the i64 <-> f64 conversion
        //  sequences; the int64 the logs need is converted off the chain.  Round 3: the chain was 0.67 us per tick per wavefront.)
        double arr_d = (double)start_ns;
        int64_t arr_time = start_ns, t_prev = start_ns;
        int64_t root_crt = start_ns;         // creation time of the SourceEvent that heads the current same-ns chain
        uint32_t depth = 0;
        int64_t rc_a2 = INT64_MIN;           // lineage of the tick a2 itself: the root of the group that created it was created at
        uint32_t dp_a2 = 0;                  // ... rc_a2, dp_a2 steps before it (0: constructed before run())
        int64_t A = kInfNs;
        size_t kpos = (size_t)s;             // slot of the next Request in the [tick][source] logs: n_req * S + s
        bool done = false, dead = false;
        // (a chunk's values are loaded when nothing else is in flight, with one explicit vmcnt(0): see run_request_order)
        // the chunk after the one being processed is in flight (registers) while the ticks of this one run out of LDS: a chunk's
        // loads take 2-3 us, as long as its sixteen ticks
        double pinc[kChunk];
        int32_t pbe[kChunk];
        auto prefetch = [&](uint64_t d0) {
            if ((int64_t)(d0 + kChunk) > n_pre) return;
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                const size_t o = lb_draw_index(d0 + j, s, S);
                pinc[j] = dinc[o]; pbe[j] = dbe[o];
            }
        };
        prefetch(0);
        for (uint64_t d0 = 0; !done; d0 += kChunk) {
            if ((int64_t)(d0 + kChunk) <= n_pre) {                       // the values hs_lb_source_draws produced
#pragma unroll
                for (int j = 0; j < kChunk; ++j) { s_inc[j][threadIdx.x] = pinc[j]; s_be[j][threadIdx.x] = pbe[j]; }
                prefetch(d0 + kChunk);
            } else {
            double inc[kChunk];
            int32_t be[kChunk];
#pragma unroll
            for (int j = 0; j < kChunk; j += 2) {
                const uint64_t blk = (d0 + j) >> 1;
                if (kind == HS_SRC_POISSON) {
                    const U4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sa, (uint32_t)(sa >> 32), k0, k1);
                    const double e0 = exp1_from_uniform(res53(o.x, o.y)), e1 = exp1_from_uniform(res53(o.z, o.w));
                    inc[j] = __ddiv_rn(e0, rate);
                    inc[j + 1] = __ddiv_rn(e1, rate);
                } else { inc[j] = inc_const; inc[j + 1] = inc[j]; }
                if constexpr (!PF && F64) { inc[j] = lb_step_encode(inc[j], margin); inc[j + 1] = lb_step_encode(inc[j + 1], margin); }
                const U4 q = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sk, (uint32_t)(sk >> 32), k0, k1);
                const int64_t c0 = __double2ll_rz(__dmul_rn(res53(q.x, q.y), nclients));
                const int64_t c1 = __double2ll_rz(__dmul_rn(res53(q.z, q.w), nclients));
                const bool ok0 = c0 >= 0 && c0 < n_table, ok1 = c1 >= 0 && c1 < n_table;
                be[j] = ok0 ? client_be[c0] : -1;
                be[j + 1] = ok1 ? client_be[c1] : -1;
            }
#pragma unroll
            for (int j = 0; j < kChunk; ++j) { s_inc[j][threadIdx.x] = inc[j]; s_be[j][threadIdx.x] = be[j]; }
            }
            // (each lane reads back only what it wrote itself: no barrier)
#pragma unroll 4
            for (int j = 0; j < kChunk; ++j) {
                if (done) continue;
                const uint64_t d = d0 + j;
                const double inc_j = s_inc[j][threadIdx.x];
                const int32_t be_j = s_be[j][threadIdx.x];
                if constexpr (!PF && F64) {
                    // The common tick as straight-line code: later than the tick before it, not beyond end_ns, a Request with a valid
                    // backend and room in the log.  When every lane of the wavefront that still ticks is there (a ballot), the tick is
                    // the ns recursion, two packed stores and four counters -- no branches; anything else (the first tick, two ticks on
                    // one nanosecond, time travel, the tick beyond end_ns, stop_after) takes the general code below, unchanged.
                    const double arr_n = __dadd_rn(arr_d, inc_j);                  // (a whole-ns step: exact; a marked increment fails `plain`)
                    const int64_t a2f = i64_from_whole_d(arr_n);
                    const bool plain = __double_as_longlong(inc_j) >= 0 && margin > 0.0 && d > 0 && a2f > t_prev && a2f <= end_ns && !(stop >= 0 && a2f > stop) && be_j >= 0 && (int64_t)n_req < cap;
                    if (__ballot(!plain) == 0ull) {
                        arr_d = arr_n; arr_time = a2f;
                        rc_a2 = root_crt; dp_a2 = depth + 1;
                        root_crt = t_prev; depth = 0;
                        t_prev = a2f; ++n_tick; last = a2f;
                        keys[kpos] = ((uint64_t)be_j << tb) | (uint64_t)a2f;
                        vals[kpos] = (uint64_t)root_crt & kCrtMask;
                        ++n_req; kpos += (size_t)S;
                        continue;
                    }
                }
                int64_t a2;
                if constexpr (PF) {
                    a2 = timevarying ? (d < (uint64_t)P.tab_cap ? tab[d] : (over = 1, kInfNs))   // load/arrival_time_provider.py:84-144
                                     : ns_from_seconds(__dadd_rn(seconds_from_ns(arr_time), inc_j));
                    if (a2 == kInfNs) { done = true; dead = true; continue; }        // the rate is zero from here on: the Source ends
                } else if (f64_times) {
                    arr_d = margin > 0.0 ? lb_step_apply(arr_d, inc_j) : ns_from_seconds_d(__dadd_rn(seconds_from_ns_d(arr_d), inc_j));
                    a2 = i64_from_whole_d(arr_d);
                } else a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(arr_time), inc_j));
                arr_time = a2;
                if (d > 0) {
                    rc_a2 = root_crt; dp_a2 = depth + 1;                  // created by the tick at t_prev: one step below it in ITS group
                    if (a2 == t_prev) ++depth;                            // next tick on the same nanosecond: a descendant
                    else if (a2 < t_prev) { done = true; dead = true; continue; }   // popped later as "time travel" and dropped (simulation.py:480-489)
                    else { root_crt = t_prev; depth = 0; }
                }
                if (a2 > end_ns) { A = a2; done = true; continue; }       // the pending SourceEvent beyond end
                const int64_t t = a2;
                t_prev = t;
                ++n_tick;
                last = t;
                if (!(stop >= 0 && t > stop)) {                           // the provider returns one Request
                    if (be_j < 0) bad = 1;
                    if ((int64_t)n_req < cap) {
                        keys[kpos] = ((uint64_t)(be_j < 0 ? 0 : be_j) << tb) | (uint64_t)t;
                        vals[kpos] = ((uint64_t)(depth > 30 ? 30 : depth) << 56) | ((uint64_t)root_crt & kCrtMask);
                    } else over = 1;
                    ++n_req; kpos += (size_t)S;
                }
            }
        }
        P.count[s] = (int64_t)(n_req < (uint32_t)cap ? n_req : (uint32_t)cap);
        P.generated[s] = n_tick;
        c.t = A; c.t_created = root_crt; c.valid = (!dead && A != kInfNs) ? 1 : 0;
        c.depth = (int)(dp_a2 > 255u ? 255u : dp_a2); c.rcrt = rc_a2;
    }
    block_min_cand(c, wc, P.cand);
    // (one atomic per WAVEFRONT instead of two per lane on the same two words)
    long long mc = (live && n_req) ? (long long)(n_req < (uint32_t)cap ? n_req : (uint32_t)cap) : 0ll, ml = live ? (long long)last : INT64_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_xor(mc, o, 64), b2 = __shfl_xor(ml, o, 64);
        mc = a > mc ? a : mc; ml = b2 > ml ? b2 : ml;
    }
    const uint32_t st = wave_sum<uint32_t>(n_tick), sr = wave_sum<uint32_t>(n_req);
    if ((threadIdx.x & 63) == 0) {
        if (mc) atomicMax(&tot->max_count, mc);
        if (st) atomicAdd(&tot->ev[HS_EV_SOURCE], (unsigned long long)st);
        if (sr) { atomicAdd(&tot->ev[HS_EV_LB], (unsigned long long)sr); atomicAdd(&tot->ev[HS_EV_LB_RESP], (unsigned long long)sr); }
        if (ml != INT64_MIN) atomicMax(&tot->last_time, ml);
    }
    if (bad) atomicOr(&tot->bad_client, 1);
    if (over) atomicOr(&tot->bad_client, 2);
}

// validity of slot i = (tick k, source s) of the [cap][S] arrival logs
struct TickValid {
    const int64_t *count; int S; bool fits32;
    __device__ __forceinline__ bool operator()(int64_t i) const {
        if (fits32) {                              // 32-bit division: a fraction of the 64-bit sequence
            const uint32_t k = (uint32_t)i / (uint32_t)S;
            return (int64_t)k < count[(uint32_t)i - k * (uint32_t)S];
        }
        const int64_t k = i / S;
        return k < count[i - k * S];
    }
};
// the arrival log is [tick][source]: only the first max_count rows hold anything
__global__ void hs_lb_rows(const LbTotals *tot, int S, int64_t *n_slots) { *n_slots = (int64_t)tot->max_count * S; }

// value carried through the Sink merge: created_at and the slot in one word when they fit (no gather afterwards)
struct PackCreatedSlot {
    const int64_t *created; int slot_bits;
    __device__ __forceinline__ uint64_t operator()(int64_t i) const { return ((uint64_t)created[i] << slot_bits) | (uint64_t)i; }
};
struct NoVal { __device__ __forceinline__ uint64_t operator()(int64_t) const { return 0ull; } };
struct SlotVal { __device__ __forceinline__ uint64_t operator()(int64_t i) const { return (uint64_t)i; } };

// ---------------------------------------------------------------------------------------------
// 2b. Segment offsets: backend b's arrivals are sorted[off[b] .. off[b+1])
// ---------------------------------------------------------------------------------------------
// The sort ran on key bits [g, ...) only (whole 8-bit passes saved): elements whose keys agree above bit g are still in
// input order.  Such runs are short (the host picks g so that a bucket holds <= 1 element on average); every element
// finds its place inside its run by counting the run's smaller (key, position) pairs -- stable -- and the list is
// written out in full-key order to the other ping-pong buffer.  g <= tb, so a run never spans two backends.
// (round 4: a workgroup stages its 1 024 keys + a halo in LDS, so the walk along a run reads LDS instead of one dependent global
//  load per step: 163 us -> see profiles/; a run that reaches past the halo continues in global memory)
constexpr int kSegTile = 1024, kSegHalo = 32;
__global__ void __launch_bounds__(256) hs_lb_segments(const uint64_t *__restrict__ skey, const uint64_t *__restrict__ sval,
                                                     uint64_t *__restrict__ fkey, uint64_t *__restrict__ fval, const int64_t *n_ptr, int tb,
                                                     int g, int B, int64_t *__restrict__ off) {
    __shared__ uint64_t lk[kSegTile + 2 * kSegHalo];
    const int64_t n = *n_ptr;
    const int64_t base = (int64_t)blockIdx.x * kSegTile;
    if (base > n) return;
    const int tid = threadIdx.x;
    for (int q = tid; q < kSegTile + 2 * kSegHalo; q += 256) {
        const int64_t i = base - kSegHalo + q;
        lk[q] = (i >= 0 && i < n) ? skey[i] : 0ull;
    }
    __syncthreads();
    const int64_t lds_lo = base - kSegHalo, lds_hi = base + kSegTile + kSegHalo;      // global indices [lds_lo, lds_hi) are in LDS
    auto key_at = [&](int64_t i) -> uint64_t { return (i >= lds_lo && i < lds_hi) ? lk[i - lds_lo] : skey[i]; };
#pragma unroll
    for (int r = 0; r < kSegTile / 256; ++r) {
        const int64_t i = base + tid + 256 * r;
        if (i > n) continue;
        const uint64_t k_prev = i == 0 ? 0ull : key_at(i - 1);
        const uint64_t k_here = i == n ? 0ull : key_at(i);
        const int64_t b_prev = i == 0 ? -1 : (int64_t)(k_prev >> tb);
        const int64_t b_here = i == n ? (int64_t)B : (int64_t)(k_here >> tb);
        for (int64_t b = b_prev + 1; b <= b_here; ++b) off[b] = i;
        if (i == n) continue;
        int64_t pos = i;
        if (g > 0) {
            const uint64_t hi = k_here >> g;
            int64_t lo = i, rank = 0;
            while (lo > 0) {                                     // elements of the run before this one
                const uint64_t k = key_at(lo - 1);
                if ((k >> g) != hi) break;
                rank += (k <= k_here) ? 1 : 0;                   // equal keys keep their input order
                --lo;
            }
            for (int64_t j = i + 1; j < n; ++j) {                // ... and after it
                const uint64_t k = key_at(j);
                if ((k >> g) != hi) break;
                rank += (k < k_here) ? 1 : 0;
            }
            pos = lo + rank;
        }
        fkey[pos] = k_here;
        fval[pos] = sval[i];
    }
}

// RoundRobin (strategies.py:50-73): the k-th Request the LoadBalancer processes goes to backend k mod B.  Input: ALL Requests sorted
// by arrival ns on key bits [g, tb) (backend column 0); inside a run of equal high bits every Request finds its place by counting
// the run's Requests that the LoadBalancer processes before it -- earlier ns; on one ns the one whose SourceEvent chain was created
// first (the stamp the per-backend order uses as well, LbBackend::order_arrival_run), then input order = (tick, Source) --, and
// that place IS the RoundRobin index: the Request leaves with key (place mod B) << tb | ns for the usual (backend, ns) sort.
__global__ void __launch_bounds__(256) hs_lb_rr_assign(const uint64_t *__restrict__ skey, const uint64_t *__restrict__ sval,
                                                      uint64_t *__restrict__ fkey, uint64_t *__restrict__ fval, const int64_t *n_ptr, int tb,
                                                      int g, int B) {
    __shared__ uint64_t lk[kSegTile + 2 * kSegHalo];
    const int64_t n = *n_ptr;
    const int64_t base = (int64_t)blockIdx.x * kSegTile;
    if (base >= n) return;
    const int tid = threadIdx.x;
    for (int q = tid; q < kSegTile + 2 * kSegHalo; q += 256) {
        const int64_t i = base - kSegHalo + q;
        lk[q] = (i >= 0 && i < n) ? skey[i] : 0ull;
    }
    __syncthreads();
    const int64_t lds_lo = base - kSegHalo, lds_hi = base + kSegTile + kSegHalo;
    auto key_at = [&](int64_t i) -> uint64_t { return (i >= lds_lo && i < lds_hi) ? lk[i - lds_lo] : skey[i]; };
#pragma unroll
    for (int r = 0; r < kSegTile / 256; ++r) {
        const int64_t i = base + tid + 256 * r;
        if (i >= n) continue;
        const uint64_t k = key_at(i), v = sval[i], c_me = v & kCrtMask;
        const uint64_t hi = k >> g;
        int64_t lo = i, rank = 0;
        auto before = [&](int64_t j) {
            const uint64_t kj = key_at(j);
            if (kj != k) return kj < k;
            const uint64_t cj = sval[j] & kCrtMask;
            return cj < c_me || (cj == c_me && j < i);
        };
        while (lo > 0 && (key_at(lo - 1) >> g) == hi) { rank += before(lo - 1) ? 1 : 0; --lo; }
        for (int64_t j = i + 1; j < n && (key_at(j) >> g) == hi; ++j) rank += before(j) ? 1 : 0;
        const int64_t pos = lo + rank;
        const uint64_t tmask = tb >= 64 ? ~0ull : ((1ull << tb) - 1);
        fkey[pos] = ((uint64_t)(pos % B) << tb) | (k & tmask);
        fval[pos] = v;
    }
}

__global__ void hs_lb_maxcount(const int64_t *__restrict__ off, int B, LbTotals *tot) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    long long c = b < B ? (long long)(off[b + 1] - off[b]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long d = __shfl_xor(c, o, 64); c = d > c ? d : c; }
    if ((threadIdx.x & 63) == 0 && c) atomicMax(&tot->max_be, c);
}
// decide the layout of this run; n_merge = slots the Sink merge has to look at
__global__ void hs_lb_layout(LbTotals *tot, LbLayout LY, int B, const int64_t *n_arr, int64_t *n_merge, int force_dense) {
    const int t = (!force_dense && LY.rows > 0 && tot->max_be <= LY.rows) ? 1 : 0;
    tot->use_t = t;
    *n_merge = t ? (int64_t)tot->max_be * B : *n_arr;
}
// dense sorted segments -> [k][backend]: one workgroup per 64 backends, 64 x 64 tiles through LDS (coalesced both ways)
__global__ void __launch_bounds__(256) hs_lb_transpose(const uint64_t *__restrict__ skey, const int64_t *__restrict__ off, int B,
                                                       LbLayout LY, const LbTotals *tot) {
    if (!tot->use_t) return;
    __shared__ uint64_t tile[64][65];
    __shared__ int64_t seg_off[65];
    __shared__ int64_t max_len;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 64;
    if (threadIdx.x < 65) seg_off[threadIdx.x] = off[(b0 + (int)threadIdx.x) < B ? b0 + (int)threadIdx.x : B];
    if (threadIdx.x == 0) max_len = 0;
    __syncthreads();
    if (threadIdx.x < 64) atomicMax((long long *)&max_len, (long long)(seg_off[threadIdx.x + 1] - seg_off[threadIdx.x]));
    __syncthreads();
    const int64_t ml = max_len;
    for (int64_t k0 = 0; k0 < ml; k0 += 64) {
        for (int r = ty; r < 64; r += 4) {                       // row r = backend b0 + r: 64 consecutive elements
            const int64_t len = seg_off[r + 1] - seg_off[r];
            tile[r][tx] = (k0 + tx < len) ? skey[seg_off[r] + k0 + tx] : 0ull;
        }
        __syncthreads();
        for (int kk = ty; kk < 64; kk += 4) {                    // row kk of the output: 64 consecutive backends
            const int64_t k = k0 + kk;
            const int64_t len = seg_off[tx + 1] - seg_off[tx];
            if (k < len && b0 + tx < B) LY.tkey[(size_t)k * B + b0 + tx] = tile[tx][kk];
        }
        __syncthreads();
    }
}
__global__ void hs_lb_gather_strided(const int64_t *__restrict__ src, int64_t stride, int64_t cnt, int64_t *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) out[k] = src[(size_t)k * stride];
}

// keys-only variant of the run fix-up above (latency statistics)
__global__ void hs_lb_fix_runs(uint64_t *__restrict__ keys, const int64_t *n_ptr, int g) {
    const int64_t n = *n_ptr;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t hi = keys[i] >> g;
    if (i > 0 && (keys[i - 1] >> g) == hi) return;
    int64_t len = 1;
    while (i + len < n && (keys[i + len] >> g) == hi) ++len;
    for (int64_t a = 1; a < len; ++a) {
        const uint64_t k = keys[i + a];
        int64_t j = a;
        while (j > 0 && keys[i + j - 1] > k) { keys[i + j] = keys[i + j - 1]; --j; }
        keys[i + j] = k;
    }
}

// ---------------------------------------------------------------------------------------------
// 2c. Service samples of single-worker FIFO backends, one lane per Request.  Such a backend starts its requests in
//     arrival order, so the request in slot off[b] + k consumes service draw k of backend b whatever happens before
//     it: the draw (Philox block, hs_log, two divisions-worth of fp64) is a pure function of (seed, b, k) and is taken
//     off the backend lane's serial recursion.  get_latency(...).to_seconds() of random.expovariate(1 / mean)
//     (distributions/exponential.py:36,43, components/server/server.py:246-247).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hs_lb_service_draws(const uint64_t *__restrict__ skey, const int64_t *n_ptr,
                                                           const int64_t *__restrict__ off, int tb, LbBe P, uint64_t seed,
                                                           double *__restrict__ sv_out, LbLayout LY, int B,
                                                           const LbTotals *tot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t b;
    uint64_t k;
    double *dst;
    if (tot->use_t) {                                        // slot i of the [k][backend] layout
        if (i >= (int64_t)tot->max_be * B) return;
        const int64_t kk = i < (1ll << 32) ? (int64_t)((uint32_t)i / (uint32_t)B) : i / B;
        b = i - kk * B;
        if (kk >= off[b + 1] - off[b]) return;
        k = (uint64_t)kk;
        dst = LY.tsv + i;
    } else {
        if (i >= *n_ptr) return;
        b = (int64_t)(skey[i] >> tb);
        k = (uint64_t)(i - off[b]);
        dst = sv_out + i;
    }
    const double mean = P.svc_mean[b];
    double sv;
    if (P.svc_kind[b] == HS_LAT_EXPONENTIAL) {
        const uint64_t sid = stream_id(P.base[b], kStreamService), blk = k >> 1;
        const U4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sid, (uint32_t)(sid >> 32), (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
        const double u = (k & 1) ? res53(o.z, o.w) : res53(o.x, o.y);
        const double lambda = __ddiv_rn(1.0, mean);
        sv = seconds_from_ns(ns_from_seconds(__ddiv_rn(exp1_from_uniform(u), lambda)));
    } else sv = seconds_from_ns(ns_from_seconds(mean));
    *dst = sv;
}

// ---------------------------------------------------------------------------------------------
// 3. Backends: QueuedResource + Queue + QueueDriver + worker in front of every Server
//    (components/queued_resource.py:52-143, queue.py:75-170, queue_driver.py:27-99, server/server.py:202-273)
// ---------------------------------------------------------------------------------------------
template <int C>
struct LbBackend {
    int32_t conc;
    uint32_t svc_kind, egress;
    double svc_lambda, svc_const_s;
    int64_t svc_const_ns, qcap;
    // arrivals
    const uint64_t *akey; uint64_t *aval;
    const double *asv;            // pre-drawn service time of the request in each slot (request-order loop)
    const uint64_t *rkey;         // request-order loop: element k of this backend at rkey[k * rs] / asv[k * rs]
    int64_t rs;                   // 1 (dense segment) or B ([k][backend] layout)
    int64_t ss;                   // stride of the completion logs (sink_t / sink_created / sink_S), same two cases
    int64_t ai, aend;
    uint64_t tmask;
    int64_t At, Acrt; uint32_t Adepth;
    // state
    int64_t buf, accepted, dropped, completed, rejected, started, received;
    int32_t active;
    uint32_t seq;
    int64_t D[C], crtD[C], crt[C];
    uint32_t seqD[C];
    double svc_s[C];
    double total_service;
    int64_t last_time;
    // lineage of the pending departures and of the event being processed (hs_station.hpp StationState::dpA)
    int32_t dpD[C], cd;
    int64_t rcD[C], cr;
    Stream svc;
    uint32_t ev[8];
    int64_t *adm, *sink_t, *sink_created, *sink_S;   // this backend's segment of the dense logs
    int pack_bits; int64_t slot0;                     // pack_bits > 0: sink_created holds (created_at << pack_bits) | slot (hs_lbk_scan)
    int qoverflow;
    uint8_t (*qmem)[kLbBlock];
    uint8_t (*qdep)[kLbBlock];
    int64_t (*qrc)[kLbBlock];
    int tid, qh, qn;

    __device__ __forceinline__ void qpush(uint32_t code) {
        if (qn >= kLbQCap) { qoverflow = 1; return; }
        const int slot = (qh + qn) % kLbQCap;
        qmem[slot][tid] = (uint8_t)code;
        qdep[slot][tid] = (uint8_t)(cd >= 254 ? 255 : cd + 1);   // created by the event being processed: one step further from the root
        qrc[slot][tid] = cr;
        ++qn;
    }
    __device__ __forceinline__ uint32_t qpop() {
        const uint32_t c = qmem[qh][tid];
        cd = qdep[qh][tid]; cr = qrc[qh][tid];
        qh = (qh + 1) % kLbQCap;
        --qn;
        return c;
    }
    __device__ __forceinline__ void load_arrival() {
        if (ai < aend) {
            const uint64_t v = aval[ai];
            At = (int64_t)(akey[ai] & tmask); Acrt = (int64_t)(v & kCrtMask); Adepth = (uint32_t)(v >> 56);
        } else { At = kInfNs; Acrt = 0; Adepth = 0; }
    }
    __device__ __forceinline__ int64_t peek_time(int64_t j) const { return j < aend ? (int64_t)(akey[j] & tmask) : kInfNs; }

    __device__ __forceinline__ void sample_service(double &s, int64_t &dur_ns) {
        if (svc_kind == HS_LAT_EXPONENTIAL) {   // random.expovariate(lambda) -> Duration.from_seconds -> to_seconds (server.py:246-247)
            const double sample = __ddiv_rn(exp1_from_uniform(svc.next_uniform()), svc_lambda);
            s = seconds_from_ns(ns_from_seconds(sample));
            dur_ns = ns_from_seconds(s);        // `yield s`: resume at now + int(s * 1e9) (core/event.py:499)
        } else { s = svc_const_s; dur_ns = svc_const_ns; }
    }
    // Queue._handle_enqueue (components/queue.py:122-147).  True: QUEUE_NOTIFY created.
    __device__ __forceinline__ bool do_enqueue(int64_t t) {
        ev[HS_EV_ENQUEUE]++;
        if (qcap >= 0 && buf >= qcap) { dropped++; return false; }       // FIFOQueue.push refuses (queue_policy.py:94-98)
        const bool was_empty = (buf == 0);
        adm[accepted] = t;                                               // context["created_at"] = the tick's time
        accepted++; buf++;
        return was_empty;
    }
    __device__ __forceinline__ bool do_notify() { ev[HS_EV_NOTIFY]++; return active < conc; }       // queue_driver.py:92-99
    __device__ __forceinline__ bool do_poll() {                                                       // queue.py:149-166
        ev[HS_EV_POLL]++;
        if (buf == 0) return false;
        buf--;
        return true;
    }
    // QUEUE_DELIVER + the retargeted payload at the worker (queue_driver.py:66-90, server/server.py:202-250).
    // Returns slot + 1 when the service takes zero nanoseconds (continuation inside this group), else 0.
    __device__ __forceinline__ uint32_t do_deliver_work(int64_t t) {
        ev[HS_EV_DELIVER]++; ev[HS_EV_WORK]++;
        const int64_t k = started++;
        if (active >= conc) { rejected++; return 0; }                    // acquire() failed (server.py:223-234)
        active++;
        double s; int64_t dur;
        sample_service(s, dur);
        const int64_t created = adm[k];
        int j = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; --i) if (D[i] == kInfNs) j = i;
        const int64_t d = t + dur;
        uint32_t same = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == j) {
            svc_s[i] = s; crt[i] = created; crtD[i] = t;
            if (d == t) { D[i] = kInfNs - 1; same = (uint32_t)i + 1; }
            else { D[i] = d; seqD[i] = seq++; dpD[i] = cd + 2 > 255 ? 255 : cd + 2; rcD[i] = cr; }   // QUEUE_DELIVER -> payload -> continuation
        }
        if (same) ++cd;                                                   // (the caller pushes the in-group continuation: deliver + 2)
        return same;
    }
    // generator resumes (server/server.py:252-273): statistics, forward to the Sink (components/common.py:36-44),
    // schedule_poll hook (queue_driver.py:79-84).  True: QUEUE_POLL created.
    __device__ __forceinline__ bool do_cont(int slot, int64_t t) {
        ev[HS_EV_CONTINUATION]++;
        double s = 0.0; int64_t cr = 0, st = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == slot) { s = svc_s[i]; cr = crt[i]; st = crtD[i]; D[i] = kInfNs; }
        active = active > 0 ? active - 1 : 0;
        completed++;
        total_service = __dadd_rn(total_service, s);
        if (egress == HS_EGRESS_SINK) {
            ev[HS_EV_SINK]++;
            sink_t[received * ss] = t; sink_S[received * ss] = st;
            sink_created[received * ss] = pack_bits ? (int64_t)(((uint64_t)cr << pack_bits) | (uint64_t)(slot0 + received * ss)) : cr;
            received++;
        }
        return active < conc;
    }
    __device__ __forceinline__ bool chain_from_poll(int64_t t) {       // (cd = the QUEUE_POLL's steps from the group's root)
        if (!do_poll()) return false;
        ++cd;
        const uint32_t same = do_deliver_work(t);
        if (same) { qpush(LQ_CONT | ((same - 1) << 3)); return true; }
        return false;
    }
    __device__ __forceinline__ void drain(int64_t t) {
        while (qn > 0) {
            const uint32_t code = qpop();
            switch (code & 7u) {
                case LQ_PRE:                      // SourceEvent -> Request@LoadBalancer -> Request@Server, one hop per generation
                    if ((code >> 3) > 0) qpush(LQ_PRE | (((code >> 3) - 1) << 3));
                    else if (do_enqueue(t)) qpush(LQ_NOTIFY);
                    break;
                case LQ_NOTIFY: if (do_notify()) qpush(LQ_POLL); break;
                case LQ_POLL: if (do_poll()) qpush(LQ_DELIVER); break;
                case LQ_DELIVER: { const uint32_t sm = do_deliver_work(t); if (sm) qpush(LQ_CONT | ((sm - 1) << 3)); } break;
                case LQ_CONT: if (do_cont((int)(code >> 3), t)) qpush(LQ_POLL); break;
                default: break;
            }
        }
    }
    // Among the arrivals at time t (a run of the sorted list starting at ai) bring the one whose SourceEvent was
    // created first to the head (stable: list order on equal stamps).  Runs longer than one are rare.
    __device__ __forceinline__ void order_arrival_run(int64_t t) {
        int64_t m = ai; uint64_t best = aval[ai] & kCrtMask;
        for (int64_t j = ai + 1; peek_time(j) == t; ++j) {
            const uint64_t c = aval[j] & kCrtMask;
            if (c < best) { best = c; m = j; }
        }
        if (m != ai) {
            const uint64_t v = aval[m];
            for (int64_t j = m; j > ai; --j) aval[j] = aval[j - 1];
            aval[ai] = v;
            load_arrival();
        }
    }
    __device__ __forceinline__ void run_group_general(int64_t t) {
        for (;;) {
            // pending departure at t with the earliest creation
            int bd = -1; uint32_t bs = 0;
#pragma unroll
            for (int i = 0; i < C; ++i)
                if (D[i] == t && (bd < 0 || (int32_t)(seqD[i] - bs) < 0)) { bd = i; bs = seqD[i]; }
            const bool arr = (At == t);
            if (arr) order_arrival_run(t);
            if (bd < 0 && !arr) break;
            bool take_arr = arr;
            if (arr && bd >= 0) {
                int64_t cd = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == bd) cd = crtD[i];
                take_arr = Acrt < cd;             // the backend's own event first on equal creation times
            }
            if (take_arr) {
                cd = 0; cr = Acrt;                    // the chain's root: the SourceEvent that heads the tick's same-ns chain
                qpush(LQ_PRE | ((1u + Adepth) << 3));
                ++ai;
                load_arrival();
            } else {
                cd = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == bd) cr = crtD[i];
                if (do_cont(bd, t)) qpush(LQ_POLL);
            }
        }
        drain(t);
    }
    __device__ __forceinline__ int64_t next_time() const {
        int64_t t = At;
#pragma unroll
        for (int i = 0; i < C; ++i) t = D[i] < t ? D[i] : t;
        return t;
    }

    // ---- one worker, unbounded FIFO: the backend in REQUEST order instead of event order ------------------------
    // Request k of the arrival list starts at S_k = max(a_k, D_{k-1}) and departs at D_k = S_k + service_k; service
    // draws are consumed in start order = arrival order.  Its reference events happen at a_k (Request@Server,
    // QUEUE_NOTIFY iff the buffer is empty, QUEUE_POLL iff the worker is idle then), at S_k (QUEUE_DELIVER,
    // Request@worker) and at D_k (ProcessContinuation, Request@Sink, the completion QUEUE_POLL), and count iff that
    // time is <= T.  Same-nanosecond coincidences have ONE outcome each here, whatever the order of the roots, because
    // an arrival's Request@Server is two generations below its SourceEvent while a completion's QUEUE_POLL is one
    // below its ProcessContinuation (the POLL always runs first):
    //   a_k == D_{k-1}, nobody waiting:  the completion's POLL finds the buffer empty, then the enqueue notifies an
    //                                    idle worker -> NOTIFY, POLL, start at a_k;
    //   a_k == S_{k-1} == D_{k-2}:       request k-1 left the buffer through the completion's POLL before the enqueue
    //                                    -> NOTIFY, but the worker is busy again when it runs -> no POLL;
    //   a_k == a_{k-1} == S_{k-1}:       both enqueues precede the first NOTIFY's POLL -> the buffer is not empty.
    // A zero-nanosecond service (probability ~1e-8 per request) chains further events into the same timestamp: the
    // lane bails out and is re-run from its initial state by the event-order loop.
    __device__ __forceinline__ bool run_request_order(int64_t T) {
        const int64_t n = aend - ai;
        const uint64_t *kp = rkey;
        int64_t Sprev = INT64_MIN, Dprev = INT64_MIN, aprev = INT64_MIN, lt = INT64_MIN;
        uint32_t n_notify = 0, n_poll = 0, n_start = 0, n_dep = 0;
        bool pend = false;
        int64_t pendD = kInfNs, pendS = 0, pendA = 0, pendSprev = INT64_MIN, pendK = 0;
        double pend_s = 0.0, tsvc = 0.0;
        // A block of requests is LOADED, then processed, with ONE explicit vmcnt(0) in between: gfx9 counts loads and stores with one
        // counter, and inside this control flow the compiler's wait before the first use of a prefetched register was vmcnt(0) at every
        // request (the disassembly of round 2's double-buffered version), i.e. every request also waited for the record stores of
        // the request before it.  Measured (scratch micro-kernels, round 3): a lone wavefront pays ~0.11 us per step for batched
        // loads, ~0.06 us more per 512-byte store it issues, ~0.7 us per step when a load's round trip is exposed; this loop is
        // ~0.95 us per request -- three record stores, ~125 instructions with five divergent branches -- so the stores and the
        // dependent instruction chain, not the loads, are what is left.
        constexpr int kAhead = 16;
        const double *sp = asv;
        uint64_t cur[kAhead];
        double scur[kAhead];
        bool blocked = false;
        for (int64_t base = 0; base < n && !blocked; base += kAhead) {
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const bool in = (base + j) < n;
                cur[j] = in ? kp[(base + j) * rs] : 0ull;
                scur[j] = in ? sp[(base + j) * rs] : 0.0;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) HERE, on every path (see above): the block's values have arrived
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                if (base + j >= n || blocked) continue;
                const int64_t a = (int64_t)(cur[j] & tmask);
                const bool notify = Sprev < a || (Sprev == a && aprev < a);
                const bool idle = notify && Dprev <= a;
                const int64_t S = Dprev > a ? Dprev : a;
                n_notify += notify ? 1u : 0u;
                lt = a > lt ? a : lt;
                if (S > T) { blocked = true; continue; }              // this request and every later one never start
                const double sv = scur[j];                            // hs_lb_service_draws
                const int64_t dur = ns_from_seconds(sv);              // `yield s`: resume at now + int(s * 1e9) (core/event.py:499)
                if (dur == 0) return false;
                const int64_t Dk = S + dur;
                n_poll += idle ? 1u : 0u;
                ++n_start;
                lt = S > lt ? S : lt;
                if (Dk <= T) {
                    lt = Dk > lt ? Dk : lt;
                    tsvc = __dadd_rn(tsvc, sv);
                    if (egress == HS_EGRESS_SINK) { sink_t[n_dep * ss] = Dk; sink_created[n_dep * ss] = a; sink_S[n_dep * ss] = S; }
                    ++n_dep;
                    pend = false;
                } else { pend = true; pendD = Dk; pendS = S; pend_s = sv; pendA = a; pendSprev = Sprev; pendK = base + j; }
                Sprev = S; Dprev = Dk; aprev = a;
            }
        }
        if (n > 0) {                                                  // arrivals behind a blocked head were not iterated
            const int64_t a_last = (int64_t)(kp[(n - 1) * rs] & tmask);
            lt = a_last > lt ? a_last : lt;
        }
        // fold into the LP state exactly as the event-order loop would have left it
        ev[HS_EV_ENQUEUE] = (uint32_t)n; ev[HS_EV_NOTIFY] = n_notify; ev[HS_EV_POLL] = n_poll + n_dep;
        ev[HS_EV_DELIVER] = n_start; ev[HS_EV_WORK] = n_start; ev[HS_EV_CONTINUATION] = n_dep;
        ev[HS_EV_SINK] = egress == HS_EGRESS_SINK ? n_dep : 0u;
        accepted = n; started = n_start; completed = n_dep; received = egress == HS_EGRESS_SINK ? n_dep : 0;
        buf = n - (int64_t)n_start;
        active = pend ? 1 : 0;
        D[0] = pend ? pendD : kInfNs; crtD[0] = pendS; svc_s[0] = pend_s; seqD[0] = 0;
        if (pend) {      // lineage of the pending departure (only the election beyond end_time reads it)
            if (pendS == pendA) {          // started on arrival: SourceEvent -> Request@LoadBalancer -> Request@Server -> QUEUE_NOTIFY ->
                const uint64_t v = aval[ai + pendK];   // QUEUE_POLL -> QUEUE_DELIVER -> payload -> continuation, below the tick's chain root
                dpD[0] = (int32_t)(v >> 56) + 7; rcD[0] = (int64_t)(v & kCrtMask);
            } else { dpD[0] = 4; rcD[0] = pendSprev; }   // started when the request before it left: four steps below that continuation
        }
        total_service = tsvc;
        last_time = n > 0 ? lt : INT64_MIN;
        ai = aend; At = kInfNs;
        return true;
    }
    __device__ __forceinline__ void run_group(int64_t t, bool force_general) {
        int n_at = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) n_at += (D[i] == t) ? 1 : 0;
        const bool arr = (At == t);
        if (arr) n_at += (peek_time(ai + 1) == t) ? 2 : 1;
        if (n_at == 1 && !force_general) {
            bool want_poll, general = false;
            if (arr) {
                cr = Acrt; cd = (int32_t)Adepth + 4;   // tick (Adepth) -> Request@LoadBalancer -> Request@Server -> QUEUE_NOTIFY -> QUEUE_POLL
                ++ai;
                load_arrival();
                want_poll = do_enqueue(t) && do_notify();
            } else {
                int slot = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (D[i] == t) slot = i;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == slot) cr = crtD[i];
                cd = 1;                                // continuation -> QUEUE_POLL
                want_poll = do_cont(slot, t);
            }
            if (want_poll) general = chain_from_poll(t);
            if (general) drain(t);
        } else run_group_general(t);
        last_time = t;
    }
};

template <int C>
__global__ void __launch_bounds__(kLbBlock) hs_lbk_backends(LbBe P, int B, int S, uint64_t seed, int64_t start_ns,
                                                           int64_t end_ns, const uint64_t *__restrict__ skey,
                                                           uint64_t *__restrict__ sval, const int64_t *__restrict__ off,
                                                           int tb, int64_t *__restrict__ adm, int64_t *__restrict__ sink_t,
                                                           int64_t *__restrict__ sink_created, int64_t *__restrict__ sink_S,
                                                           const double *__restrict__ svdraw, LbTotals *tot, int flags,
                                                           LbLayout LY, int lanes, const uint8_t *__restrict__ only, int pack_bits) {
    __shared__ uint8_t qmem[kLbQCap][kLbBlock];
    __shared__ uint8_t qdep[kLbQCap][kLbBlock];          // lineage of the in-group FIFO's entries
    __shared__ int64_t qrc[kLbQCap][kLbBlock];
    __shared__ LbCand wc[kLbBlock / 64];
    const int tid = threadIdx.x;
    // `lanes` backends per wavefront (lb_lanes()): one backend per lane would leave half the SIMDs without a wavefront at the
    // configs[4] size, and a lone wavefront per SIMD waits out every dependent instruction
    const int b = ((blockIdx.x * kLbBlock + tid) >> 6) * lanes + (tid & 63);
    // `only` != nullptr: the backends hs_lbk_scan handed back (bounded queues, a zero-nanosecond service), nobody else
    const bool live = (tid & 63) < lanes && b < B && (only == nullptr || only[b] != 0);
    LbCand c = lb_cand_none(S + (live ? b : 0));
    LbBackend<C> X;
#pragma unroll
    for (int k = 0; k < 8; ++k) X.ev[k] = 0;
    X.qoverflow = 0; X.last_time = INT64_MIN; X.completed = 0; X.received = 0;
    if (live) {
        X.conc = P.conc[b]; X.svc_kind = P.svc_kind[b]; X.egress = P.egress[b]; X.qcap = P.qcap[b];
        const double mean = P.svc_mean[b];
        X.svc_lambda = __ddiv_rn(1.0, mean);                             // ExponentialLatency._lambda = 1 / mean
        X.svc_const_s = seconds_from_ns(ns_from_seconds(mean));          // ConstantLatency
        X.svc_const_ns = ns_from_seconds(X.svc_const_s);
        X.akey = skey; X.aval = sval; X.asv = svdraw; X.ai = off[b]; X.aend = off[b + 1];
        X.tmask = tb >= 64 ? ~0ull : ((1ull << tb) - 1);
        X.buf = 0; X.accepted = 0; X.dropped = 0; X.rejected = 0; X.started = 0; X.active = 0; X.seq = 0;
        X.total_service = 0.0;
#pragma unroll
        for (int i = 0; i < C; ++i) { X.D[i] = kInfNs; X.crtD[i] = start_ns; X.crt[i] = 0; X.seqD[i] = 0; X.svc_s[i] = 0.0; X.dpD[i] = 0; X.rcD[i] = INT64_MIN; }
        X.cd = 0; X.cr = INT64_MIN;
        X.svc.init(seed, stream_id(P.base[b], kStreamService), 0);
        const int64_t o = off[b];
        const bool T = tot->use_t != 0;
        X.rkey = T ? LY.tkey + b : skey + o; X.asv = T ? LY.tsv + b : svdraw + o; X.rs = T ? B : 1;
        X.ss = T ? B : 1;
        const int64_t so = T ? b : o;
        X.adm = adm + o; X.sink_t = sink_t + so; X.sink_created = sink_created + so; X.sink_S = sink_S + so;
        X.pack_bits = only != nullptr ? pack_bits : 0; X.slot0 = so;
        X.qmem = qmem; X.qdep = qdep; X.qrc = qrc; X.tid = tid; X.qh = 0; X.qn = 0;
        const bool force_general = (flags & 1) != 0;
        bool event_order = true;
        if constexpr (C == 1) {
            if (!force_general && (flags & 2) == 0 && X.qcap < 0 && X.conc == 1) {
                event_order = !X.run_request_order(end_ns);
                if (event_order) {                                       // zero-length service: start over in event order
#pragma unroll
                    for (int k = 0; k < 8; ++k) X.ev[k] = 0;
                    X.svc.init(seed, stream_id(P.base[b], kStreamService), 0);
                }
            }
        }
        if (event_order) {
            X.load_arrival();
            for (;;) {
                const int64_t t = X.next_time();
                if (t > end_ns) break;                                   // also kInfNs: nothing pending
                X.run_group(t, force_general);
            }
        }
        P.accepted[b] = X.accepted; P.dropped[b] = X.dropped; P.completed[b] = X.completed; P.rejected[b] = X.rejected;
        P.received[b] = X.received; P.depth[b] = X.buf; P.active[b] = X.active; P.total_service[b] = X.total_service;
        // the earliest pending departure is this backend's candidate for the one event beyond end_ns
        uint32_t bs = 0;
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (X.D[i] != kInfNs && (!c.valid || X.D[i] < c.t || (X.D[i] == c.t && (int32_t)(X.seqD[i] - bs) < 0))) {
                c.t = X.D[i]; c.t_created = X.crtD[i]; c.svc_s = X.svc_s[i]; c.valid = 1; bs = X.seqD[i];
                c.depth = X.dpD[i]; c.rcrt = X.rcD[i];
            }
    }
    block_min_cand(c, wc, P.cand);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const uint32_t s = wave_sum<uint32_t>(live ? X.ev[k] : 0u);
        if ((tid & 63) == 0 && s) atomicAdd(&tot->ev[k], (unsigned long long)s);
    }
    const uint32_t sc = wave_sum<uint32_t>(live ? X.ev[HS_EV_CONTINUATION] : 0u), sr = wave_sum<uint32_t>(live ? X.ev[HS_EV_SINK] : 0u);
    if ((tid & 63) == 0) {
        if (sc) atomicAdd(&tot->completed, (unsigned long long)sc);
        if (sr) atomicAdd(&tot->received, (unsigned long long)sr);
    }
    {   // (one atomic per wavefront, not per lane: same-address atomics serialise in the L2)
        long long ml = live ? (long long)X.last_time : INT64_MIN;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const long long d = __shfl_xor(ml, o, 64); ml = d > ml ? d : ml; }
        if ((tid & 63) == 0 && ml != INT64_MIN) atomicMax(&tot->last_time, ml);
    }
    if (live && X.qoverflow) atomicOr(&tot->qoverflow, 1);
}

// ---------------------------------------------------------------------------------------------
// 3b. One-worker unbounded FIFO backends as a SEGMENTED (max, +) SCAN (round 4): one WAVEFRONT per backend over its dense
//     segment of the sorted arrival list, two requests per lane, 128 per step.
//
// run_request_order above is a serial recursion per backend lane -- D_k = max(a_k, D_{k-1}) + dur_k -- on 512 wavefronts (half the
// SIMDs idle, ~1 us per request: 354 us at the configs[4] size, behind a 108 us transpose into the [k][backend] layout it needs
// for coalescing and a 129 us kernel that draws the service times).  But request k's step is the map x -> max(x + dur_k, a_k + dur_k),
// such maps compose associatively (x -> max(x + p, q) then x -> max(x + p', q') is x -> max(x + p + p', max(q + p', q'))), exactly,
// in int64 ns: the departures of a backend are a prefix scan.  Everything else the request-order loop derives is a function of
// (a_k, a_{k-1}, S_{k-1}, D_{k-1}, S_k, D_k) -- the same formulas, evaluated by every lane for its own requests:
//     notify_k = S_{k-1} < a_k || (S_{k-1} == a_k && a_{k-1} < a_k)         idle_k = notify_k && D_{k-1} <= a_k
//     started_k = S_k <= T (S is monotone: the requests behind a blocked head never start)      departed_k = D_k <= T
// and the counts are ballots.  The service draw of request k is a pure function of (seed, backend, k) and is computed right here
// (one Philox block serves the lane's two requests).  What stays serial is the binary64 `_total_service_time` -- the reference
// adds the service times in completion order, and fp addition does not re-associate -- so the wavefront adds the step's
// samples one after the other (128 dependent v_add_f64 per step, issue-bound, hidden behind the other wavefronts of the SIMD).
// Completion records go to the DENSE segment (slot = position in the sorted arrival list) with coalesced stores; a request that
// does not complete by T leaves an invalid marker, so the Sink merge scans a dense input.  Backends this formulation does not
// cover (bounded queue, a zero-nanosecond service among the started requests) are handed to hs_lbk_backends<1> (`redo`).
// Bit-identical to run_request_order (tests/test_gpu_lb.py compares both against the oracle and each other, debug flag 64).
// ---------------------------------------------------------------------------------------------
constexpr int64_t kNegInfNs = INT64_MIN / 4;      // "-infinity" of the (max, +) maps: sums of durations stay far from overflow
constexpr int64_t kSinkInvalid = INT64_MAX;       // sink_t marker: this slot of the dense completion log holds no record

struct MaxPlus { int64_t p, q; };                 // x -> max(x + p, q)
__device__ __forceinline__ MaxPlus mp_then(const MaxPlus &f, const MaxPlus &g) {   // first f, then g
    const int64_t a = f.q + g.p;
    return MaxPlus{f.p + g.p, a > g.q ? a : g.q};
}

// F64: every time of the run is a whole number of nanoseconds below 2^50 (hs_lb::f64_times), so the (max, +) maps, the service
// durations and the comparisons run on binary64 values -- exact there -- and lose the 64-bit integer pairs and the i64 <-> f64
// conversion sequences (hs_device.hpp ns_from_seconds_d); bit-identical to the int64 instantiation.
// the value of lane `l` (wave-uniform) as a wave-uniform value: two v_readlane into SGPRs, so that what the whole wavefront shares
// (carries, the pending request) does not occupy vector registers
template <typename T>
__device__ __forceinline__ T bcast64(T v, int l) {
    static_assert(sizeof(T) == 8, "64-bit values");
    uint64_t u;
    __builtin_memcpy(&u, &v, 8);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    u = ((uint64_t)hi << 32) | lo;
    T r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
template <bool F64> struct ScanT { using T = int64_t; };
template <> struct ScanT<true> { using T = double; };

template <bool F64>
__global__ void __launch_bounds__(kLbBlock) hs_lbk_scan(LbBe P, int B, int S, uint64_t seed, int64_t T_ns,
                                                          const uint64_t *__restrict__ skey, const uint64_t *__restrict__ sval,
                                                          const int64_t *__restrict__ off, int tb, int64_t *__restrict__ sink_t,
                                                          int64_t *__restrict__ sink_created, int64_t *__restrict__ sink_S,
                                                          uint32_t *__restrict__ bev, int64_t *__restrict__ blast,
                                                          uint8_t *__restrict__ redo, LbCand *__restrict__ cand_out, int pack_bits) {
    // pack_bits > 0 (one Sink shared by all backends): sink_created holds (created_at << pack_bits) | slot, the value the Sink merge
    // carries -- its first pass then reads a ready-made (key, value) pair like every later pass
    using TT = typename ScanT<F64>::T;
    __shared__ LbCand wc[kLbBlock / 64];
    __shared__ double ssum[kLbBlock / 64][128];                             // the step's service samples, in completion order
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = (int)(((size_t)blockIdx.x * kLbBlock + threadIdx.x) >> 6);
    LbCand c = lb_cand_none(S + (b < B ? b : 0));
    const TT neg_inf = F64 ? (TT)(-__builtin_huge_val()) : (TT)kNegInfNs;
    auto to_i64 = [](TT v) -> int64_t { if constexpr (F64) return i64_from_whole_d((double)v); else return (int64_t)v; };
    if (b < B) {
        const int64_t o = off[b], n = off[b + 1] - o;
        const uint64_t tmask = tb >= 64 ? ~0ull : ((1ull << tb) - 1);
        const bool eligible = P.conc[b] == 1 && P.qcap[b] < 0;
        const uint32_t svc_kind = P.svc_kind[b];
        const bool sink = P.egress[b] == HS_EGRESS_SINK;
        const double mean = P.svc_mean[b];
        ConstDiv by_lambda;
        by_lambda.init(__ddiv_rn(1.0, mean));                              // ExponentialLatency._lambda = 1 / mean
        const double const_s = seconds_from_ns(ns_from_seconds(mean));     // ConstantLatency
        const uint64_t sid = stream_id(P.base[b], kStreamService);
        const TT T = (TT)T_ns;
        TT carryD = neg_inf, carryS = neg_inf, carryA = neg_inf;
        uint32_t n_notify = 0, n_poll = 0, n_start = 0, n_dep = 0;
        double tsvc = 0.0;
        bool bail = !eligible;
        TT lastS = neg_inf, lastD = neg_inf;
        bool pend = false;
        TT pendD = 0, pendS = 0, pendA = 0, pendSprev = 0;
        int64_t pendK = 0;
        double pend_s = 0.0;
        // the keys of a step are loaded while the step before it is computed
        uint64_t nk0 = (2 * lane < n) ? skey[o + 2 * lane] : 0ull, nk1 = (2 * lane + 1 < n) ? skey[o + 2 * lane + 1] : 0ull;
        for (int64_t base = 0; base < n && !bail; base += 128) {
            const int64_t k0 = base + 2 * lane, k1 = k0 + 1;
            const bool in0 = k0 < n, in1 = k1 < n;
            const uint64_t key0 = nk0, key1 = nk1;
            nk0 = (k0 + 128 < n) ? skey[o + k0 + 128] : 0ull;
            nk1 = (k1 + 128 < n) ? skey[o + k1 + 128] : 0ull;
            TT a0, a1;
            if constexpr (F64) {       // whole ns < 2^52: the bits under an exponent of 2^52, minus 2^52
                a0 = __longlong_as_double((long long)((key0 & tmask) | 0x4330000000000000ull)) - 4503599627370496.0;
                a1 = __longlong_as_double((long long)((key1 & tmask) | 0x4330000000000000ull)) - 4503599627370496.0;
            } else { a0 = (TT)(key0 & tmask); a1 = (TT)(key1 & tmask); }
            double sv0 = const_s, sv1 = const_s;
            if (svc_kind == HS_LAT_EXPONENTIAL) {   // random.expovariate(lambda) -> Duration.from_seconds -> to_seconds (server.py:246-247)
                const uint64_t blk = (uint64_t)k0 >> 1;                      // draws k0 (even) and k0 + 1 share a block
                const U4 q = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sid, (uint32_t)(sid >> 32), (uint32_t)seed,
                                           (uint32_t)(seed >> 32));
                const double e0 = by_lambda.div(exp1_from_uniform(res53(q.x, q.y))), e1 = by_lambda.div(exp1_from_uniform(res53(q.z, q.w)));
                if constexpr (F64) { sv0 = seconds_from_ns_d(ns_from_seconds_d(e0)); sv1 = seconds_from_ns_d(ns_from_seconds_d(e1)); }
                else { sv0 = seconds_from_ns(ns_from_seconds(e0)); sv1 = seconds_from_ns(ns_from_seconds(e1)); }
            }
            TT dur0, dur1;                                                   // `yield s`: now + int(s * 1e9) (core/event.py:499)
            if constexpr (F64) { dur0 = ns_from_seconds_d(sv0); dur1 = ns_from_seconds_d(sv1); }
            else { dur0 = ns_from_seconds(sv0); dur1 = ns_from_seconds(sv1); }
            // inclusive scan over the lanes' pairs of the maps x -> max(x + p, q)
            TT ip = (in0 ? dur0 : (TT)0) + (in1 ? dur1 : (TT)0);
            TT iq;
            {
                const TT q0 = in0 ? a0 + dur0 : neg_inf, q1 = in1 ? a1 + dur1 : neg_inf;
                const TT x = q0 + (in1 ? dur1 : (TT)0);
                iq = x > q1 ? x : q1;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const TT up_p = __shfl_up(ip, d, 64), up_q = __shfl_up(iq, d, 64);
                if (lane >= d) { const TT x = up_q + ip; iq = x > iq ? x : iq; ip = up_p + ip; }
            }
            TT ep = __shfl_up(ip, 1, 64), eq = __shfl_up(iq, 1, 64);          // everything of this step before the lane's pair
            if (lane == 0) { ep = (TT)0; eq = neg_inf; }
            const TT x0 = carryD + ep;
            const TT Dp0 = x0 > eq ? x0 : eq;                                // D of request k0 - 1 (-infinity before the first)
            const TT S0 = Dp0 > a0 ? Dp0 : a0, D0 = S0 + dur0;
            const TT S1 = D0 > a1 ? D0 : a1, D1 = S1 + dur1;
            TT Sp0 = __shfl_up(in1 ? S1 : S0, 1, 64), Ap0 = __shfl_up(in1 ? a1 : a0, 1, 64);
            if (lane == 0) { Sp0 = carryS; Ap0 = carryA; }
            const bool notify0 = in0 && (Sp0 < a0 || (Sp0 == a0 && Ap0 < a0));
            const bool notify1 = in1 && (S0 < a1 || (S0 == a1 && a0 < a1));
            const bool st0 = in0 && S0 <= T, st1 = in1 && S1 <= T;
            const bool idle0 = notify0 && Dp0 <= a0 && st0, idle1 = notify1 && D0 <= a1 && st1;
            const bool dp0 = st0 && D0 <= T, dp1 = st1 && D1 <= T;
            if (__ballot((st0 && dur0 == (TT)0) || (st1 && dur1 == (TT)0))) { bail = true; break; }   // a zero-nanosecond service: event order
            const uint64_t bs0 = __ballot(st0), bs1 = __ballot(st1), bd0 = __ballot(dp0), bd1 = __ballot(dp1);
            n_notify += (uint32_t)(__popcll(__ballot(notify0)) + __popcll(__ballot(notify1)));
            n_poll += (uint32_t)(__popcll(__ballot(idle0)) + __popcll(__ballot(idle1)));
            n_start += (uint32_t)(__popcll(bs0) + __popcll(bs1));
            n_dep += (uint32_t)(__popcll(bd0) + __popcll(bd1));
            if (in0) { sink_t[o + k0] = dp0 && sink ? to_i64(D0) : kSinkInvalid; if (dp0 && sink) { sink_created[o + k0] = pack_bits ? (int64_t)(((uint64_t)to_i64(a0) << pack_bits) | (uint64_t)(o + k0)) : to_i64(a0); sink_S[o + k0] = to_i64(S0); } }
            if (in1) { sink_t[o + k1] = dp1 && sink ? to_i64(D1) : kSinkInvalid; if (dp1 && sink) { sink_created[o + k1] = pack_bits ? (int64_t)(((uint64_t)to_i64(a1) << pack_bits) | (uint64_t)(o + k1)) : to_i64(a1); sink_S[o + k1] = to_i64(S1); } }
            if (bd0) {       // _total_service_time: left to right, in completion order -- through LDS: broadcast reads off the VALU
                ssum[w][2 * lane] = dp0 ? sv0 : 0.0;                        // (+ 0.0 is exact; departures are a prefix)
                ssum[w][2 * lane + 1] = dp1 ? sv1 : 0.0;
                const int cnt = 2 * (64 - (int)__builtin_clzll(bd0));
                for (int j = 0; j < cnt; j += 8) {
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = ssum[w][j + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) tsvc = __dadd_rn(tsvc, v[q]);   // (slots past cnt of the last 8 hold this step's 0.0 / samples not yet departed: 0.0)
                }
            }
            if (bs0) {                                                       // the last request that started so far: pending iff not departed
                const int hl = 63 - (int)__builtin_clzll(bs0);
                const bool second = ((bs1 >> hl) & 1ull) != 0;
                const TT Sl = bcast64(second ? S1 : S0, hl), Dl = bcast64(second ? D1 : D0, hl);
                lastS = Sl;
                pend = Dl > T;
                if (pend) {
                    pendD = Dl; pendS = Sl; pend_s = bcast64(second ? sv1 : sv0, hl); pendA = bcast64(second ? a1 : a0, hl);
                    pendSprev = bcast64(second ? S0 : Sp0, hl); pendK = base + 2 * hl + (second ? 1 : 0);
                }
            }
            if (bd0) {
                const int hl = 63 - (int)__builtin_clzll(bd0);
                const bool second = ((bd1 >> hl) & 1ull) != 0;
                lastD = bcast64(second ? D1 : D0, hl);
            }
            // carries: the step's last valid request
            const int64_t left = n - base;
            const int ll = left >= 128 ? 63 : (int)((left - 1) >> 1);
            const bool two = left >= 128 || ((left & 1) == 0);
            carryD = bcast64(two ? D1 : D0, ll); carryS = bcast64(two ? S1 : S0, ll); carryA = bcast64(two ? a1 : a0, ll);
            if (!bs0) {                                                      // S is monotone: nobody behind this step starts either
                for (int64_t r = base + 128 + lane; r < n; r += 64) sink_t[o + r] = kSinkInvalid;
                if (n > 0) carryA = (TT)(int64_t)(skey[o + n - 1] & tmask);
                break;
            }
        }
        if (bail) {                                                          // hs_lbk_backends<1> runs this backend; its log starts empty
            for (int64_t r = lane; r < n; r += 64) sink_t[o + r] = kSinkInvalid;
            if (lane == 0) redo[b] = 1;
        } else if (lane == 0) {
            redo[b] = 0;
            P.accepted[b] = n; P.dropped[b] = 0; P.completed[b] = n_dep; P.rejected[b] = 0;
            P.received[b] = sink ? n_dep : 0; P.depth[b] = n - (int64_t)n_start; P.active[b] = pend ? 1 : 0;
            P.total_service[b] = tsvc;
            bev[b] = n_notify; bev[(size_t)B + b] = n_poll; bev[(size_t)2 * B + b] = n_start;
            TT lt = n > 0 ? carryA : neg_inf;                                // the last arrival (arrivals are sorted), last start, last departure
            lt = lastS > lt ? lastS : lt; lt = lastD > lt ? lastD : lt;
            blast[b] = n > 0 ? to_i64(lt) : INT64_MIN;
            if (pend) {          // the one pending departure: this backend's candidate for the event beyond end_time, with its lineage
                c.t = to_i64(pendD); c.t_created = to_i64(pendS); c.svc_s = pend_s; c.valid = 1;
                if (pendS == pendA) {          // started on arrival: SourceEvent -> Request@LoadBalancer -> Request@Server -> QUEUE_NOTIFY ->
                    const uint64_t v = sval[o + pendK];   // QUEUE_POLL -> QUEUE_DELIVER -> payload -> continuation, below the tick's chain root
                    c.depth = (int)(v >> 56) + 7; c.rcrt = (int64_t)(v & kCrtMask);
                } else { c.depth = 4; c.rcrt = to_i64(pendSprev); }   // started when the request before it left: four steps below that continuation
            }
        }
    }
    block_min_cand(c, wc, cand_out);
}

// totals of the backends hs_lbk_scan ran (the ones it handed back add theirs in hs_lbk_backends) and the earliest of its workgroups'
// candidates -> scan_cand[0].  kScanParts workgroups reduce their share to one partial each; the last one to finish (a ticket) adds
// the partials up -- no same-address atomics (512 wavefronts adding to the same eight words took 49 us; one workgroup walking
// all 32 768 backends 151 us).
constexpr int kScanParts = 64;
struct ScanPartial { unsigned long long v[6]; long long lt; LbCand c; };
static_assert(sizeof(ScanPartial) % 8 == 0, "whole words");

__global__ void __launch_bounds__(256) hs_lb_scan_totals(LbBe P, int B, const uint32_t *__restrict__ bev, const int64_t *__restrict__ blast,
                                                        const uint8_t *__restrict__ redo, LbTotals *tot, LbCand *__restrict__ scan_cand,
                                                        int n_cand, ScanPartial *__restrict__ part, unsigned *__restrict__ ticket) {
    __shared__ unsigned long long sv[4][6];
    __shared__ long long slt[4];
    __shared__ LbCand wc[4];
    __shared__ bool is_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    long long lt = INT64_MIN;
    for (int b = blockIdx.x * 256 + tid; b < B; b += kScanParts * 256) {
        if (redo[b] != 0) continue;
        const unsigned long long n = (unsigned long long)P.accepted[b], dep = (unsigned long long)P.completed[b];
        v[0] += n;                                                   // Request@Server
        v[1] += bev[b];                                              // QUEUE_NOTIFY
        v[2] += bev[(size_t)B + b] + dep;                            // QUEUE_POLL: idle arrivals + one per completion
        v[3] += bev[(size_t)2 * B + b];                              // QUEUE_DELIVER = Request@worker
        v[4] += dep;                                                 // ProcessContinuation
        v[5] += (unsigned long long)P.received[b];                   // Request@Sink
        const long long l = (long long)blast[b];
        lt = l > lt ? l : lt;
    }
    LbCand best = lb_cand_none(0);
    for (int i = blockIdx.x * 256 + tid; i < n_cand; i += kScanParts * 256) { const LbCand c = scan_cand[i]; if (cand_before(c, best)) best = c; }
    auto reduce_block = [&]() {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = wave_sum<unsigned long long>(v[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long d = __shfl_xor(lt, o, 64); lt = d > lt ? d : lt;
            const LbCand cd = cand_shfl_xor(best, o);
            if (cand_before(cd, best)) best = cd;
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) sv[w][k] = v[k];
            slt[w] = lt; wc[w] = best;
        }
        __syncthreads();
        if (tid == 0)
            for (int q = 1; q < 4; ++q) {
                for (int k = 0; k < 6; ++k) v[k] += sv[q][k];
                lt = slt[q] > lt ? slt[q] : lt;
                if (cand_before(wc[q], best)) best = wc[q];
            }
    };
    reduce_block();
    if (tid == 0) {
        ScanPartial pp;
        for (int k = 0; k < 6; ++k) pp.v[k] = v[k];
        pp.lt = lt; pp.c = best;
        unsigned long long raw[sizeof(ScanPartial) / 8];
        __builtin_memcpy(raw, &pp, sizeof pp);
        unsigned long long *q = (unsigned long long *)&part[blockIdx.x];
        for (int k = 0; k < (int)(sizeof(ScanPartial) / 8); ++k) __hip_atomic_store(q + k, raw[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        is_last = atomicAdd(ticket, 1u) == (unsigned)(gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    for (int k = 0; k < 6; ++k) v[k] = 0;
    lt = INT64_MIN; best = lb_cand_none(0);
    if (tid < (int)gridDim.x) {
        // (agent-scope loads: the partials were written by other workgroups, possibly on another XCD's L2)
        const unsigned long long *q = (const unsigned long long *)&part[tid];
        unsigned long long raw[sizeof(ScanPartial) / 8];
        for (int k = 0; k < (int)(sizeof(ScanPartial) / 8); ++k) raw[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ScanPartial pp;
        __builtin_memcpy(&pp, raw, sizeof pp);
        for (int k = 0; k < 6; ++k) v[k] = pp.v[k];
        lt = pp.lt; best = pp.c;
    }
    __syncthreads();
    reduce_block();
    if (tid != 0) return;
    tot->ev[HS_EV_ENQUEUE] += v[0]; tot->ev[HS_EV_NOTIFY] += v[1]; tot->ev[HS_EV_POLL] += v[2];
    tot->ev[HS_EV_DELIVER] += v[3]; tot->ev[HS_EV_WORK] += v[3];
    tot->ev[HS_EV_CONTINUATION] += v[4]; tot->completed += v[4];
    tot->ev[HS_EV_SINK] += v[5]; tot->received += v[5];
    if (lt > tot->last_time) tot->last_time = lt;
    scan_cand[n_cand] = best;     // (one slot behind the workgroups' candidates)
    *ticket = 0;
}
// the dense completion log after hs_lbk_scan (+ hs_lbk_backends for the backends it handed back): a slot holds a record iff it
// is not the marker
struct SinkMark {
    const int64_t *sink_t;
    __device__ __forceinline__ bool operator()(int64_t i) const { return sink_t[i] != kSinkInvalid; }
};

// ---------------------------------------------------------------------------------------------
// 4. Shared Sink: slot i of the dense completion log is valid iff it lies in the used part of its backend's segment
// ---------------------------------------------------------------------------------------------
struct SinkValid {
    const uint64_t *skey; const int64_t *off; const int64_t *received; int tb; const LbTotals *tot; int B;
    __device__ __forceinline__ bool operator()(int64_t i) const {
        if (tot->use_t) {                                    // slot i = (completion m, backend b) of the [m][backend] logs
            if (i < (1ll << 32)) { const uint32_t m = (uint32_t)i / (uint32_t)B; return (int64_t)m < received[(uint32_t)i - m * (uint32_t)B]; }
            const int64_t m = i / B;
            return m < received[i - m * B];
        }
        const int64_t b = (int64_t)(skey[i] >> tb);
        return i - off[b] < received[b];
    }
};

// Completions with equal timestamps: the Sink processes them in the creation order of their ProcessContinuations,
// i.e. by service start (then backend; a backend's own completions are already in its order).  The radix sort is
// stable in slot order = (backend, completion order), so only runs of equal keys need a look.
// slot_bits > 0: the merged value is (created_at << slot_bits) | slot; slot_bits == 0: the value is the slot and created_at
// is gathered from the completion log.  The merge sorted on key bits [g, tb) only: a run of completions within the same
// 2^g ns is still in slot order; every element finds its place in its run by counting the run's elements that precede it
// in (completion ns, service start, slot) order.
__global__ void __launch_bounds__(256) hs_lb_sink_finish(const uint64_t *__restrict__ mkey, const uint64_t *__restrict__ mval, const int64_t *n_ptr,
                                                        const int64_t *__restrict__ sink_created, const int64_t *__restrict__ sink_S,
                                                        int64_t *__restrict__ out_t, int64_t *__restrict__ out_created, int slot_bits, int g) {
    __shared__ uint64_t lk[kSegTile + 2 * kSegHalo];         // the tile's keys + a halo (round 4: the walk along a run reads LDS)
    const int64_t n = *n_ptr;
    const int64_t base = (int64_t)blockIdx.x * kSegTile;
    if (base >= n) return;
    const int tid = threadIdx.x;
    for (int q = tid; q < kSegTile + 2 * kSegHalo; q += 256) {
        const int64_t i = base - kSegHalo + q;
        lk[q] = (i >= 0 && i < n) ? mkey[i] : 0ull;
    }
    __syncthreads();
    const int64_t lds_lo = base - kSegHalo, lds_hi = base + kSegTile + kSegHalo;
    auto key_at = [&](int64_t i) -> uint64_t { return (i >= lds_lo && i < lds_hi) ? lk[i - lds_lo] : mkey[i]; };
    const uint64_t smask = slot_bits ? ((1ull << slot_bits) - 1) : ~0ull;
#pragma unroll
    for (int r = 0; r < kSegTile / 256; ++r) {
        const int64_t i = base + tid + 256 * r;
        if (i >= n) continue;
        const uint64_t k = key_at(i), v = mval[i], slot = v & smask;
        const uint64_t hi = k >> g;
        int64_t lo = i, rank = 0, S_me = 0;
        bool haveS = false;
        // before(c): element c precedes this one: smaller time, or equal time and (earlier service start, then lower slot)
        auto before = [&](int64_t c) {
            const uint64_t kc = key_at(c);
            if (kc != k) return kc < k;
            if (!haveS) { S_me = sink_S[slot]; haveS = true; }
            const uint64_t sc = mval[c] & smask;
            const int64_t Sc = sink_S[sc];
            return Sc < S_me || (Sc == S_me && sc < slot);
        };
        while (lo > 0 && (key_at(lo - 1) >> g) == hi) { rank += before(lo - 1) ? 1 : 0; --lo; }
        for (int64_t j = i + 1; j < n && (key_at(j) >> g) == hi; ++j) rank += before(j) ? 1 : 0;
        const int64_t pos = lo + rank;
        out_t[pos] = (int64_t)k;
        out_created[pos] = slot_bits ? (int64_t)(v >> slot_bits) : sink_created[slot];
    }
}

// ---------------------------------------------------------------------------------------------
// 5. The one event beyond end_ns (core/simulation.py:472: the loop tests the PREVIOUS event's time) + totals
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// 5. Probes on a load-balancer graph (instrumentation/probe.py:81-164): Probe.on(server | sink, metric, interval).
// A probe is a daemon Source of its own whose ticks do not touch the simulation, so its tick times are a property of
// (interval, start) alone -- computed once, with the reference's own procedure (ConstantArrivalTimeProvider over
// _ProbeProfile: the general numerical path, hs_profile.hpp) -- and what it samples is a function of the run's logs:
//     stats_accepted(T) = #{admissions <= T}     completed(T) = #{completions <= T}     dropped = arrivals - accepted
//     started(T) = min(accepted, completed + c)  (work-conserving FIFO, c workers)      depth = accepted - started
//     active = started - completed               Sink.events_received(T) = #{records <= T}
// A sample on the very nanosecond of one of its target's events would need the reference's sort-index order between the
// probe's chain and the Request's chain: such a run is refused (probe_tie), never guessed.
// ---------------------------------------------------------------------------------------------
struct LbProbes {
    const int32_t *kind, *idx;    // [n] 0: backend Server idx;  1: Sink idx (shared Sink: 0; per-backend Sinks: backend);  2: Source idx
    const uint8_t *metric;        // [n] hs_probe_metric
    const double *rate;           // [n] 1.0 / interval
    int64_t *tick;                // [n][pcap] tick times up to the first one beyond the horizon
    int64_t *n_tick;              // [n] entries of tick[] (the last one lies beyond the horizon)
    int64_t *val;                 // [n][pcap] sampled values of the last run
    int64_t *cnt;                 // [n] samples of the last run (ticks <= end)
    LbCand *cand;                 // [n] the pending tick beyond end
    int64_t pcap;
    int n;
};

__device__ __forceinline__ int64_t count_le(const int64_t *a, int64_t n, int64_t stride, int64_t T) {   // #{a[i * stride] <= T}, a ascending
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (a[m * stride] <= T) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ int64_t count_lt(const int64_t *a, int64_t n, int64_t stride, int64_t T) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (a[m * stride] < T) lo = m + 1; else hi = m; }
    return lo;
}

// per probe: how many of its ticks lie at or before end (= samples of this run) and the pending tick beyond it
__global__ void hs_lb_probe_cands(LbProbes Q, int S, int B, int64_t end_ns, LbTotals *tot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Q.n) return;
    const int64_t *tk = Q.tick + (size_t)j * Q.pcap;
    const int64_t n = Q.n_tick[j];
    const int64_t c = count_le(tk, n, 1, end_ns);
    Q.cnt[j] = c;
    LbCand cd = lb_cand_none(S + B + j);
    cd.valid = c < n ? 1 : 0; cd.t = c < n ? tk[c] : kInfNs; cd.t_created = c > 0 ? tk[c - 1] : INT64_MIN;   // (constructed before run())
    cd.depth = c > 0 ? 1 : 0; cd.rcrt = c > 1 ? tk[c - 2] : INT64_MIN;    // a tick is created by the tick before it, always a root
    Q.cand[j] = cd;
    if (c > 0) {
        atomicAdd(&tot->ev[13], (unsigned long long)c);           // SourceEvent@Probe
        atomicAdd(&tot->ev[14], (unsigned long long)c);           // probe_event
        atomicMax(&tot->last_time, (long long)tk[c - 1]);
    }
}

// one lane per (probe, sample)
__global__ void hs_lb_probe_sample(LbProbes Q, LbBe PB, int B, int tb, const uint64_t *__restrict__ skey,
                                   const int64_t *__restrict__ off, const int64_t *__restrict__ adm,
                                   const int64_t *__restrict__ sink_t, const int64_t *__restrict__ out_t,
                                   const int64_t *n_done, int shared_sink, int phase, LbTotals *tot,
                                   const uint64_t *__restrict__ keys0, const int64_t *__restrict__ src_count, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(i / Q.pcap);
    const int64_t k = i - (int64_t)j * Q.pcap;
    if (j >= Q.n || k >= Q.cnt[j]) return;
    if ((Q.kind[j] == 1 && shared_sink) != (phase == 1)) return;  // phase 1: probes on the shared Sink (after its merge)
    const int64_t T = Q.tick[(size_t)j * Q.pcap + k];
    const uint32_t m = Q.metric[j];
    int64_t v = 0;
    bool tie = false;
    if (Q.kind[j] == 2) {                                         // Source.generated_count: its column of the [tick][source] log
        const int sidx = Q.idx[j];                                // (every tick carries a Request: Sources with stop_after are refused)
        const uint64_t tmask = tb >= 64 ? ~0ull : ((1ull << tb) - 1);
        const int64_t n = src_count[sidx];
        int64_t lo = 0, hi = n, lt = 0;
        while (lo < hi) { const int64_t q = (lo + hi) >> 1; if ((int64_t)(keys0[(size_t)q * S + sidx] & tmask) <= T) lo = q + 1; else hi = q; }
        hi = n;
        while (lt < hi) { const int64_t q = (lt + hi) >> 1; if ((int64_t)(keys0[(size_t)q * S + sidx] & tmask) < T) lt = q + 1; else hi = q; }
        v = lo;
        tie = lo != lt;
    } else if (Q.kind[j] == 1 && shared_sink) {                   // the shared Sink: the merged completion log
        const int64_t n = *n_done;
        v = count_le(out_t, n, 1, T);
        tie = v != count_lt(out_t, n, 1, T);
    } else {
        const int b = Q.idx[j];
        const int64_t o = off[b], na = off[b + 1] - o;
        const bool Tl = tot->use_t != 0;                          // [m][backend] completion logs this run?
        const int64_t *st = sink_t + (Tl ? b : o);
        const int64_t ss = Tl ? B : 1;
        const int64_t ndep = PB.received[b];                      // completions at or before end that reached the log
        const int64_t done = count_le(st, ndep, ss, T);
        tie = done != count_lt(st, ndep, ss, T);
        if (Q.kind[j] == 1) v = done;                             // a backend's own Sink
        else {
            // arrivals of the backend at or before T: its segment of the sorted (backend << tb | arrival ns) keys
            const uint64_t tmask = tb >= 64 ? ~0ull : ((1ull << tb) - 1);
            int64_t lo = 0, hi = na, lt = 0;
            while (lo < hi) { const int64_t q = (lo + hi) >> 1; if ((int64_t)(skey[o + q] & tmask) <= T) lo = q + 1; else hi = q; }
            const int64_t arrived = lo;
            hi = na;
            while (lt < hi) { const int64_t q = (lt + hi) >> 1; if ((int64_t)(skey[o + q] & tmask) < T) lt = q + 1; else hi = q; }
            tie = tie || arrived != lt;
            int64_t accepted = arrived;
            if (PB.qcap[b] >= 0) {                                // bounded queue: the admission log holds the accepted ones
                accepted = count_le(adm + o, PB.accepted[b], 1, T);
            }
            int64_t started = done + PB.conc[b];
            started = accepted < started ? accepted : started;
            if (PB.rejected[b] > 0) atomicOr(&tot->probe_tie, 2);     // (deliveries that found no free worker: not work-conserving)
            switch (m) {
                case HS_PROBE_DEPTH: v = accepted - started; break;
                case HS_PROBE_ACTIVE: v = started - done; break;
                case HS_PROBE_ACCEPTED: v = accepted; break;
                case HS_PROBE_DROPPED: v = arrived - accepted; break;
                default: v = done; break;                        // HS_PROBE_COMPLETED
            }
        }
    }
    Q.val[(size_t)j * Q.pcap + k] = v;
    if (tie) atomicOr(&tot->probe_tie, 1);
}

__global__ void __launch_bounds__(kLbBlock) hs_lb_finalize(LbSrc PS, LbBe PB, int S, int B, int64_t start_ns, LbTotals *tot,
                                                          LbProbes Q, int nbs, int nbb, const LbCand *__restrict__ scan_cand, int nsc) {
    __shared__ LbCand wc[kLbBlock / 64];
    const int tid = threadIdx.x;
    LbCand best = lb_cand_none(0);
    // (nbs, nbb, nsc: the workgroups of hs_lbk_sources / hs_lbk_backends / hs_lbk_scan -- one candidate each)
    for (int i = tid; i < nbs + nbb + Q.n + nsc; i += kLbBlock) {
        const LbCand c = i < nbs ? PS.cand[i] : i < nbs + nbb ? PB.cand[i - nbs] : i < nbs + nbb + Q.n ? Q.cand[i - nbs - nbb]
                                                                                 : scan_cand[i - nbs - nbb - Q.n];
        if (cand_before(c, best)) best = c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const LbCand d = cand_shfl_xor(best, o);
        if (cand_before(d, best)) best = d;
    }
    if ((tid & 63) == 0) wc[tid >> 6] = best;
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < kLbBlock / 64; ++w) if (cand_before(wc[w], best)) best = wc[w];
    long long fin = tot->last_time == INT64_MIN ? start_ns : tot->last_time;
    if (best.valid) {
        if (best.idx >= S + B) {                  // a Probe's SourceEvent: counted, its probe_event is not processed
            tot->ev[13] += 1;
        } else if (best.idx < S) {                // SourceEvent: the tick is counted, its Request is not processed
            PS.generated[best.idx] += 1;
            tot->ev[HS_EV_SOURCE] += 1;
        } else {                                  // ProcessContinuation: the Server's statistics move, the Sink's do not
            const int b = best.idx - S;
            PB.completed[b] += 1;
            PB.active[b] = PB.active[b] > 0 ? PB.active[b] - 1 : 0;
            PB.total_service[b] = __dadd_rn(PB.total_service[b], best.svc_s);
            tot->ev[HS_EV_CONTINUATION] += 1;
            tot->completed += 1;
        }
        fin = best.t;
    }
    tot->final_time = fin;
}

// ---------------------------------------------------------------------------------------------
// Sink.latency_stats() of the shared Sink on the device (components/common.py:59-76, instrumentation/data.py:197-210):
// latencies (completion ns - created_at ns; latency_s = float(ns) / 1e9 is monotone in ns) are radix-sorted as integers,
// then count / avg / min / max / p50 / p99 follow the reference's formulas: avg = sum(sorted) / n with the sum taken
// left to right in binary64 (CPython < 3.12 `sum`), percentile p = s[lo] * (1 - frac) + s[hi] * frac at pos = p * (n - 1).
// ---------------------------------------------------------------------------------------------
__global__ void hs_lb_latency_keys(const int64_t *__restrict__ t, const int64_t *__restrict__ created, const int64_t *n_ptr,
                                   uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *n_ptr) keys[i] = (uint64_t)(t[i] - created[i]);
}
__device__ __forceinline__ double percentile_sorted_ns(const uint64_t *s, int64_t n, double p) {
    const double pos = __dmul_rn(p, (double)(n - 1));
    const int64_t lo = (int64_t)pos;
    const int64_t hi = lo + 1 < n - 1 ? lo + 1 : n - 1;
    const double frac = __dsub_rn(pos, (double)lo);
    const double a = seconds_from_ns_ieee((int64_t)s[lo]), b = seconds_from_ns_ieee((int64_t)s[hi]);
    return __dadd_rn(__dmul_rn(a, __dsub_rn(1.0, frac)), __dmul_rn(b, frac));
}
static int g_float_sum_mode = 0;     // hs_set_float_sum_mode: 0 left to right, 1 Neumaier (what the host interpreter's `sum` does)
// `sum(sorted_vals)` as the interpreter that runs the reference computes it: CPython's float fast path (Python/bltinmodule.c
// builtin_sum) is a plain left-to-right binary64 sum before 3.12 and Neumaier's compensated sum from 3.12 on (the reference
// requires >= 3.13, pyproject.toml:11) -- `compensated` says which; hs_set_float_sum_mode, set by the host side from sys.version_info.
__global__ void __launch_bounds__(64) hs_lb_latency_stats_kernel(const uint64_t *__restrict__ sorted, const int64_t *n_ptr,
                                                                 double *__restrict__ out, int compensated) {
    const int64_t n = *n_ptr;
    const int lane = threadIdx.x;
    if (n <= 0) { if (lane < 6) out[lane] = 0.0; return; }
    // sequential sum: the wavefront converts 64 values at a time, then they are added in index order
    double sum = 0.0, comp = 0.0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        const double v = i < n ? seconds_from_ns_ieee((int64_t)sorted[i]) : 0.0;
        const int m = (n - base) < 64 ? (int)(n - base) : 64;
        if (!compensated) { for (int j = 0; j < m; ++j) sum = __dadd_rn(sum, __shfl(v, j, 64)); }
        else {
            for (int j = 0; j < m; ++j) {                 // t = sum + x; c += |sum| >= |x| ? (sum - t) + x : (x - t) + sum; sum = t
                const double x = __shfl(v, j, 64), t = __dadd_rn(sum, x);
                comp = __dadd_rn(comp, __builtin_fabs(sum) >= __builtin_fabs(x) ? __dadd_rn(__dsub_rn(sum, t), x) : __dadd_rn(__dsub_rn(x, t), sum));
                sum = t;
            }
        }
    }
    if (compensated && comp != 0.0 && __builtin_isfinite(comp)) sum = __dadd_rn(sum, comp);
    if (lane == 0) {
        out[0] = (double)n;
        out[1] = __ddiv_rn(sum, (double)n);
        out[2] = seconds_from_ns_ieee((int64_t)sorted[0]);
        out[3] = seconds_from_ns_ieee((int64_t)sorted[n - 1]);
        out[4] = percentile_sorted_ns(sorted, n, 0.50);
        out[5] = percentile_sorted_ns(sorted, n, 0.99);
    }
}

__global__ void hs_lb_clear(LbTotals *tot) {
    for (int k = 0; k < HS_EV_KINDS; ++k) tot->ev[k] = 0;
    tot->completed = 0; tot->received = 0; tot->last_time = INT64_MIN; tot->final_time = 0;
    tot->qoverflow = 0; tot->bad_client = 0; tot->max_count = 0; tot->max_be = 0; tot->use_t = 0; tot->probe_tie = 0; tot->src_redo = 0;
}

using hs::ring::Md5; using hs::ring::RingPoint; using hs::ring::md5_u128; using hs::ring::ring_select;   // (hs_ring.hpp)

thread_local std::string g_lb_error;

}  // namespace

struct hs_lb {
    hs_lb_config cfg{};
    int C = 1, tb = 1, bb = 1, slot_bits = 0;
    int g_arr = 0, g_sink = 0;     // low key bits the two sorts skip (fixed up afterwards inside short runs)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evs0 = nullptr, evs1 = nullptr, evs2 = nullptr, evs3 = nullptr;
    std::vector<void *> allocs;
    std::vector<RingPoint> ring;
    LbSrc PS{};
    LbBe PB{};
    LbTotals *tot = nullptr;
    int32_t *client_be = nullptr;
    int64_t n_table = 0, cap = 0, n_slots = 0;
    uint64_t *keys0 = nullptr, *vals0 = nullptr;      // [cap][S] arrival logs
    int64_t n_pre = 0;                                 // ticks per Source whose stream values hs_lb_source_draws produces
    bool any_stop = false;                             // some Source has stop_after
    int src_lanes_run = 64;                            // Sources per wavefront of the last run's Source kernel (hs_lb_finalize reads one candidate per workgroup)
    bool lean_off = false, lean_ran = false;           // hs_lbk_sources_lean: switched off after a run it could not cover; used by the last run
    int n_simd = 1024;                                 // SIMDs of the device (4 per CU): lb_lanes()
    bool f64_times = false;                            // every time of a run is a whole ns in [0, 2^51): exact in binary64
    uint64_t *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr;   // dense ping-pong buffers [n_slots]
    uint64_t *skey = nullptr, *sval = nullptr;        // where the sorted arrivals ended up
    uint64_t *mkey = nullptr, *mslot = nullptr;       // where the merged Sink order ended up
    int64_t *off = nullptr, *adm = nullptr, *sink_t = nullptr, *sink_created = nullptr, *sink_S = nullptr;
    int64_t *out_t = nullptr, *out_created = nullptr;
    LbLayout LY{};                                    // [k][backend] copies of the backend streams (rows == 0: not allocated)
    int64_t n_layout = 0;                             // max(n_slots, rows * B): slots of the completion logs / draw grid
    int64_t *n_merge = nullptr;                       // device: slots the Sink merge scans
    double *svdraw = nullptr;                         // [n_slots] service sample per Request slot (single-worker FIFO backends)
    // hs_lbk_scan (round 4): per-backend event counts [3][B], last event time [B], "run me in event order" [B], one candidate per workgroup
    uint32_t *bev = nullptr; int64_t *blast = nullptr; uint8_t *redo = nullptr; LbCand *scan_cand = nullptr;
    ScanPartial *scan_part = nullptr; unsigned *scan_ticket = nullptr;
    bool any_simple = false;
    bool any_src_profile = false;                      // some Source has a time-varying profile (hs_lbk_sources<true>)
    // ... their tick tables (hs_tables.hpp), built once on the first run
    TickRow *tab_rows = nullptr; int n_tab_rows = 0; int64_t *tab_times = nullptr, *tab_count = nullptr;
    unsigned long long *tab_status = nullptr; bool tables_built = false;
    long long lane_budget = kDefaultLaneBudget;
    bool any_no_sink = false;                          // some backend has no Sink behind it: no completion log (probes refused)
    int64_t *n_slots_dev = nullptr, *n_arr = nullptr, *n_done = nullptr, *n_tmp = nullptr;
    uint32_t *hist = nullptr, *row_total = nullptr, *digit_base = nullptr;
    uint32_t *ghist = nullptr, *tickets = nullptr;     // hs_radix.hpp round 3: digit histograms of all passes, tile tickets per pass
    int *radix_err = nullptr;                          // a look-back gave up (bounded spin)
    int n_tiles = 0;
    LbProbes Q{};                                     // probes (hs_lb_set_probes); n == 0: none
    std::vector<int64_t> src_stop_h;                  // stop_after of every Source (host copy: probes on stopping Sources are refused)
    bool ran = false;
    int flags = 0;
    double last_run_ms = 0.0, last_sort_ms = 0.0;
    int64_t launches = 0;
    std::string error;
};

namespace {

int lfail(hs_lb *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf;
    g_lb_error = buf;
    return code;
}

#define LB_HIP(h, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return lfail(h, HS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

template <typename T>
int lalloc(hs_lb *h, T **p, size_t count) {
    void *q = nullptr;
    const size_t bytes = count * sizeof(T) ? count * sizeof(T) : sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return lfail(h, HS_E_HIP, "hipMalloc(%zu B): %s", bytes, hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return HS_OK;
}
template <typename T>
int lupload(hs_lb *h, const T **dst, const T *src, size_t n, T dflt) {
    T *d = nullptr;
    int rc = lalloc(h, &d, n);
    if (rc) return rc;
    std::vector<T> tmp;
    if (!src) { tmp.assign(n, dflt); src = tmp.data(); }
    if (hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return lfail(h, HS_E_HIP, "hipMemcpy H2D failed");
    *dst = d;
    return HS_OK;
}

int bit_length(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b < 1 ? 1 : b; }


// One stable LSD sort over key bits [0, bits): pass 0 reads (k_in, v_in) through `valid` / `mk`, later passes
// ping-pong between (kA, vA) and (kB, vB).  n_in_dev = slots of pass 0, n_out_dev receives the valid count.
template <typename Valid, typename MakeVal>
void radix_sort_async(hs_lb *h, const uint64_t *k_in, const uint64_t *v_in, const int64_t *n_in_dev, int64_t *n_out_dev,
                      int bits, Valid valid, MakeVal mk, uint64_t **k_res, uint64_t **v_res,
                      const uint64_t *keep_through_pass0 = nullptr, int shift0 = 0, int64_t slots_pass0 = 0) {
    const int passes = (bits - shift0 + kRadixBits - 1) / kRadixBits;
    // pass 0 may scan a sparse input (slots_pass0 of them); every later pass sees at most n_slots dense elements
    const int tiles0 = slots_pass0 > 0 ? (int)((slots_pass0 + kRadixTile - 1) / kRadixTile) : h->n_tiles;
    const int tiles_rest = (int)((h->n_slots + kRadixTile - 1) / kRadixTile);
    const dim3 blk(kRadixThreads);
    // `valid` may read a buffer that is itself one of the ping-pong buffers (the previous sort's result): pass 0, the
    // only pass that evaluates `valid`, must then write the OTHER buffer
    const bool startB = keep_through_pass0 == h->kA;
    uint64_t *ko = startB ? h->kB : h->kA, *vo = startB ? h->vB : h->vA;
    const uint64_t *ki = k_in, *vi = v_in;
    if (h->ghist == nullptr) {       // (the one-off sorts of hs_debug_radix_sort / hs_merge_sink_records / the latency statistics)
        if (lalloc(h, &h->ghist, (size_t)kRadixMaxPasses * kRadixBins) || lalloc(h, &h->tickets, (size_t)kRadixMaxPasses + 1) ||
            lalloc(h, &h->radix_err, (size_t)1)) h->ghist = nullptr;
        else { hipMemset(h->tickets, 0, (kRadixMaxPasses + 1) * sizeof(uint32_t)); hipMemset(h->radix_err, 0, sizeof(int)); }
    }
    if ((h->flags & 16) != 0 && passes <= kRadixMaxPasses && h->ghist != nullptr) {
        // debug flag 16 (hs_radix.hpp, round 3): one histogram read for all passes, then one look-back scatter per pass.  Measured on
        // MI355X at the configs[4] size: a dense pass 162 us instead of 101 + 35 (histogram) + 23 (scans) -- the tiles of a launch
        // start together, so a tile's look-back walks hundreds of aggregates, and every descriptor load is a device-scope access
        // that leaves its XCD's L2 (8 XCDs, one L2 each): ~1 us per window of 16.  The three-kernels-per-pass sort stays the default.
        hipMemsetAsync(h->ghist, 0, (size_t)kRadixMaxPasses * kRadixBins * sizeof(uint32_t), h->stream);
        const int hist_blocks = tiles0 < 1024 ? tiles0 : 1024;
        hipLaunchKernelGGL((radix_hist_all<Valid>), dim3((unsigned)hist_blocks), blk, 0, h->stream, ki, n_in_dev, shift0, passes, h->ghist, valid);
        hipLaunchKernelGGL(radix_digit_bases, dim3(1), blk, 0, h->stream, h->ghist, passes, n_out_dev, h->tickets);
        h->launches += 2;
        for (int p = 0; p < passes; ++p) {
            const int shift = shift0 + p * kRadixBits;
            const int64_t *n_dev = p == 0 ? n_in_dev : n_out_dev;
            const int nt = p == 0 ? tiles0 : (tiles_rest < h->n_tiles ? tiles_rest : h->n_tiles);
            hipMemsetAsync(h->hist, 0, (size_t)nt * kRadixBins * sizeof(uint32_t), h->stream);     // the pass's tile descriptors
            if (p == 0)
                hipLaunchKernelGGL((radix_scatter_lb<Valid, MakeVal>), dim3((unsigned)nt), blk, 0, h->stream, ki, vi, ko, vo, n_dev, shift,
                                   h->ghist + p * kRadixBins, h->hist, h->tickets + p, h->radix_err, valid, mk);
            else
                hipLaunchKernelGGL((radix_scatter_lb<RadixAll, NoVal>), dim3((unsigned)nt), blk, 0, h->stream, ki, vi, ko, vo, n_dev, shift,
                                   h->ghist + p * kRadixBins, h->hist, h->tickets + p, h->radix_err, RadixAll{}, NoVal{});
            h->launches += 2;
            ki = ko; vi = vo;
            if (ko == h->kA) { ko = h->kB; vo = h->vB; } else { ko = h->kA; vo = h->vA; }
        }
        *k_res = const_cast<uint64_t *>(ki);
        *v_res = const_cast<uint64_t *>(vi);
        return;
    }
    for (int p = 0; p < passes; ++p) {
        const int shift = shift0 + p * kRadixBits;
        const int64_t *n_dev = p == 0 ? n_in_dev : n_out_dev;
        const int nt = p == 0 ? tiles0 : (tiles_rest < h->n_tiles ? tiles_rest : h->n_tiles);
        const dim3 grid((unsigned)nt);
        if (p == 0) {
            hipLaunchKernelGGL((radix_hist<Valid>), grid, blk, 0, h->stream, ki, n_dev, shift, h->hist, nt, valid);
        } else {
            hipLaunchKernelGGL((radix_hist<RadixAll>), grid, blk, 0, h->stream, ki, n_dev, shift, h->hist, nt, RadixAll{});
        }
        hipLaunchKernelGGL(radix_scan_rows, dim3(kRadixBins), blk, 0, h->stream, h->hist, nt, h->row_total, h->digit_base,
                           p == 0 ? n_out_dev : (int64_t *)nullptr, h->tickets + kRadixMaxPasses);
        if (p == 0) {
            hipLaunchKernelGGL((radix_scatter<Valid, MakeVal>), grid, blk, 0, h->stream, ki, vi, ko, vo, n_dev, shift, h->hist,
                               h->digit_base, nt, valid, mk);
        } else {
            hipLaunchKernelGGL((radix_scatter<RadixAll, NoVal>), grid, blk, 0, h->stream, ki, vi, ko, vo, n_dev, shift, h->hist,
                               h->digit_base, nt, RadixAll{}, NoVal{});
        }
        h->launches += 3;
        ki = ko; vi = vo;
        if (ko == h->kA) { ko = h->kB; vo = h->vB; } else { ko = h->kA; vo = h->vA; }
    }
    *k_res = const_cast<uint64_t *>(ki);
    *v_res = const_cast<uint64_t *>(vi);
}

// LPs (Sources, backends) per wavefront of the one-LP-per-lane kernels.
int lb_lanes(const hs_lb *h, int n) {
    (void)n;
    return (h->flags & 32) ? 32 : 64;     // (measured: a per-LP serial chain does not get shorter on more SIMDs; debug flag 32 keeps the mapping tested)
}

template <int C>
void launch_backends(hs_lb *h, int64_t end_ns, int flags, const uint8_t *only = nullptr, int pack_bits = 0) {
    const int B = h->cfg.n_backends;
    const int lanes = lb_lanes(h, B);
    const int per_block = lanes * (kLbBlock / 64);
    hipLaunchKernelGGL(hs_lbk_backends<C>, dim3((B + per_block - 1) / per_block), dim3(kLbBlock), 0, h->stream, h->PB, B,
                       h->cfg.n_sources, h->cfg.seed, h->cfg.start_ns, end_ns, h->skey, h->sval, h->off, h->tb, h->adm,
                       h->sink_t, h->sink_created, h->sink_S, h->svdraw, h->tot, flags, h->LY, lanes, only, pack_bits);
}
// Every backend has one worker and no debug flag asks for a particular legacy path: the backends run as segmented (max, +) scans over
// the dense sorted arrival list (hs_lbk_scan), no [k][backend] layout, no separate service draws; debug flag 64 keeps round 3's pipeline.
bool scan_path(const hs_lb *h) { return h->C == 1 && (h->flags & (1 | 2 | 4 | 64)) == 0; }

int run_async(hs_lb *h, int64_t end_ns) {
    const int S = h->cfg.n_sources, B = h->cfg.n_backends;
    h->launches = 0;
    hipLaunchKernelGGL(hs_lb_clear, dim3(1), dim3(1), 0, h->stream, h->tot);
    if (h->any_src_profile && !h->tables_built) {      // the time-varying Sources' tick tables: once (hs_tables.hpp)
        LB_HIP(h, tick_tables_launch(h->stream, h->tab_rows, h->n_tab_rows, h->cfg.start_ns, h->cfg.horizon_ns, h->PS.tab_cap,
                                     h->tab_times, h->tab_count, h->tab_status, h->lane_budget, false));
        h->tables_built = true;
    }
    // the Sources' stream values first, with every SIMD (hs_lb_source_draws; the sort's ping-pong buffers are free until the sort);
    // debug flag 8: the Sources draw their own
    const int64_t n_pre = (h->flags & 8) ? 0 : h->n_pre;
    // speculated arrival steps (lb_step_encode): constant-rate Sources on the exact-binary64 path, horizons below 2^40 ns; debug flag 256 off
    const double margin = (h->f64_times && !h->any_src_profile && h->cfg.horizon_ns < (1ll << 40) && (h->flags & 256) == 0) ? 1.0 / 1024.0 : 0.0;
    double *dinc = (double *)h->kA;
    int32_t *dbe = (int32_t *)h->vA;
    if (n_pre > 0) {
        const int64_t threads = ((n_pre + 1) / 2) * (int64_t)S;
        hipLaunchKernelGGL(hs_lb_source_draws, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, h->PS, S, h->cfg.seed,
                           h->client_be, h->n_table, n_pre, dinc, dbe, margin);
        h->launches += 1;
    }
    const int src_lanes = lb_lanes(h, S), src_per_block = src_lanes * (kLbBlock / 64);
#define HS_LAUNCH_SOURCES(PFV, F64V) hipLaunchKernelGGL((hs_lbk_sources<PFV, F64V>), dim3((S + src_per_block - 1) / src_per_block), dim3(kLbBlock), 0, h->stream, \
        h->PS, S, h->cfg.seed, h->cfg.start_ns, end_ns, h->client_be, h->n_table, h->keys0, h->vals0, h->cap, h->tb, h->tot, dinc, dbe, n_pre, src_lanes, margin)
    // the common run (constant rates, speculated whole-ns steps, no stop_after, every value pre-drawn): hs_lbk_sources_lean; debug flag
    // 512 keeps hs_lbk_sources, and so does an engine whose last run raised LbTotals::src_redo
    const int64_t lean_rows = (n_pre < h->cap ? n_pre : h->cap) & ~(int64_t)15;
    h->lean_ran = !h->lean_off && (h->flags & 512) == 0 && margin > 0.0 && !h->any_stop && lean_rows >= 16 &&
                  (double)lean_rows * (double)S * 8.0 < 4.0e9;
    h->src_lanes_run = src_lanes;
    if (h->lean_ran) {
        // (32 Sources per wavefront -- twice the wavefronts, twice the loads in flight -- measured: 138 us against 120)
        const int ll = src_lanes, lpb = ll * (kLbBlock / 64);
        h->src_lanes_run = ll;
        hipLaunchKernelGGL(hs_lbk_sources_lean, dim3((S + lpb - 1) / lpb), dim3(kLbBlock), 0, h->stream, h->PS, S, h->cfg.start_ns,
                           end_ns, h->keys0, h->vals0, lean_rows, h->tb, h->tot, dinc, dbe, ll, (h->flags & 1024) ? 1 : 0);
    }
    else if (h->any_src_profile) HS_LAUNCH_SOURCES(true, false);
    else if (h->f64_times) HS_LAUNCH_SOURCES(false, true);
    else HS_LAUNCH_SOURCES(false, false);
#undef HS_LAUNCH_SOURCES
    hipLaunchKernelGGL(hs_lb_rows, dim3(1), dim3(1), 0, h->stream, h->tot, S, h->n_slots_dev);
    hipEventRecord(h->evs0, h->stream);
    if (h->cfg.strategy == HS_LB_ROUND_ROBIN) {
        // the LoadBalancer's processing order first: all Requests by arrival ns (backend column 0), every Request's place in that
        // order, backend = place mod B (hs_lb_rr_assign); then the (backend, ns) sort of any other strategy over the dense result
        uint64_t *tk = nullptr, *tv = nullptr;
        radix_sort_async(h, h->keys0, h->vals0, h->n_slots_dev, h->n_arr, h->tb, TickValid{h->PS.count, S, h->n_slots < (1ll << 31)},
                         NoVal{}, &tk, &tv, nullptr, h->g_sink, h->n_slots);
        uint64_t *fk = tk == h->kA ? h->kB : h->kA, *fv = tk == h->kA ? h->vB : h->vA;
        hipLaunchKernelGGL(hs_lb_rr_assign, dim3((unsigned)((h->n_slots + kSegTile - 1) / kSegTile)), dim3(256), 0, h->stream, tk, tv, fk, fv,
                           h->n_arr, h->tb, h->g_sink, B);
        h->launches += 1;
        radix_sort_async(h, fk, fv, h->n_arr, h->n_arr, h->tb + h->bb, RadixAll{}, NoVal{}, &h->skey, &h->sval, fk, h->g_arr, h->n_slots);
    } else
        radix_sort_async(h, h->keys0, h->vals0, h->n_slots_dev, h->n_arr, h->tb + h->bb,
                         TickValid{h->PS.count, S, h->n_slots < (1ll << 31)}, NoVal{}, &h->skey, &h->sval, nullptr, h->g_arr,
                         h->n_slots);
    hipEventRecord(h->evs1, h->stream);
    {   // segment offsets + full-key order inside the runs the sort left, written to the other ping-pong buffer
        uint64_t *fk = h->skey == h->kA ? h->kB : h->kA, *fv = h->skey == h->kA ? h->vB : h->vA;
        hipLaunchKernelGGL(hs_lb_segments, dim3((unsigned)((h->n_slots + 1 + kSegTile - 1) / kSegTile)), dim3(256), 0, h->stream, h->skey,
                           h->sval, fk, fv, h->n_arr, h->tb, h->g_arr, B, h->off);
        h->skey = fk; h->sval = fv;
    }
    const bool scan = scan_path(h);
    const int n_scan_blocks = (B + kLbBlock / 64 - 1) / (kLbBlock / 64);
    const int pack_bits = (scan && h->cfg.shared_sink) ? h->slot_bits : 0;
    if (scan) {
        // one wavefront per backend over its dense segment: service draws, the (max, +) scan, counts, completion records (section 3b);
        // what it hands back (bounded queues, a zero-nanosecond service) runs in event order on the dense layout
        if (h->f64_times && (h->flags & 128) == 0)      // (debug flag 128: the int64 instantiation)
            hipLaunchKernelGGL(hs_lbk_scan<true>, dim3((unsigned)n_scan_blocks), dim3(kLbBlock), 0, h->stream, h->PB, B, S, h->cfg.seed, end_ns, h->skey,
                               h->sval, h->off, h->tb, h->sink_t, h->sink_created, h->sink_S, h->bev, h->blast, h->redo, h->scan_cand, pack_bits);
        else
            hipLaunchKernelGGL(hs_lbk_scan<false>, dim3((unsigned)n_scan_blocks), dim3(kLbBlock), 0, h->stream, h->PB, B, S, h->cfg.seed, end_ns, h->skey,
                               h->sval, h->off, h->tb, h->sink_t, h->sink_created, h->sink_S, h->bev, h->blast, h->redo, h->scan_cand, pack_bits);
        // (the totals kernel runs in the stream BEFORE the handed-back backends add theirs with atomics: plain read-modify-writes are safe)
        hipLaunchKernelGGL(hs_lb_scan_totals, dim3(kScanParts), dim3(256), 0, h->stream, h->PB, B, h->bev, h->blast, h->redo, h->tot, h->scan_cand,
                           n_scan_blocks, h->scan_part, h->scan_ticket);
        launch_backends<1>(h, end_ns, h->flags | 2, h->redo, pack_bits);
        h->launches += 3;
    } else {
        // layout of the backend streams for this run: [k][backend] when the busiest backend fits the allocated rows
        hipLaunchKernelGGL(hs_lb_maxcount, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->off, B, h->tot);
        hipLaunchKernelGGL(hs_lb_layout, dim3(1), dim3(1), 0, h->stream, h->tot, h->LY, B, h->n_arr, h->n_merge,
                           (h->flags & 4) ? 1 : 0);
        if (h->LY.rows > 0 && (h->flags & 4) == 0)
            hipLaunchKernelGGL(hs_lb_transpose, dim3((B + 63) / 64), dim3(256), 0, h->stream, h->skey, h->off, B, h->LY, h->tot);
        if (h->C == 1 && h->any_simple && (h->flags & 3) == 0)
            hipLaunchKernelGGL(hs_lb_service_draws, dim3((unsigned)((h->n_layout + 255) / 256)), dim3(256), 0, h->stream, h->skey,
                               h->n_arr, h->off, h->tb, h->PB, h->cfg.seed, h->svdraw, h->LY, B, h->tot);
        h->launches += 4;
        switch (h->C) {
            case 1: launch_backends<1>(h, end_ns, h->flags); break;
            case 2: launch_backends<2>(h, end_ns, h->flags); break;
            case 4: launch_backends<4>(h, end_ns, h->flags); break;
            case 8: launch_backends<8>(h, end_ns, h->flags); break;
            case 16: launch_backends<16>(h, end_ns, h->flags); break;
            default: launch_backends<32>(h, end_ns, h->flags); break;    // (the 32 slots' departure times do not fit the register file: they live in scratch)
        }
    }
    if (h->Q.n > 0) {     // probes: the ticks of this run and the pending one beyond end; then the samples of everything but a
                          // shared Sink, from the backends' logs (the Sink merge below reuses the buffer of the sorted arrivals)
        hipLaunchKernelGGL(hs_lb_probe_cands, dim3((h->Q.n + 63) / 64), dim3(64), 0, h->stream, h->Q, S, B, end_ns, h->tot);
        const int64_t lanes = (int64_t)h->Q.n * h->Q.pcap;
        hipLaunchKernelGGL(hs_lb_probe_sample, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, h->stream, h->Q, h->PB, B, h->tb,
                           h->skey, h->off, h->adm, h->sink_t, h->out_t, h->n_done, h->cfg.shared_sink ? 1 : 0, 0, h->tot, h->keys0, h->PS.count, S);
        h->launches += 2;
    }
    hipEventRecord(h->evs2, h->stream);
    if (h->cfg.shared_sink) {
        // completions by completion ns; the validity functor reads the sorted arrival keys (which slot belongs to which
        // backend), so pass 0 must not overwrite them
        if (scan && h->slot_bits)        // the dense completion log of hs_lbk_scan: a slot holds a record iff it is not the marker;
                                         // sink_created already holds the merge's value (created_at << slot_bits | slot)
            radix_sort_async(h, (const uint64_t *)h->sink_t, (const uint64_t *)h->sink_created, h->n_arr, h->n_done, h->tb, SinkMark{h->sink_t},
                             NoVal{}, &h->mkey, &h->mslot, h->skey, h->g_sink, h->n_slots);
        else if (scan)
            radix_sort_async(h, (const uint64_t *)h->sink_t, (const uint64_t *)nullptr, h->n_arr, h->n_done, h->tb, SinkMark{h->sink_t},
                             SlotVal{}, &h->mkey, &h->mslot, h->skey, h->g_sink, h->n_slots);
        else if (h->slot_bits)
            radix_sort_async(h, (const uint64_t *)h->sink_t, (const uint64_t *)nullptr, h->n_merge, h->n_done, h->tb,
                             SinkValid{h->skey, h->off, h->PB.received, h->tb, h->tot, B},
                             PackCreatedSlot{h->sink_created, h->slot_bits}, &h->mkey, &h->mslot, h->skey, h->g_sink, h->n_layout);
        else
            radix_sort_async(h, (const uint64_t *)h->sink_t, (const uint64_t *)nullptr, h->n_merge, h->n_done, h->tb,
                             SinkValid{h->skey, h->off, h->PB.received, h->tb, h->tot, B}, SlotVal{}, &h->mkey, &h->mslot,
                             h->skey, h->g_sink, h->n_layout);
        hipLaunchKernelGGL(hs_lb_sink_finish, dim3((unsigned)((h->n_slots + kSegTile - 1) / kSegTile)), dim3(256), 0, h->stream, h->mkey,
                           h->mslot, h->n_done, h->sink_created, h->sink_S, h->out_t, h->out_created, h->slot_bits, h->g_sink);
    }
    hipEventRecord(h->evs3, h->stream);
    if (h->Q.n > 0 && h->cfg.shared_sink) {   // probes on the shared Sink: its merged log exists now
        const int64_t lanes = (int64_t)h->Q.n * h->Q.pcap;
        hipLaunchKernelGGL(hs_lb_probe_sample, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, h->stream, h->Q, h->PB, B, h->tb,
                           h->skey, h->off, h->adm, h->sink_t, h->out_t, h->n_done, 1, 1, h->tot, h->keys0, h->PS.count, S);
        h->launches += 1;
    }
    {
        const int sl = h->src_lanes_run * (kLbBlock / 64), bl = lb_lanes(h, B) * (kLbBlock / 64);
        hipLaunchKernelGGL(hs_lb_finalize, dim3(1), dim3(kLbBlock), 0, h->stream, h->PS, h->PB, S, B, h->cfg.start_ns, h->tot, h->Q,
                           (S + sl - 1) / sl, (B + bl - 1) / bl, h->scan_cand + n_scan_blocks, scan ? 1 : 0);
    }
    h->launches += 6;
    LB_HIP(h, hipGetLastError());
    h->ran = true;
    return HS_OK;
}

// hs_lbk_sources_lean left a Source to hs_lbk_sources (LbTotals::src_redo): the caller repeats the run with that kernel
bool lean_gave_up(hs_lb *h) {
    if (!h->lean_ran) return false;
    int redo = 0;
    if (hipMemcpy(&redo, &h->tot->src_redo, sizeof redo, hipMemcpyDeviceToHost) != hipSuccess || redo == 0) return false;
    h->lean_off = true;
    return true;
}

int check_flags(hs_lb *h) {
    LbTotals t;
    LB_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (t.qoverflow) return lfail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (h->radix_err != nullptr) {
        int re = 0;
        LB_HIP(h, hipMemcpy(&re, h->radix_err, sizeof re, hipMemcpyDeviceToHost));
        if (re) return lfail(h, HS_E_HIP, "the radix sort's look-back (debug flag 16) gave up waiting for an earlier tile (bounded spin)");
    }
    if (t.bad_client & 1) return lfail(h, HS_E_INVALID, "a client id fell outside the client table");
    if (h->any_src_profile) {          // a Source whose inversion gave up would simply stop ticking: never silently
        unsigned long long st[2] = {0ull, 0ull};
        LB_HIP(h, hipMemcpy(st, h->tab_status, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] != 0ull)
            return lfail(h, HS_E_UNSUPPORTED, "Source %lld: one arrival of its time-varying profile needs more than 64 x %lld adaptive-Simpson "
                         "intervals (csrc/hs_tables.hpp; hs_lb_set_profile_budget raises the limit) -- refused instead of stalling the device",
                         (long long)st[0] - 2, (long long)h->lane_budget);
        if (st[1] != 0ull)
            return lfail(h, HS_E_OVERFLOW, "Source %lld: its tick table overflowed (capacity %lld ticks); raise tick_capacity",
                         (long long)st[1] - 2, (long long)h->PS.tab_cap);
    }
    if (t.bad_client & 2) return lfail(h, HS_E_OVERFLOW, "a source's tick log overflowed (capacity %lld ticks); raise tick_capacity", (long long)h->cap);
    if (t.probe_tie & 1) return lfail(h, HS_E_UNSUPPORTED, "a probe sample fell on the nanosecond of an event of its target: on load-balancer "
                                      "graphs that order (the reference's sort indices) is not lowered");
    if (t.probe_tie & 2) return lfail(h, HS_E_UNSUPPORTED, "a probed backend rejected deliveries (no free worker): its queue is not "
                                      "work-conserving and the samples cannot be read off the logs");
    return HS_OK;
}

}  // namespace

extern "C" {

const char *hs_lb_last_error(const hs_lb *h) { return h ? h->error.c_str() : g_lb_error.c_str(); }

void hs_md5(const char *msg, int64_t len, uint8_t out[16]) {
    Md5 m;
    m.digest(msg, (size_t)len, out);
}

int hs_lb_create(const hs_lb_config *cfg, const hs_lb_sources *src, const hs_lb_backends *be, hs_lb **out) {
    if (!cfg || !src || !be || !out) return lfail(nullptr, HS_E_INVALID, "hs_lb_create: null argument");
    if (cfg->struct_size != sizeof(hs_lb_config)) return lfail(nullptr, HS_E_INVALID, "hs_lb_create: hs_lb_config size mismatch (ABI %d)", HS_ABI_VERSION);
    const int S = cfg->n_sources, B = cfg->n_backends;
    if (S <= 0) return lfail(nullptr, HS_E_INVALID, "hs_lb_create: n_sources must be > 0");
    if (B <= 0) return lfail(nullptr, HS_E_INVALID, "hs_lb_create: n_backends must be > 0 (the reference rejects every request otherwise)");
    if (cfg->strategy < HS_LB_CONSISTENT_HASH || cfg->strategy > HS_LB_RANDOM) return lfail(nullptr, HS_E_UNSUPPORTED, "load-balancing strategy %d is not lowered", cfg->strategy);
    const bool chash = cfg->strategy == HS_LB_CONSISTENT_HASH;
    if (chash && cfg->virtual_nodes < 1) return lfail(nullptr, HS_E_INVALID, "virtual_nodes must be >= 1, got %d", cfg->virtual_nodes);
    if (cfg->horizon_ns < cfg->start_ns || cfg->start_ns < 0) return lfail(nullptr, HS_E_INVALID, "hs_lb_create: bad start / horizon");
    if (!src->src_rate || (chash && !src->n_clients)) return lfail(nullptr, HS_E_INVALID, "src_rate and n_clients are required");
    if (!be->names || !be->name_off) return lfail(nullptr, HS_E_INVALID, "backend names are required (the ring hashes them)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return lfail(nullptr, HS_E_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return lfail(nullptr, HS_E_INVALID, "device ordinal %d out of range (%d devices)", cfg->device, ndev);
    // ---- validation + sizing
    const double horizon_s = (double)(cfg->horizon_ns - cfg->start_ns) / 1e9;
    double max_ticks = 0.0, total_rate = 0.0, min_rate = 1e300;
    int64_t kmax = 0;
    for (int i = 0; i < S; ++i) {
        const int sk = src->src_kind ? src->src_kind[i] : HS_SRC_POISSON;
        if (sk != HS_SRC_POISSON && sk != HS_SRC_CONSTANT) return lfail(nullptr, HS_E_INVALID, "source %d: unknown source kind %d", i, sk);
        const double r = src->src_rate[i];
        if (!(r > 0.0) || !std::isfinite(r)) return lfail(nullptr, HS_E_INVALID, "source %d: rate must be > 0 (got %g)", i, r);
        if (r > 1e8) return lfail(nullptr, HS_E_UNSUPPORTED, "source %d: rate %g above 1e8/s is not supported", i, r);
        max_ticks = std::max(max_ticks, r * horizon_s);
        min_rate = std::min(min_rate, r);
        total_rate += r;
        if (chash) {
            if (src->n_clients[i] < 1) return lfail(nullptr, HS_E_INVALID, "source %d: n_clients must be >= 1", i);
            kmax = std::max(kmax, src->n_clients[i]);
        }
    }
    // RANDOM: the key draw IS the backend index -- int(u * B) through an identity table; ROUND_ROBIN: the Sources' backend column is
    // a placeholder (0), the assignment follows the global arrival order (hs_lb_rr_assign)
    if (cfg->strategy == HS_LB_RANDOM) kmax = B;
    if (cfg->strategy == HS_LB_ROUND_ROBIN) kmax = 1;
    if (kmax > (1ll << 26)) return lfail(nullptr, HS_E_UNSUPPORTED, "n_clients above 2^26 is not supported (client -> backend table)");
    int maxc = 1;
    bool any_no_sink = false;
    for (int j = 0; j < B; ++j) {
        const int c = be->concurrency ? be->concurrency[j] : 1;
        if (c < 1) return lfail(nullptr, HS_E_INVALID, "backend %d: max_concurrent must be >= 1, got %d", j, c);
        if (c > 32) return lfail(nullptr, HS_E_UNSUPPORTED, "backend %d: concurrency %d > 32 is not lowered yet", j, c);
        maxc = std::max(maxc, c);
        const int vk = be->svc_kind ? be->svc_kind[j] : HS_LAT_CONSTANT;
        if (vk != HS_LAT_EXPONENTIAL && vk != HS_LAT_CONSTANT) return lfail(nullptr, HS_E_UNSUPPORTED, "backend %d: service distribution kind %d is not lowered", j, vk);
        const double mean = be->svc_mean_s ? be->svc_mean_s[j] : 0.01;
        if (!(mean >= 0.0) || !std::isfinite(mean) || (vk == HS_LAT_EXPONENTIAL && !(mean > 0.0)))
            return lfail(nullptr, HS_E_INVALID, "backend %d: bad service mean %g", j, mean);
        const int eg = be->egress ? be->egress[j] : HS_EGRESS_SINK;
        if (eg != HS_EGRESS_NONE && eg != HS_EGRESS_SINK) return lfail(nullptr, HS_E_UNSUPPORTED, "backend %d: egress kind %d is not lowered", j, eg);
        if (eg == HS_EGRESS_NONE) any_no_sink = true;
        if (be->name_off[j + 1] < be->name_off[j] || be->name_off[j + 1] - be->name_off[j] > 200)
            return lfail(nullptr, HS_E_INVALID, "backend %d: bad name", j);
    }
    hs_lb *h = new (std::nothrow) hs_lb();
    if (!h) return lfail(nullptr, HS_E_INVALID, "out of host memory");
    h->cfg = *cfg;
    h->C = maxc <= 1 ? 1 : maxc <= 2 ? 2 : maxc <= 4 ? 4 : maxc <= 8 ? 8 : maxc <= 16 ? 16 : 32;
    h->tb = bit_length((uint64_t)cfg->horizon_ns);
    h->bb = bit_length((uint64_t)(B - 1));
    if (h->tb + h->bb > 64 || h->tb > 56) { delete h; return lfail(nullptr, HS_E_UNSUPPORTED, "horizon x backends do not fit the 64-bit sort key"); }
    int64_t cap = cfg->tick_capacity;
    if (cap <= 0) cap = ((int64_t)(max_ticks + 10.0 * std::sqrt(max_ticks + 1.0) + 64.0) + 15) & ~(int64_t)15;
    h->cap = cap;
    // hs_lb_source_draws produces whole chunks of 16 ticks (lb_draw_index tiles kA / vA by 16-tick chunks: a partial last chunk would
    // reach past cap * roundup64(S) slots when S > 64), so a tick_capacity that is no multiple of 16 rounds DOWN; the rest is self-drawn
    h->n_pre = std::min<int64_t>(cap & ~(int64_t)15, ((int64_t)(max_ticks + 5.0 * std::sqrt(max_ticks + 1.0) + 16.0) + 15) & ~(int64_t)15);
    h->n_slots = cap * (int64_t)S;
    h->f64_times = cfg->start_ns >= 0 && cfg->horizon_ns < (1ll << 50) && min_rate > 1e-3;   // (one increment <= 36.8 / rate seconds)
    if ((double)h->n_slots * 88.0 > 200e9) { delete h; return lfail(nullptr, HS_E_INVALID, "buffers would need %.1f GB", (double)h->n_slots * 88.0 / 1e9); }
    {   // rows of the [k][backend] layout: three times the mean load of a backend (consistent hashing with >= 100 virtual
        // nodes keeps the busiest backend below ~2x); capped so that the five transposed arrays stay within ~4x n_slots
        const double mean_be = total_rate * horizon_s / (double)B;
        int64_t rows = (int64_t)(3.0 * mean_be) + 64;
        if ((double)rows * (double)B > 4.0 * (double)h->n_slots) rows = 0;
        h->LY.rows = rows;
        h->n_layout = std::max<int64_t>(h->n_slots, rows * (int64_t)B);
    }
    h->n_tiles = (int)((h->n_layout + kRadixTile - 1) / kRadixTile);
    {   // The sorts skip low key bits: elements whose keys agree on the sorted bits stay in input order and are put in
        // full-key order inside those runs afterwards (hs_lb_segments / hs_lb_sink_finish).  Skip whole 8-bit digits while
        // the EXPECTED number of elements per bucket stays <= 1 (runs of two or three, rare longer ones): a bucket of the
        // arrival sort is (one backend, 2^g ns) and sees total_rate / B requests per second on average, a bucket of the
        // Sink merge is 2^g ns of all completions.
        const double per_backend = total_rate / (double)B;
        const int r_arr = (h->tb + h->bb) % kRadixBits;
        h->g_arr = 0;
        for (int g = r_arr; g <= h->tb && h->tb + h->bb - g >= kRadixBits; g += kRadixBits)
            if (per_backend * std::ldexp(1.0, g) * 1e-9 <= 1.0 || g == r_arr) h->g_arr = g; else break;
        h->g_sink = 0;
        const int r_snk = h->tb % kRadixBits;
        for (int g = r_snk; h->tb - g >= kRadixBits; g += kRadixBits)
            if (total_rate * std::ldexp(1.0, g) * 1e-9 <= 1.0 || g == r_snk) h->g_sink = g; else break;
    }
    {
        const int sb = bit_length((uint64_t)(h->n_layout > 1 ? h->n_layout - 1 : 1));
        h->slot_bits = (h->tb + sb <= 64) ? sb : 0;                     // created_at (tb bits) and the slot share one word
    }
    // ---- the ring: ConsistentHash.add_backend for every backend in order (strategies.py:381-391)
    const int V = chash ? cfg->virtual_nodes : 0;
    h->ring.resize((size_t)B * V);
    if (chash) {
        char key[256];
        size_t k = 0;
        for (int j = 0; j < B; ++j) {
            const int nl = be->name_off[j + 1] - be->name_off[j];
            memcpy(key, be->names + be->name_off[j], (size_t)nl);
            for (int i = 0; i < V; ++i, ++k) {
                const int m = snprintf(key + nl, sizeof(key) - (size_t)nl, ":%d", i);          // f"{backend.name}:{i}"
                md5_u128(key, (size_t)(nl + m), h->ring[k].hi, h->ring[k].lo);
                h->ring[k].backend = j; h->ring[k].seq = (int32_t)k;
            }
        }
        std::sort(h->ring.begin(), h->ring.end(), [](const RingPoint &x, const RingPoint &y) {
            if (x.hi != y.hi) return x.hi < y.hi;
            if (x.lo != y.lo) return x.lo < y.lo;
            return x.seq < y.seq;                     // list.sort is stable
        });
    }
    // ---- client id -> backend (ConsistentHash.select for key str(id)); a pure function of the id
    std::vector<int32_t> table((size_t)kmax);
    if (chash) {
        char key[32];
        for (int64_t c = 0; c < kmax; ++c) {
            const int m = snprintf(key, sizeof key, "%lld", (long long)c);
            table[(size_t)c] = ring_select(h->ring, key, (size_t)m);
        }
    } else for (int64_t c = 0; c < kmax; ++c) table[(size_t)c] = (int32_t)c;
    h->n_table = kmax;
    hipError_t e = hipSetDevice(cfg->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) h->n_simd = 4 * cus; }
    hipEvent_t *evs[6] = {&h->ev0, &h->ev1, &h->evs0, &h->evs1, &h->evs2, &h->evs3};
    for (auto pe : evs) if (e == hipSuccess) e = hipEventCreate(pe);
    if (e != hipSuccess) { int rc = lfail(nullptr, HS_E_HIP, "device setup: %s", hipGetErrorString(e)); hs_lb_destroy(h); return rc; }
    int rc = HS_OK;
    std::vector<uint64_t> sbase((size_t)S), bbase((size_t)B);
    for (int i = 0; i < S; ++i) sbase[(size_t)i] = (uint64_t)i;
    for (int j = 0; j < B; ++j) bbase[(size_t)j] = (uint64_t)S + (uint64_t)j;
#define TRY(x) do { if ((rc = (x))) { g_lb_error = h->error; hs_lb_destroy(h); return rc; } } while (0)
    TRY(lupload<uint8_t>(h, &h->PS.kind, src->src_kind, (size_t)S, (uint8_t)HS_SRC_POISSON));
    TRY(lupload<double>(h, &h->PS.rate, src->src_rate, (size_t)S, 1.0));
    TRY(lupload<int64_t>(h, &h->PS.stop, src->src_stop_after_ns, (size_t)S, (int64_t)-1));
    h->src_stop_h.assign((size_t)S, (int64_t)-1);
    if (src->src_stop_after_ns) h->src_stop_h.assign(src->src_stop_after_ns, src->src_stop_after_ns + S);
    for (int i = 0; i < S && src->src_stop_after_ns; ++i) if (src->src_stop_after_ns[i] >= 0) h->any_stop = true;
    if (chash) TRY(lupload<int64_t>(h, &h->PS.n_clients, src->n_clients, (size_t)S, (int64_t)1));
    else TRY(lupload<int64_t>(h, &h->PS.n_clients, (const int64_t *)nullptr, (size_t)S, (int64_t)kmax));
    {   // time-varying profiles (load/profile.py:52-113): src_rate of such a Source is its PEAK rate (it sizes the tick log)
        std::vector<uint8_t> pk((size_t)S, (uint8_t)0);
        std::vector<double> pp((size_t)S * 4, 0.0);
        for (int i = 0; i < S && src->src_profile_kind; ++i) {
            const int k = src->src_profile_kind[i];
            if (k == 0) continue;
            if (k != 1 && k != 2) { rc = lfail(nullptr, HS_E_UNSUPPORTED, "source %d: profile kind %d is not lowered", i, k); hs_lb_destroy(h); return rc; }
            if (!src->src_profile_params) { rc = lfail(nullptr, HS_E_INVALID, "src_profile_params is required with src_profile_kind"); hs_lb_destroy(h); return rc; }
            const double *q = src->src_profile_params + 4 * (size_t)i;
            for (int j = 0; j < 4; ++j) {
                if (!std::isfinite(q[j]) || q[j] < 0.0) { rc = lfail(nullptr, HS_E_INVALID, "source %d: bad profile parameter %g", i, q[j]); hs_lb_destroy(h); return rc; }
                pp[(size_t)j * S + i] = q[j];
            }
            if (k == 1 && !(q[0] > 0.0)) { rc = lfail(nullptr, HS_E_INVALID, "source %d: LinearRampProfile needs duration_s > 0", i); hs_lb_destroy(h); return rc; }
            pk[(size_t)i] = (uint8_t)k;
            h->any_src_profile = true;
        }
        TRY(lupload<uint8_t>(h, &h->PS.prof_kind, pk.data(), (size_t)S, (uint8_t)0));
        TRY(lupload<double>(h, &h->PS.prof_p, pp.data(), (size_t)S * 4, 0.0));
        if (h->any_src_profile) {
            // one tick-table row per time-varying Source (hs_tables.hpp); a table holds cap + 2 ticks: everything the tick
            // log can take plus the two beyond the horizon
            std::vector<TickRow> rows;
            std::vector<int32_t> row_of((size_t)S, -1);
            for (int i = 0; i < S; ++i) {
                if (pk[(size_t)i] == 0) continue;
                TickRow r{};
                r.kind = pk[(size_t)i];
                r.poisson = (src->src_kind ? src->src_kind[i] : HS_SRC_POISSON) == HS_SRC_POISSON ? 1u : 0u;
                r.p0 = pp[(size_t)i]; r.p1 = pp[(size_t)S + i]; r.p2 = pp[(size_t)2 * S + i]; r.p3 = pp[(size_t)3 * S + i];
                r.seed = cfg->seed;
                r.sid = stream_id(src->stream_base ? src->stream_base[i] : sbase[(size_t)i], kStreamArrival);
                r.owner = i;
                row_of[(size_t)i] = (int32_t)rows.size();
                rows.push_back(r);
            }
            h->n_tab_rows = (int)rows.size();
            h->PS.tab_cap = h->cap + 2;
            TRY(lalloc(h, &h->tab_rows, rows.size()));
            if (hipMemcpy(h->tab_rows, rows.data(), rows.size() * sizeof(TickRow), hipMemcpyHostToDevice) != hipSuccess) {
                rc = lfail(nullptr, HS_E_HIP, "memcpy"); hs_lb_destroy(h); return rc;
            }
            TRY(lupload<int32_t>(h, &h->PS.tab_row, row_of.data(), (size_t)S, -1));
            TRY(lalloc(h, &h->tab_times, rows.size() * (size_t)h->PS.tab_cap));
            TRY(lalloc(h, &h->tab_count, rows.size()));
            TRY(lalloc(h, &h->tab_status, 2));
            h->PS.tab_times = h->tab_times;
        }
    }
    TRY(lupload<uint64_t>(h, &h->PS.base, src->stream_base ? src->stream_base : sbase.data(), (size_t)S, 0));
    TRY(lalloc(h, &h->PS.count, (size_t)S)); TRY(lalloc(h, &h->PS.generated, (size_t)S)); TRY(lalloc(h, &h->PS.cand, (size_t)S));
    TRY(lupload<int32_t>(h, &h->PB.conc, be->concurrency, (size_t)B, 1));
    TRY(lupload<uint8_t>(h, &h->PB.svc_kind, be->svc_kind, (size_t)B, (uint8_t)HS_LAT_CONSTANT));
    TRY(lupload<double>(h, &h->PB.svc_mean, be->svc_mean_s, (size_t)B, 0.01));
    TRY(lupload<int64_t>(h, &h->PB.qcap, be->queue_cap, (size_t)B, (int64_t)-1));
    TRY(lupload<uint8_t>(h, &h->PB.egress, be->egress, (size_t)B, (uint8_t)HS_EGRESS_SINK));
    TRY(lupload<uint64_t>(h, &h->PB.base, be->stream_base ? be->stream_base : bbase.data(), (size_t)B, 0));
    TRY(lalloc(h, &h->PB.accepted, (size_t)B)); TRY(lalloc(h, &h->PB.dropped, (size_t)B)); TRY(lalloc(h, &h->PB.completed, (size_t)B));
    TRY(lalloc(h, &h->PB.rejected, (size_t)B)); TRY(lalloc(h, &h->PB.received, (size_t)B)); TRY(lalloc(h, &h->PB.depth, (size_t)B));
    TRY(lalloc(h, &h->PB.active, (size_t)B)); TRY(lalloc(h, &h->PB.total_service, (size_t)B)); TRY(lalloc(h, &h->PB.cand, (size_t)B));
    {
        const int32_t *dt = nullptr;
        TRY(lupload<int32_t>(h, &dt, table.data(), (size_t)kmax, 0));
        h->client_be = const_cast<int32_t *>(dt);
    }
    const size_t NS = (size_t)h->n_slots;
    TRY(lalloc(h, &h->keys0, NS)); TRY(lalloc(h, &h->vals0, NS));
    // (kA / vA also hold hs_lb_source_draws' tiled output before the first sort: whole wavefronts of Sources, lb_draw_index)
    const size_t NSP = (size_t)h->cap * (((size_t)S + 63) & ~(size_t)63);
    TRY(lalloc(h, &h->kA, NSP)); TRY(lalloc(h, &h->vA, NSP)); TRY(lalloc(h, &h->kB, NS)); TRY(lalloc(h, &h->vB, NS));
    TRY(lalloc(h, &h->off, (size_t)B + 1));
    const size_t NL = (size_t)h->n_layout;
    TRY(lalloc(h, &h->adm, NS)); TRY(lalloc(h, &h->sink_t, NL)); TRY(lalloc(h, &h->sink_created, NL)); TRY(lalloc(h, &h->sink_S, NL));
    if (h->LY.rows > 0) {
        TRY(lalloc(h, &h->LY.tkey, (size_t)h->LY.rows * (size_t)B));
        TRY(lalloc(h, &h->LY.tsv, (size_t)h->LY.rows * (size_t)B));
    }
    TRY(lalloc(h, &h->n_merge, 1));
    TRY(lalloc(h, &h->bev, (size_t)3 * (size_t)B)); TRY(lalloc(h, &h->blast, (size_t)B)); TRY(lalloc(h, &h->redo, (size_t)B));
    TRY(lalloc(h, &h->scan_cand, (size_t)(B + kLbBlock / 64 - 1) / (kLbBlock / 64) + 1));
    TRY(lalloc(h, &h->scan_part, (size_t)kScanParts)); TRY(lalloc(h, &h->scan_ticket, (size_t)1));
    LB_HIP(h, hipMemset(h->scan_ticket, 0, sizeof(unsigned)));
    if (cfg->shared_sink) { TRY(lalloc(h, &h->out_t, NS)); TRY(lalloc(h, &h->out_created, NS)); }
    for (int j = 0; j < B; ++j)
        if ((be->concurrency ? be->concurrency[j] : 1) == 1 && (be->queue_cap ? be->queue_cap[j] : -1) < 0) h->any_simple = true;
    h->any_no_sink = any_no_sink;
    TRY(lalloc(h, &h->svdraw, (h->C == 1 && h->any_simple) ? NS : (size_t)1));
    TRY(lalloc(h, &h->n_slots_dev, 1)); TRY(lalloc(h, &h->n_arr, 1)); TRY(lalloc(h, &h->n_done, 1)); TRY(lalloc(h, &h->n_tmp, 1));
    TRY(lalloc(h, &h->hist, (size_t)kRadixBins * (size_t)h->n_tiles)); TRY(lalloc(h, &h->row_total, (size_t)kRadixBins));
    TRY(lalloc(h, &h->digit_base, (size_t)kRadixBins));
    TRY(lalloc(h, &h->ghist, (size_t)kRadixMaxPasses * kRadixBins)); TRY(lalloc(h, &h->tickets, (size_t)kRadixMaxPasses + 1));
    TRY(lalloc(h, &h->radix_err, (size_t)1));
    LB_HIP(h, hipMemset(h->tickets, 0, (kRadixMaxPasses + 1) * sizeof(uint32_t)));
    LB_HIP(h, hipMemset(h->radix_err, 0, sizeof(int)));
    TRY(lalloc(h, &h->tot, 1));
#undef TRY
    if (hipMemcpy(h->n_slots_dev, &h->n_slots, 8, hipMemcpyHostToDevice) != hipSuccess) {
        rc = lfail(nullptr, HS_E_HIP, "hipMemcpy failed"); hs_lb_destroy(h); return rc;
    }
    *out = h;
    return HS_OK;
}

int hs_lb_run(hs_lb *h, int64_t end_ns) {
    if (!h) return lfail(h, HS_E_INVALID, "hs_lb_run: null handle");
    if (end_ns < h->cfg.start_ns || end_ns > h->cfg.horizon_ns) return lfail(h, HS_E_INVALID, "end_ns outside [start_ns, horizon_ns]");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    LB_HIP(h, hipEventRecord(h->ev0, h->stream));
    int rc = run_async(h, end_ns);
    if (rc) return rc;
    LB_HIP(h, hipEventRecord(h->ev1, h->stream));
    LB_HIP(h, hipStreamSynchronize(h->stream));
    float ms = 0, s1 = 0, s2 = 0;
    hipEventElapsedTime(&ms, h->ev0, h->ev1);
    hipEventElapsedTime(&s1, h->evs0, h->evs1);
    hipEventElapsedTime(&s2, h->evs2, h->evs3);
    h->last_run_ms = ms; h->last_sort_ms = s1 + s2;
    if (lean_gave_up(h)) return hs_lb_run(h, end_ns);          // (once: lean_off is set)
    return check_flags(h);
}

int hs_lb_bench_runs(hs_lb *h, int64_t end_ns, int32_t repeats, float *run_ms_out, float *sort_ms_out) {
    if (!h || repeats <= 0) return lfail(h, HS_E_INVALID, "hs_lb_bench_runs: bad argument");
    if (end_ns < h->cfg.start_ns || end_ns > h->cfg.horizon_ns) return lfail(h, HS_E_INVALID, "end_ns outside [start_ns, horizon_ns]");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    // `repeats` complete runs enqueued back to back (every count lives in device memory: a run needs no host synchronisation), each
    // bracketed by its own HIP events on the engine's stream; ONE synchronisation at the end, as hs_engine_bench_runs does
    std::vector<hipEvent_t> ev((size_t)repeats * 6);
    for (auto &e : ev) LB_HIP(h, hipEventCreate(&e));
    hipEvent_t keep[4] = {h->evs0, h->evs1, h->evs2, h->evs3};
    int rc = HS_OK;
    for (int r = 0; r < repeats && rc == HS_OK; ++r) {
        hipEvent_t *e = &ev[(size_t)r * 6];
        h->evs0 = e[2]; h->evs1 = e[3]; h->evs2 = e[4]; h->evs3 = e[5];
        if (hipEventRecord(e[0], h->stream) != hipSuccess) rc = lfail(h, HS_E_HIP, "hipEventRecord failed");
        if (rc == HS_OK) rc = run_async(h, end_ns);
        if (rc == HS_OK && hipEventRecord(e[1], h->stream) != hipSuccess) rc = lfail(h, HS_E_HIP, "hipEventRecord failed");
    }
    h->evs0 = keep[0]; h->evs1 = keep[1]; h->evs2 = keep[2]; h->evs3 = keep[3];
    if (rc == HS_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = lfail(h, HS_E_HIP, "hipStreamSynchronize failed");
    for (int r = 0; r < repeats && rc == HS_OK; ++r) {
        hipEvent_t *e = &ev[(size_t)r * 6];
        float ms = 0, s1 = 0, s2 = 0;
        hipEventElapsedTime(&ms, e[0], e[1]); hipEventElapsedTime(&s1, e[2], e[3]); hipEventElapsedTime(&s2, e[4], e[5]);
        h->last_run_ms = ms; h->last_sort_ms = s1 + s2;
        if (run_ms_out) run_ms_out[r] = ms;
        if (sort_ms_out) sort_ms_out[r] = s1 + s2;
    }
    for (auto &e : ev) hipEventDestroy(e);
    if (rc == HS_OK && lean_gave_up(h)) return hs_lb_bench_runs(h, end_ns, repeats, run_ms_out, sort_ms_out);   // (once)
    return rc == HS_OK ? check_flags(h) : rc;
}

int hs_lb_get_summary(hs_lb *h, hs_summary *out) {
    if (!h || !out) return lfail(h, HS_E_INVALID, "hs_lb_get_summary: null argument");
    if (!h->ran) return lfail(h, HS_E_STATE, "hs_lb_run has not been called");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    LB_HIP(h, hipStreamSynchronize(h->stream));
    LbTotals t;
    LB_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    int64_t total = 0;
    for (int k = 0; k < HS_EV_KINDS; ++k) { out->events_by_kind[k] = (int64_t)t.ev[k]; total += (int64_t)t.ev[k]; }
    out->events_processed = total;
    out->final_time_ns = t.final_time;
    out->requests_completed = (int64_t)t.completed;
    out->sink_records = (int64_t)t.received;
    out->last_run_ms = h->last_run_ms;
    out->kernel_ms = h->last_sort_ms;
    out->launches = h->launches;
    return HS_OK;
}

int hs_lb_get_stats(hs_lb *h, const hs_lb_stats *o) {
    if (!h || !o) return lfail(h, HS_E_INVALID, "hs_lb_get_stats: null argument");
    if (!h->ran) return lfail(h, HS_E_STATE, "hs_lb_run has not been called");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    LB_HIP(h, hipStreamSynchronize(h->stream));
    const size_t S = (size_t)h->cfg.n_sources, B = (size_t)h->cfg.n_backends;
    if (o->generated) LB_HIP(h, hipMemcpy(o->generated, h->PS.generated, S * 8, hipMemcpyDeviceToHost));
#define DL(dst, srcp, T) if (o->dst) LB_HIP(h, hipMemcpy(o->dst, h->PB.srcp, B * sizeof(T), hipMemcpyDeviceToHost))
    DL(accepted, accepted, int64_t); DL(dropped, dropped, int64_t); DL(completed, completed, int64_t);
    DL(rejected, rejected, int64_t); DL(total_service_s, total_service, double); DL(queue_depth, depth, int64_t);
    DL(active, active, int32_t); DL(sink_received, received, int64_t);
#undef DL
    if (o->total_requests || o->lb) {
        std::vector<int64_t> off(B + 1);
        LB_HIP(h, hipMemcpy(off.data(), h->off, (B + 1) * 8, hipMemcpyDeviceToHost));
        if (o->total_requests) for (size_t j = 0; j < B; ++j) o->total_requests[j] = off[j + 1] - off[j];
        if (o->lb) { o->lb[0] = off[B]; o->lb[1] = off[B]; o->lb[2] = 0; o->lb[3] = 0; o->lb[4] = 0; }
    }
    return HS_OK;
}

int hs_lb_set_probes(hs_lb *h, int32_t n_probes, const int32_t *target_kind, const int32_t *target_index,
                     const uint8_t *metric, const double *interval_s) {
    if (!h) return lfail(h, HS_E_INVALID, "hs_lb_set_probes: null handle");
    if (h->Q.n > 0) return lfail(h, HS_E_STATE, "probes already set");
    if (n_probes <= 0) return HS_OK;
    if (!target_kind || !target_index || !metric || !interval_s) return lfail(h, HS_E_INVALID, "hs_lb_set_probes: null argument");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.n_backends;
    std::vector<double> rate((size_t)n_probes);
    double min_iv = 0.0;
    for (int j = 0; j < n_probes; ++j) {
        const int k = target_kind[j], i = target_index[j], m = metric[j];
        if (k < 0 || k > 2) return lfail(h, HS_E_UNSUPPORTED, "probe %d: target kind %d (0 = backend Server, 1 = Sink, 2 = Source)", j, k);
        if (k == 2) {
            if (i < 0 || i >= h->cfg.n_sources) return lfail(h, HS_E_INVALID, "probe %d: source %d out of range", j, i);
            if (m != HS_PROBE_GENERATED) return lfail(h, HS_E_UNSUPPORTED, "probe %d: metric %d is not an attribute of a Source", j, m);
            if (h->src_stop_h[(size_t)i] >= 0)
                return lfail(h, HS_E_UNSUPPORTED, "probe %d: a Source with stop_after keeps ticking without Requests, which its log does not hold", j);
        }
        if (k == 0 && (i < 0 || i >= B)) return lfail(h, HS_E_INVALID, "probe %d: backend %d out of range", j, i);
        if (k == 1 && (i < 0 || i >= (h->cfg.shared_sink ? 1 : B))) return lfail(h, HS_E_INVALID, "probe %d: sink %d out of range", j, i);
        if (k == 0 && !(m == HS_PROBE_DEPTH || m == HS_PROBE_ACTIVE || m == HS_PROBE_ACCEPTED || m == HS_PROBE_DROPPED ||
                        m == HS_PROBE_COMPLETED))
            return lfail(h, HS_E_UNSUPPORTED, "probe %d: metric %d is not an attribute of a Server", j, m);
        if (k == 1 && m != HS_PROBE_RECEIVED) return lfail(h, HS_E_UNSUPPORTED, "probe %d: metric %d is not an attribute of a Sink", j, m);
        if (h->any_no_sink) return lfail(h, HS_E_UNSUPPORTED, "probes on a load-balancer graph need the backends' completion logs (a Sink downstream)");
        const double iv = interval_s[j];
        if (!(iv > 0.0) || !std::isfinite(iv)) return lfail(h, HS_E_INVALID, "Probe interval must be positive.");   // probe.py:29-30
        rate[(size_t)j] = 1.0 / iv;
        if (min_iv == 0.0 || iv < min_iv) min_iv = iv;
    }
    const double horizon_s = (double)(h->cfg.horizon_ns - h->cfg.start_ns) / 1e9;
    const int64_t pcap = (int64_t)(horizon_s / min_iv) + 8;
    if ((double)pcap * n_probes * 16.0 > 8e9) return lfail(h, HS_E_INVALID, "probe logs would need %.1f GB", (double)pcap * n_probes * 16.0 / 1e9);
    LbProbes Q{};
    Q.n = n_probes; Q.pcap = pcap;
    int32_t *dk = nullptr, *di = nullptr; uint8_t *dm = nullptr; double *dr = nullptr;
    int rc;
#define PTRY(x) do { if ((rc = (x))) return rc; } while (0)
    PTRY(lalloc(h, &dk, (size_t)n_probes)); PTRY(lalloc(h, &di, (size_t)n_probes)); PTRY(lalloc(h, &dm, (size_t)n_probes));
    PTRY(lalloc(h, &dr, (size_t)n_probes));
    PTRY(lalloc(h, &Q.tick, (size_t)n_probes * (size_t)pcap)); PTRY(lalloc(h, &Q.val, (size_t)n_probes * (size_t)pcap));
    PTRY(lalloc(h, &Q.n_tick, (size_t)n_probes)); PTRY(lalloc(h, &Q.cnt, (size_t)n_probes)); PTRY(lalloc(h, &Q.cand, (size_t)n_probes));
#undef PTRY
    LB_HIP(h, hipMemcpy(dk, target_kind, (size_t)n_probes * 4, hipMemcpyHostToDevice));
    LB_HIP(h, hipMemcpy(di, target_index, (size_t)n_probes * 4, hipMemcpyHostToDevice));
    LB_HIP(h, hipMemcpy(dm, metric, (size_t)n_probes, hipMemcpyHostToDevice));
    LB_HIP(h, hipMemcpy(dr, rate.data(), (size_t)n_probes * 8, hipMemcpyHostToDevice));
    LB_HIP(h, hipMemset(Q.cnt, 0, (size_t)n_probes * 8));
    Q.kind = dk; Q.idx = di; Q.metric = dm; Q.rate = dr;
    // the tick times do not depend on the run: once, from the tick-table kernel (hs_tables.hpp: _ProbeProfile goes through
    // the reference's numerical path), Q.tick IS the table (row j, pcap entries), up to the second tick beyond the horizon
    {
        std::vector<TickRow> rows((size_t)n_probes);
        for (int j = 0; j < n_probes; ++j) {
            TickRow r{};
            r.kind = kProfGeneralConstant; r.poisson = 0; r.p0 = rate[(size_t)j]; r.owner = j;
            rows[(size_t)j] = r;
        }
        TickRow *drows = nullptr;
        unsigned long long *dstat = nullptr;
        if ((rc = lalloc(h, &drows, (size_t)n_probes))) return rc;
        if ((rc = lalloc(h, &dstat, 2))) return rc;
        LB_HIP(h, hipMemcpy(drows, rows.data(), rows.size() * sizeof(TickRow), hipMemcpyHostToDevice));
        LB_HIP(h, tick_tables_launch(h->stream, drows, n_probes, h->cfg.start_ns, h->cfg.horizon_ns, pcap, Q.tick, Q.n_tick, dstat,
                                     h->lane_budget, false));
        LB_HIP(h, hipStreamSynchronize(h->stream));
        unsigned long long st[2] = {0ull, 0ull};
        LB_HIP(h, hipMemcpy(st, dstat, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] != 0ull) return lfail(h, HS_E_UNSUPPORTED, "probe %lld: a tick needs more than 64 x %lld adaptive-Simpson intervals", (long long)st[0] - 2, (long long)h->lane_budget);
        // (a table that ends before `pcap` entries simply has fewer ticks; one that fills it ends on its last entry, as before)
    }
    h->Q = Q;
    return HS_OK;
}

int hs_lb_set_profile_budget(hs_lb *h, int64_t intervals_per_lane) {
    if (!h) return lfail(h, HS_E_INVALID, "hs_lb_set_profile_budget: null handle");
    if (intervals_per_lane < 1) return lfail(h, HS_E_INVALID, "hs_lb_set_profile_budget: the budget must be >= 1");
    h->lane_budget = (long long)intervals_per_lane;
    h->tables_built = false;
    return HS_OK;
}

int64_t hs_lb_read_probe(hs_lb *h, int32_t probe, int64_t *t_ns, int64_t *values, int64_t cap) {
    if (!h || !h->ran) return lfail(h, HS_E_STATE, "hs_lb_read_probe before hs_lb_run");
    if (probe < 0 || probe >= h->Q.n) return lfail(h, HS_E_INVALID, "probe %d out of range", probe);
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return lfail(h, HS_E_HIP, "device synchronisation failed");
    int64_t cnt = 0;
    if (hipMemcpy(&cnt, h->Q.cnt + probe, 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
    if (cnt > cap) cnt = cap;
    if (cnt > 0) {
        const size_t o = (size_t)probe * (size_t)h->Q.pcap;
        if (t_ns && hipMemcpy(t_ns, h->Q.tick + o, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
        if (values && hipMemcpy(values, h->Q.val + o, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
    }
    return cnt;
}

int64_t hs_lb_read_sink(hs_lb *h, int32_t sink, int64_t *t_ns, int64_t *created_ns, int64_t cap) {
    if (!h || !h->ran) return lfail(h, HS_E_STATE, "hs_lb_run has not been called");
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return lfail(h, HS_E_HIP, "device synchronisation failed");
    const int B = h->cfg.n_backends;
    int64_t cnt = 0;
    const int64_t *src_t, *src_c;
    if (h->cfg.shared_sink) {
        if (sink != 0) return lfail(h, HS_E_INVALID, "shared sink: the only sink index is 0");
        if (hipMemcpy(&cnt, h->n_done, 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
        src_t = h->out_t; src_c = h->out_created;
    } else {
        if (sink < 0 || sink >= B) return lfail(h, HS_E_INVALID, "sink index %d out of range", sink);
        int64_t o = 0;
        LbTotals tt;
        if (hipMemcpy(&cnt, h->PB.received + sink, 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&o, h->off + sink, 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&tt, h->tot, sizeof tt, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
        if (tt.use_t) {                      // [m][backend] logs: gather the backend's column
            if (cnt > cap) cnt = cap;
            if (cnt <= 0) return 0;
            int64_t *tmp = nullptr;
            if (hipMalloc(&tmp, (size_t)cnt * 8) != hipSuccess) return lfail(h, HS_E_HIP, "hipMalloc of the read-back staging buffer failed");
            const int64_t *cols[2] = {h->sink_t + sink, h->sink_created + sink};
            int64_t *dsts[2] = {t_ns, created_ns};
            for (int c = 0; c < 2; ++c) {
                if (!dsts[c]) continue;
                hipLaunchKernelGGL(hs_lb_gather_strided, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, cols[c],
                                   (int64_t)B, cnt, tmp);
                if (hipStreamSynchronize(h->stream) != hipSuccess ||
                    hipMemcpy(dsts[c], tmp, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                    hipFree(tmp);
                    return lfail(h, HS_E_HIP, "sink read-back failed");
                }
            }
            hipFree(tmp);
            return cnt;
        }
        src_t = h->sink_t + o; src_c = h->sink_created + o;
    }
    if (cnt > cap) cnt = cap;
    if (cnt > 0) {
        if (t_ns && hipMemcpy(t_ns, src_t, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
        if (created_ns && hipMemcpy(created_ns, src_c, (size_t)cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, HS_E_HIP, "memcpy");
    }
    return cnt;
}

int hs_lb_latency_stats(hs_lb *h, double out[6]) {
    if (!h || !out) return lfail(h, HS_E_INVALID, "hs_lb_latency_stats: null argument");
    if (!h->ran) return lfail(h, HS_E_STATE, "hs_lb_run has not been called");
    if (!h->cfg.shared_sink) return lfail(h, HS_E_INVALID, "hs_lb_latency_stats is for the shared Sink; per-backend Sinks are small: read them");
    LB_HIP(h, hipSetDevice(h->cfg.device));
    // the sort buffers are free between runs: keys0 <- latencies, sorted in the ping-pong buffers
    hipLaunchKernelGGL(hs_lb_latency_keys, dim3((unsigned)((h->n_slots + 255) / 256)), dim3(256), 0, h->stream, h->out_t,
                       h->out_created, h->n_done, h->keys0);
    uint64_t *kr = nullptr, *vr = nullptr;
    const int g = h->tb % kRadixBits;
    radix_sort_async(h, h->keys0, (const uint64_t *)nullptr, h->n_done, h->n_tmp, h->tb, RadixAll{}, NoVal{}, &kr, &vr, nullptr,
                     h->tb > kRadixBits ? g : 0);
    (void)vr;
    double *d_out = nullptr;
    LB_HIP(h, hipMalloc(&d_out, 6 * sizeof(double)));
    if (h->tb > kRadixBits && g) {
        // the ragged low bits were skipped: finish short runs in place (keys only) with the segment kernel's fix-up
        hipLaunchKernelGGL(hs_lb_fix_runs, dim3((unsigned)((h->n_slots + 255) / 256)), dim3(256), 0, h->stream, kr, h->n_done, g);
    }
    hipLaunchKernelGGL(hs_lb_latency_stats_kernel, dim3(1), dim3(64), 0, h->stream, kr, h->n_done, d_out, g_float_sum_mode);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 6 * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d_out);
    if (e != hipSuccess) return lfail(h, HS_E_HIP, "latency statistics failed: %s", hipGetErrorString(e));
    return HS_OK;
}

int hs_lb_ring(hs_lb *h, int32_t *ring_backend) {
    if (!h || !ring_backend) return lfail(h, HS_E_INVALID, "hs_lb_ring: null argument");
    if (h->cfg.strategy != HS_LB_CONSISTENT_HASH) return lfail(h, HS_E_STATE, "hs_lb_ring: this LoadBalancer has no ConsistentHash ring");
    for (size_t i = 0; i < h->ring.size(); ++i) ring_backend[i] = h->ring[i].backend;
    return HS_OK;
}

int32_t hs_lb_select(hs_lb *h, const char *key) {
    if (!h || !key) return lfail(h, HS_E_INVALID, "hs_lb_select: null argument");
    if (h->cfg.strategy != HS_LB_CONSISTENT_HASH) return lfail(h, HS_E_STATE, "hs_lb_select: this LoadBalancer has no ConsistentHash ring");
    return ring_select(h->ring, key, strlen(key));
}

void hs_lb_destroy(hs_lb *h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    hipEvent_t evs[6] = {h->ev0, h->ev1, h->evs0, h->evs1, h->evs2, h->evs3};
    for (auto e : evs) if (e) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int hs_debug_lb_flags(hs_lb *h, int flags) {
    if (!h) return HS_E_INVALID;
    h->flags = flags;
    return HS_OK;
}

int hs_debug_radix_sort(int32_t device, int64_t n, int32_t key_bits, const uint64_t *keys_in, const uint64_t *vals_in,
                        uint64_t *keys_out, uint64_t *vals_out, float *device_ms) {
    if (n <= 0 || key_bits < 1 || key_bits > 64 || !keys_in || !vals_in || !keys_out || !vals_out)
        return lfail(nullptr, HS_E_INVALID, "hs_debug_radix_sort: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return lfail(nullptr, HS_E_NO_DEVICE, "no HIP device visible");
    hs_lb h;
    h.cfg.device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking) != hipSuccess)
        return lfail(nullptr, HS_E_HIP, "device setup failed");
    hipEventCreate(&h.ev0); hipEventCreate(&h.ev1);
    h.n_slots = n;
    h.n_tiles = (int)((n + kRadixTile - 1) / kRadixTile);
    uint64_t *k0 = nullptr, *v0 = nullptr;
    int rc = HS_OK;
    const size_t N = (size_t)n;
    if ((rc = lalloc(&h, &k0, N)) || (rc = lalloc(&h, &v0, N)) || (rc = lalloc(&h, &h.kA, N)) || (rc = lalloc(&h, &h.vA, N)) ||
        (rc = lalloc(&h, &h.kB, N)) || (rc = lalloc(&h, &h.vB, N)) || (rc = lalloc(&h, &h.n_slots_dev, 1)) ||
        (rc = lalloc(&h, &h.n_arr, 1)) || (rc = lalloc(&h, &h.hist, (size_t)kRadixBins * (size_t)h.n_tiles)) ||
        (rc = lalloc(&h, &h.row_total, (size_t)kRadixBins)) || (rc = lalloc(&h, &h.digit_base, (size_t)kRadixBins))) {
        g_lb_error = h.error;
    } else {
        hipMemcpy(k0, keys_in, N * 8, hipMemcpyHostToDevice);
        hipMemcpy(v0, vals_in, N * 8, hipMemcpyHostToDevice);
        hipMemcpy(h.n_slots_dev, &n, 8, hipMemcpyHostToDevice);
        uint64_t *kr = nullptr, *vr = nullptr;
        hipEventRecord(h.ev0, h.stream);
        radix_sort_async(&h, k0, v0, h.n_slots_dev, h.n_arr, key_bits, RadixAll{}, NoVal{}, &kr, &vr);
        hipEventRecord(h.ev1, h.stream);
        if (hipStreamSynchronize(h.stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = lfail(nullptr, HS_E_HIP, "radix sort failed");
        else {
            hipMemcpy(keys_out, kr, N * 8, hipMemcpyDeviceToHost);
            hipMemcpy(vals_out, vr, N * 8, hipMemcpyDeviceToHost);
            float ms = 0;
            hipEventElapsedTime(&ms, h.ev0, h.ev1);
            if (device_ms) *device_ms = ms;
        }
    }
    for (void *p : h.allocs) hipFree(p);
    h.allocs.clear();
    hipEventDestroy(h.ev0); hipEventDestroy(h.ev1);
    hipStreamDestroy(h.stream);
    h.stream = nullptr; h.ev0 = h.ev1 = nullptr;
    return rc;
}

// One Sink fed by several stations (the canonical `servers = [Server(..., downstream=sink) ...]` wiring): each station
// logs its own completions in processing order; the shared Sink's lists are their merge by completion time.  Stable, so
// records of one station keep their order and equal timestamps of different stations come in station order (the
// reference orders those by event creation: DESIGN.md "known deviations" (i)).  In place.
int hs_merge_sink_records(int32_t device, int64_t n, int64_t *t_ns, int64_t *created_ns) {
    if (n < 0 || (n > 0 && (!t_ns || !created_ns))) return lfail(nullptr, HS_E_INVALID, "hs_merge_sink_records: bad argument");
    if (n < 2) return HS_OK;
    int64_t tmax = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (t_ns[i] < 0) return lfail(nullptr, HS_E_INVALID, "hs_merge_sink_records: negative timestamp");
        if (t_ns[i] > tmax) tmax = t_ns[i];
    }
    int bits = 1;
    while (bits < 63 && (tmax >> bits) != 0) ++bits;
    std::vector<uint64_t> ko((size_t)n), vo((size_t)n);
    const int rc = hs_debug_radix_sort(device, n, bits, (const uint64_t *)t_ns, (const uint64_t *)created_ns, ko.data(),
                                       vo.data(), nullptr);
    if (rc != HS_OK) return rc;
    memcpy(t_ns, ko.data(), (size_t)n * 8);
    memcpy(created_ns, vo.data(), (size_t)n * 8);
    return HS_OK;
}

// Sink.latency_stats() (components/common.py:59-76, instrumentation/data.py:197-210) of any Sink's records on the device:
int hs_set_float_sum_mode(int compensated) {
    if (compensated != 0 && compensated != 1) return lfail(nullptr, HS_E_INVALID, "hs_set_float_sum_mode: 0 (left to right) or 1 (Neumaier, CPython >= 3.12)");
    g_float_sum_mode = compensated;
    return HS_OK;
}

// latencies = t - created_at, radix-sorted, summed in index order in binary64 (hs_set_float_sum_mode), interpolated p50 / p99 -- the shared-Sink
// routine of the load-balancer engine (hs_lb_latency_stats) for the station engines' Sinks.  out = {count, avg, min, max,
// p50, p99}.  Host buffers in, six doubles out.
int hs_sink_latency_stats(int32_t device, int64_t n, const int64_t *t_ns, const int64_t *created_ns, double out[6]) {
    if (n < 0 || !out || (n > 0 && (!t_ns || !created_ns))) return lfail(nullptr, HS_E_INVALID, "hs_sink_latency_stats: bad argument");
    for (int k = 0; k < 6; ++k) out[k] = 0.0;
    if (n == 0) return HS_OK;
    std::vector<uint64_t> lat((size_t)n), zero((size_t)n, 0ull), ko((size_t)n), vo((size_t)n);
    uint64_t lmax = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (t_ns[i] < created_ns[i]) return lfail(nullptr, HS_E_INVALID, "hs_sink_latency_stats: a record completes before it was created");
        lat[(size_t)i] = (uint64_t)(t_ns[i] - created_ns[i]);
        if (lat[(size_t)i] > lmax) lmax = lat[(size_t)i];
    }
    int bits = 1;
    while (bits < 63 && (lmax >> bits) != 0) ++bits;
    int rc = hs_debug_radix_sort(device, n, bits, lat.data(), zero.data(), ko.data(), vo.data(), nullptr);
    if (rc != HS_OK) return rc;
    if (hipSetDevice(device) != hipSuccess) return lfail(nullptr, HS_E_HIP, "hipSetDevice failed");
    uint64_t *d_sorted = nullptr; int64_t *d_n = nullptr; double *d_out = nullptr;
    hipError_t e = hipMalloc(&d_sorted, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc(&d_n, 8);
    if (e == hipSuccess) e = hipMalloc(&d_out, 6 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_sorted, ko.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_n, &n, 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hs_lb_latency_stats_kernel, dim3(1), dim3(64), 0, nullptr, d_sorted, d_n, d_out, g_float_sum_mode);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 6 * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d_sorted); hipFree(d_n); hipFree(d_out);
    if (e != hipSuccess) return lfail(nullptr, HS_E_HIP, "hs_sink_latency_stats: %s", hipGetErrorString(e));
    return HS_OK;
}

}  // extern "C"
