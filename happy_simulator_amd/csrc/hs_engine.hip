// hs_engine.hip -- libhs_hip.so: HIP kernels (gfx950) and the C ABI declared in include/hs_engine.h.
//
// Replaces `Simulation._execute_until` / `_build_summary` (happysimulator/core/simulation.py:449-505,
// :543-591) and the replica fan-out of happysimulator/parallel for station LPs.  Data layout, kernels and
// their rooflines are described in DESIGN.md.  There is no CPU fallback in this file by design.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "../../include/hs_engine.h"
#include "hs_netstation.hpp"
#include "hs_exact.hpp"

using namespace hs;

// =============================================================================================
// device helpers
// =============================================================================================
namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ long long shfl_xor_ll(long long v, int o) {
    int lo = (int)(unsigned)(v & 0xffffffffll), hi = (int)(v >> 32);
    lo = __shfl_xor(lo, o, 64);
    hi = __shfl_xor(hi, o, 64);
    return ((long long)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ long long shfl_up_ll(long long v, int o) {
    int lo = (int)(unsigned)(v & 0xffffffffll), hi = (int)(v >> 32);
    lo = __shfl_up(lo, o, 64);
    hi = __shfl_up(hi, o, 64);
    return ((long long)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ bool cand_less(const Candidate &a, const Candidate &b) {
    if (a.valid != b.valid) return a.valid > b.valid;
    if (!a.valid) return false;
    if (a.t != b.t) return a.t < b.t;
    if (a.t_created != b.t_created) return a.t_created < b.t_created;
    return a.rank < b.rank;
}

__device__ __forceinline__ Candidate wave_min_cand(Candidate c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Candidate d;
        d.t = shfl_xor_ll(c.t, o);
        d.t_created = shfl_xor_ll(c.t_created, o);
        d.lp = __shfl_xor(c.lp, o, 64);
        d.rank = __shfl_xor(c.rank, o, 64);
        d.valid = __shfl_xor(c.valid, o, 64);
        if (cand_less(d, c)) c = d;
    }
    return c;
}

template <int C, bool PF, bool UNI = false>
__device__ __forceinline__ void load_station(Station<C, PF, UNI> &S, const StationParams &P, const StationState &X,
                                             const RecordLogs &L, int lp, int n, uint8_t (*qmem)[kBlock],
                                             double (*ring_a)[kBlock], double (*ring_s)[kBlock], int tid) {
    S.lp = lp; S.n = n;
    S.src_kind = P.src_kind[lp]; S.svc_kind = P.svc_kind[lp]; S.egress = P.egress[lp];
    S.conc = P.conc[lp];
    S.rate = P.src_rate[lp]; S.svc_mean = P.svc_mean[lp];
    S.svc_lambda = __ddiv_rn(1.0, S.svc_mean);                       // ExponentialLatency._lambda = 1 / mean
    S.svc_const_s = seconds_from_ns(ns_from_seconds(S.svc_mean));    // ConstantLatency: from_seconds(mean).to_seconds()
    S.svc_const_ns = ns_from_seconds(S.svc_const_s);
    S.stop_ns = P.src_stop[lp]; S.qcap = P.qcap[lp];
    S.prof.kind = kProfConstant;
    S.n_probes = 0; S.evp[0] = S.evp[1] = 0;
#pragma unroll
    for (int j = 0; j < kMaxProbes; ++j) { S.p_metric[j] = kProbeNone; S.PA[j] = kInfNs; S.seqP[j] = 0; S.crtP[j] = 0; S.p_arr[j] = 0; S.p_n[j] = 0; S.p_rate[j] = 1.0; }
    S.SA = kInfNs; S.sc_i = S.sc_end = 0; S.sc_t = P.sched_t; S.sc_idx = P.sched_idx;
    S.n_xsrc = 0; S.x_base = P.stream_base[lp];
#pragma unroll
    for (int j = 0; j < kMaxXSrc; ++j) { S.x_kind[j] = 0; S.XA[j] = kInfNs; S.seqX[j] = 0; S.crtX[j] = 0; S.x_arr[j] = 0; S.x_n[j] = 0; S.x_k[j] = 0; S.x_rate[j] = 1.0; S.x_stop[j] = -1; }
    if constexpr (PF) {
        if (P.xsrc_kind != nullptr) {
#pragma unroll
            for (int j = 0; j < kMaxXSrc; ++j) {
                const size_t o = (size_t)j * n + lp;
                S.x_kind[j] = P.xsrc_kind[o];
                if (S.x_kind[j] != 0) {
                    S.n_xsrc = j + 1;
                    S.x_rate[j] = P.xsrc_rate[o]; S.x_stop[j] = P.xsrc_stop[o];
                    S.XA[j] = X.XA[o]; S.seqX[j] = X.seqX[o]; S.crtX[j] = X.crtX[o]; S.x_arr[j] = X.x_arr[o]; S.x_n[j] = X.x_n[o];
                    S.x_k[j] = X.x_k[o];
                }
            }
        }
        if (P.sched_off != nullptr) {
            S.sc_i = X.sched_i[lp]; S.sc_end = P.sched_off[lp + 1];
            S.SA = S.sc_i < S.sc_end ? P.sched_t[S.sc_i] : kInfNs;
        }
        S.prof.kind = P.prof_kind[lp]; S.prof.owner = lp;
        S.prof.p0 = P.prof_p[lp]; S.prof.p1 = P.prof_p[(size_t)n + lp]; S.prof.p2 = P.prof_p[(size_t)2 * n + lp];
        S.prof.p3 = P.prof_p[(size_t)3 * n + lp];
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) {
            const size_t o = (size_t)j * n + lp;
            S.p_metric[j] = P.probe_metric[o];
            if (S.p_metric[j] != kProbeNone) {
                S.n_probes = j + 1;
                S.p_rate[j] = P.probe_rate[o];
                S.PA[j] = X.PA[o]; S.seqP[j] = X.seqP[o]; S.crtP[j] = X.crtP[o]; S.p_arr[j] = X.p_arr[o]; S.p_n[j] = X.p_n[o];
            }
        }
        S.probe_t = L.probe_t + lp; S.probe_v = L.probe_v + lp; S.pcap = L.pcap;
    }
    S.A = X.A[lp]; S.seqA = X.seqA[lp]; S.crtA = X.crtA[lp]; S.arr_time = X.arr_time[lp];
    S.buf = X.buf[lp]; S.active = X.active[lp]; S.seq = X.seq[lp];
    S.generated = X.generated[lp]; S.accepted = X.accepted[lp]; S.dropped = X.dropped[lp];
    S.completed = X.completed[lp]; S.rejected = X.rejected[lp]; S.started = X.started[lp];
    S.received = X.received[lp]; S.sink_w = X.sink_w[lp];
    S.total_service = X.total_service[lp];
    S.last_time = X.last_time[lp]; S.grp_time = X.grp_time[lp];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        S.D[i] = X.D[(size_t)i * n + lp]; S.seqD[i] = X.seqD[(size_t)i * n + lp];
        S.crtD[i] = X.crtD[(size_t)i * n + lp]; S.svc_s[i] = X.svc_s[(size_t)i * n + lp];
        S.crt[i] = (C > 1) ? X.crt[(size_t)i * n + lp] : 0;
    }
    S.tid = tid;
    S.init_streams(P.seed[lp], P.stream_base[lp], X.arr_k[lp], X.svc_k[lp], ring_a, ring_s);
#pragma unroll
    for (int k = 0; k < 8; ++k) S.ev[k] = 0;
    S.adm = L.adm + lp;
    S.sink_t = L.sink_t + lp;
    S.sink_created = (C > 1) ? L.sink_created + lp : nullptr;
    S.cap = L.cap; S.ls = n;
    S.overflow = 0; S.qoverflow = 0;
    S.qmem = qmem; S.tid = tid; S.qh = 0; S.qn = 0;
    const uint32_t q = X.q[lp];
    const int qn = (int)(q >> 16);
    for (int i = 0; i < qn; ++i) S.qpush((q >> (8 * i)) & 0xffu);
}

template <int C, bool PF, bool UNI = false>
__device__ __forceinline__ void store_station(const Station<C, PF, UNI> &Sc, const StationState &X, int lp, int n) {
    Station<C, PF, UNI> &S = const_cast<Station<C, PF, UNI> &>(Sc);
    X.A[lp] = S.A; X.seqA[lp] = S.seqA; X.crtA[lp] = S.crtA; X.arr_time[lp] = S.arr_time;
    X.buf[lp] = S.buf; X.active[lp] = S.active; X.seq[lp] = S.seq;
    X.generated[lp] = S.generated; X.accepted[lp] = S.accepted; X.dropped[lp] = S.dropped;
    X.completed[lp] = S.completed; X.rejected[lp] = S.rejected; X.started[lp] = S.started;
    X.received[lp] = S.received; X.sink_w[lp] = S.sink_w;
    X.total_service[lp] = S.total_service;
    X.last_time[lp] = S.last_time; X.grp_time[lp] = S.grp_time;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = S.D[i]; X.seqD[(size_t)i * n + lp] = S.seqD[i];
        X.crtD[(size_t)i * n + lp] = S.crtD[i]; X.svc_s[(size_t)i * n + lp] = S.svc_s[i];
        if (C > 1) X.crt[(size_t)i * n + lp] = S.crt[i];
    }
    X.arr_k[lp] = S.arr_k; X.svc_k[lp] = S.svc_k;   // draws CONSUMED; pre-drawn values still in the rings are dropped
    uint32_t q = 0;
    int qn = S.qn > 2 ? 2 : S.qn;   // an overshoot root leaves at most two in-group events
    for (int i = 0; i < qn; ++i) q |= (uint32_t)S.qmem[(S.qh + i) % kQCap][S.tid] << (8 * i);
    q |= (uint32_t)qn << 16;
    X.q[lp] = q;
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { X.ev_kind[(size_t)k * n + lp] += S.ev[k]; tot += S.ev[k]; }
    if constexpr (PF) {
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) if (j < S.n_probes) {
            const size_t o = (size_t)j * n + lp;
            X.PA[o] = S.PA[j]; X.seqP[o] = S.seqP[j]; X.crtP[o] = S.crtP[j]; X.p_arr[o] = S.p_arr[j]; X.p_n[o] = S.p_n[j];
        }
        X.ev_probe[lp] += S.evp[0]; X.ev_probe[(size_t)n + lp] += S.evp[1];
        tot += S.evp[0] + S.evp[1];
        if (S.sc_t != nullptr) X.sched_i[lp] = S.sc_i;
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) if (j < S.n_xsrc) {
            const size_t o = (size_t)j * n + lp;
            X.XA[o] = S.XA[j]; X.seqX[o] = S.seqX[j]; X.crtX[o] = S.crtX[j]; X.x_arr[o] = S.x_arr[j]; X.x_n[o] = S.x_n[j];
            X.x_k[o] = S.x_k[j];
        }
    }
    X.events[lp] += tot;
}

// first pending event of an LP: time, creation time, which root
template <int C, bool PF, bool UNI = false>
__device__ __forceinline__ Candidate make_candidate(const Station<C, PF, UNI> &S) {
    Candidate c;
    c.lp = S.lp; c.rank = S.lp; c.valid = 0; c.t = kInfNs; c.t_created = 0; c.pad = 0;
    if (S.qn > 0) {   // a group already in progress keeps the floor
        c.t = S.grp_time; c.t_created = S.grp_time; c.valid = 1;
        return c;
    }
    const int64_t t = S.next_time();
    if (t == kInfNs) return c;
    const int w = S.pick_root(t);
    c.t = t; c.valid = 1;
    if (w == 0) { c.t_created = S.crtA; c.pad = 2; }
    else if (w >= kRootXSrc) {
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) if (j == w - kRootXSrc) c.t_created = S.crtX[j];
        c.pad = 3 + (w - kRootXSrc);
    }
    else if (w >= kRootProbe) {
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) if (j == w - kRootProbe) c.t_created = S.crtP[j];
        c.pad = 1;              // a Probe's tick: constructed after every Source (core/simulation.py:145-160) -- ranks behind them
    }
    else if (w == kRootSched) c.t_created = INT64_MIN;   // constructed before run()
    else {
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == w - 1) c.t_created = S.crtD[i];
    }
    return c;
}

// process exactly ONE event beyond end_ns: the first micro-event of the LP's next group
template <int C, bool PF, bool UNI = false>
__device__ __forceinline__ void overshoot_one(Station<C, PF, UNI> &S) {
    if (S.qn > 0) {   // continue the in-progress group by one event
        // (only reachable when a previous window ended inside this group and the new end is still before it)
        return;
    }
    const int64_t t = S.next_time();
    if (t == kInfNs) return;
    S.run_root(S.pick_root(t), t);
    S.last_time = t;
    S.grp_time = t;
}

}  // namespace

// =============================================================================================
// kernels
// =============================================================================================

__device__ __noinline__ int64_t first_probe_tick(double rate, int64_t start_ns, int owner) {
    Profile pp;
    pp.kind = kProfGeneralConstant; pp.p0 = rate; pp.p1 = pp.p2 = pp.p3 = 0.0; pp.owner = owner;
    return prof_next_arrival(pp, start_ns, 1.0);
}
__device__ __noinline__ int64_t first_profile_arrival(const StationParams &P, int lp, int n, int64_t start_ns, double area) {
    Profile pf;
    pf.kind = P.prof_kind[lp];
    pf.p0 = P.prof_p[lp]; pf.p1 = P.prof_p[(size_t)n + lp]; pf.p2 = P.prof_p[(size_t)2 * n + lp];
    pf.p3 = P.prof_p[(size_t)3 * n + lp]; pf.owner = lp;
    return prof_next_arrival(pf, start_ns, area);
}

// Simulation.__init__ bootstrap (core/simulation.py:145-154, load/source.py:120-140): every Source draws
// its first arrival from start_ns.  Also zeroes the per-LP state.
// PF = false: no LP has a time-varying profile, a probe, a scheduled Request or a further Source -- the instantiation every
// headline workload uses carries none of the numerical inversion's scratch frame (round 1: 4 176 B per lane in the one kernel).
template <bool PF>
__global__ void __launch_bounds__(kBlock) hs_station_reset(StationParams P, StationState X, Totals *tot, int n, int C,
                                                           int64_t start_ns, NetState NX, int n_links) {
    const int lp = blockIdx.x * kBlock + threadIdx.x;
    if (NX.next_time != nullptr) {   // network engine: clear routing / link / bag state
        for (int l = lp; l < n_links; l += gridDim.x * kBlock) {
            NX.link_k[l] = 0; NX.link_in[l] = 0; NX.link_sent[l] = 0; NX.link_packets[l] = 0;
            if (NX.aq_tail != nullptr) { NX.aq_tail[l] = 0; NX.aq_head[l] = 0; NX.aq_ea[l] = 0; }   // (bound = start, tail = 0)
        }
        if (lp < n) {
            NX.route_k[lp] = 0; NX.routed[lp] = 0; NX.bag_cnt[lp] = 0; NX.in_cnt[lp] = 0; NX.in_cnt[n + lp] = 0;
            if (NX.early_upto != nullptr) { NX.early_upto[lp] = 0; NX.d_pre[lp] = start_ns; }
        }
    }
    if (lp == 0) {
        for (int k = 0; k < 15; ++k) tot->ev[k] = 0;
        tot->completed = 0; tot->received = 0; tot->final_time = start_ns; tot->cur_time = start_ns;
        tot->overflow = 0; tot->qoverflow = 0; tot->done = 0;
        tot->dbg[0] = tot->dbg[1] = tot->dbg[2] = tot->dbg[3] = 0; tot->not_done = 0;
    }
    if (lp >= n) return;
    int64_t A = kInfNs, arr_time = start_ns;
    uint64_t arr_k = 0;
    const uint32_t sk = P.src_kind[lp];
    if (sk != 0) {
        double area = 1.0;
        if (sk == 1) {
            Stream s;
            s.init(P.seed[lp], stream_id(P.stream_base[lp], kStreamArrival), 0);
            area = exp1_from_uniform(s.next_uniform());
            arr_k = 1;
        }
        bool timevarying = false;
        if constexpr (PF) timevarying = P.prof_kind[lp] != kProfConstant;
        if (timevarying) {                        // time-varying rate: the general path (hs_profile.hpp)
            if constexpr (PF) arr_time = first_profile_arrival(P, lp, n, start_ns, area);
        } else {
            const double t_next = __dadd_rn(seconds_from_ns(start_ns), __ddiv_rn(area, P.src_rate[lp]));
            arr_time = ns_from_seconds(t_next);
        }
        A = arr_time;
    }
    if (NX.next_time != nullptr) NX.next_time[lp] = A;
    X.A[lp] = A; X.seqA[lp] = 0; X.crtA[lp] = start_ns; X.arr_k[lp] = arr_k; X.arr_time[lp] = arr_time;
    X.svc_k[lp] = 0; X.seq[lp] = 1; X.buf[lp] = 0; X.active[lp] = 0;
    X.generated[lp] = 0; X.accepted[lp] = 0; X.dropped[lp] = 0; X.completed[lp] = 0; X.rejected[lp] = 0;
    X.started[lp] = 0; X.received[lp] = 0; X.sink_w[lp] = 0; X.total_service[lp] = 0.0;
    X.q[lp] = 0; X.grp_time[lp] = start_ns; X.last_time[lp] = start_ns; X.events[lp] = 0;
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = kInfNs; X.seqD[(size_t)i * n + lp] = 0; X.crtD[(size_t)i * n + lp] = start_ns;
        X.svc_s[(size_t)i * n + lp] = 0.0; X.crt[(size_t)i * n + lp] = 0;
    }
    for (int k = 0; k < 11; ++k) X.ev_kind[(size_t)k * n + lp] = 0;
    if constexpr (!PF) return;
    if (X.XA != nullptr) {   // the LP's further Sources: each draws its first arrival from start_ns like the first one
        for (int j = 0; j < kMaxXSrc; ++j) {
            const size_t o = (size_t)j * n + lp;
            int64_t XA = kInfNs, x_arr = start_ns;
            uint64_t x_k = 0;
            const uint32_t xk = P.xsrc_kind[o];
            if (xk != 0) {
                double area = 1.0;
                if (xk == 1) {
                    Stream s;
                    s.init(P.seed[lp], xsrc_stream_id(P.stream_base[lp], j), 0);
                    area = exp1_from_uniform(s.next_uniform());
                    x_k = 1;
                }
                x_arr = ns_from_seconds(__dadd_rn(seconds_from_ns(start_ns), __ddiv_rn(area, P.xsrc_rate[o])));
                XA = x_arr;
            }
            // (default stamps, replaced by the prologue's true sort indices: after the first Source, before the probes)
            X.XA[o] = XA; X.seqX[o] = 1u + (uint32_t)j; X.crtX[o] = start_ns; X.x_arr[o] = x_arr; X.x_n[o] = 0; X.x_k[o] = x_k;
            if (NX.next_time != nullptr && XA < NX.next_time[lp]) NX.next_time[lp] = XA;   // network engine: first pending event
        }
    }
    if (X.PA != nullptr) {   // probes start after the sources (core/simulation.py:156-160): first tick from start_ns
        uint32_t stamp = 1 + kMaxXSrc;
        for (int j = 0; j < kMaxProbes; ++j) {
            const size_t o = (size_t)j * n + lp;
            int64_t PA = kInfNs, p_arr = start_ns;
            if (P.probe_metric[o] != kProbeNone) {
                if constexpr (PF) p_arr = first_probe_tick(P.probe_rate[o], start_ns, lp);
                PA = p_arr;
            }
            if (NX.next_time != nullptr && PA < NX.next_time[lp]) NX.next_time[lp] = PA;   // network engine: first pending event
            X.PA[o] = PA; X.seqP[o] = stamp++; X.crtP[o] = start_ns; X.p_arr[o] = p_arr; X.p_n[o] = 0;
        }
        X.ev_probe[lp] = 0; X.ev_probe[(size_t)n + lp] = 0;
        X.seq[lp] = 1 + kMaxXSrc + kMaxProbes;
        if (P.sched_off != nullptr) {
            X.sched_i[lp] = P.sched_off[lp];
            if (NX.next_time != nullptr && P.sched_off[lp] < P.sched_off[lp + 1]) {   // network engine: first pending event
                const int64_t s0 = P.sched_t[P.sched_off[lp]];
                if (s0 < NX.next_time[lp]) NX.next_time[lp] = s0;
            }
        }
    }
}

// The hot kernel: every LP advances to end_ns (== Simulation._execute_until for its events), then the
// one-event overshoot is applied (per LP in REPLICAS mode; to the globally first event in SINGLE mode,
// elected across workgroups with a last-block reduction).
//
// PC (producer / consumer, the <1, false> instantiation): the per-request work has two halves of about equal cost -- the
// stream values (Philox4x32-10, hs_log, the constant-divisor quotients, the ns truncations: independent from request to
// request) and the serial request step (the ns recursion, the Lindley recursion, event counting, the log appends).  With one
// LP per lane a 65 536-LP grid is ONE wavefront per SIMD, so inside one wavefront the two halves only alternate and the
// SIMD idles on every dependent-instruction latency (measured: VALU busy 47 % of the wave's cycles).  PC launches 512
// threads per 256 LPs: wavefronts 0-3 run the request step for LP `tid`, wavefronts 4-7 -- one on each SIMD, next to its
// consumer -- produce the same LP's stream values into the same LDS rings, and the SIMD interleaves the two instruction
// streams.  The rings become single-producer / single-consumer queues with 16-bit produced / consumed counters in LDS
// (release / acquire at workgroup scope); values are pure functions of (stream, index), so who computes them is invisible.
template <int C, bool PF, bool PC = false, bool UNI = false>
__global__ void __launch_bounds__(PC ? 2 * kBlock : kBlock) hs_station_run(StationParams P, StationState X, RecordLogs L, Totals *tot,
                                                         Candidate *cands, int n, int64_t end_ns, int mode, int flags) {
    static_assert(!PC || (C == 1 && !PF), "producer / consumer waves serve the request-order loop of <1, false>");
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ double ring_a[kRing][kBlock];    // pre-drawn arrival increments, one column per LP
    __shared__ double ring_s[kRing][kBlock];    // pre-drawn service times
    __shared__ unsigned long long red[12];
    __shared__ long long red_time;
    __shared__ int red_flags[2];
    __shared__ Candidate wave_c[kBlock / 64];
    __shared__ int is_last;
    __shared__ uint32_t pc_prod[PC ? kBlock : 1];   // values produced so far: arrival count | service count << 16 (mod 2^16)
    __shared__ uint32_t pc_cons[PC ? kBlock : 1];   // values consumed so far, same packing
    __shared__ int pc_done[kBlock / 64];            // consumer wavefront w has left the request-order loop

    const int tid = PC ? (int)(threadIdx.x & (kBlock - 1)) : (int)threadIdx.x;
    const bool producer = PC && threadIdx.x >= kBlock;
    const int lp = blockIdx.x * kBlock + tid;
    const bool live = lp < n && !producer;
    if (threadIdx.x < 12) red[threadIdx.x] = 0;
    if (threadIdx.x == 0) { red_time = INT64_MIN; red_flags[0] = 0; red_flags[1] = 0; }
    if constexpr (PC) {
        if (!producer) { pc_prod[tid] = 0; pc_cons[tid] = 0; if ((tid & 63) == 0) pc_done[tid >> 6] = 0; }
    }
    const long long cur = tot->cur_time;   // SINGLE: Simulation._current_time (written by the previous launch)
    __syncthreads();

    Station<C, PF, UNI> S;
    Candidate mine;
    mine.valid = 0; mine.t = kInfNs; mine.t_created = 0; mine.lp = lp; mine.rank = lp;
    if constexpr (PC) {
        if (producer) {
            // ---- producer wavefront: stream values for LP `tid`, as long as its consumer is in the request-order loop
            const int w = tid >> 6;
            bool wants_a = false, wants_s = false;
            uint64_t gen_a = 0, gen_s = 0;          // absolute index of the next value to generate
            uint32_t prod_a = 0, prod_s = 0;        // values written to the rings so far
            int slot_a = 0, slot_s = 0;
            if (lp < n) {
                S.tid = tid;
                S.rate = P.src_rate[lp];
                S.svc_lambda = __ddiv_rn(1.0, P.svc_mean[lp]);
                S.prof.kind = kProfConstant;
                S.init_streams(P.seed[lp], P.stream_base[lp], X.arr_k[lp], X.svc_k[lp], ring_a, ring_s);
                wants_a = P.src_kind[lp] == 1 && X.A[lp] != kInfNs;
                wants_s = P.svc_kind[lp] == 0;
                gen_a = S.arr_k; gen_s = S.svc_k;
            }
            for (;;) {
                if (__hip_atomic_load(&pc_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                const uint32_t c = __hip_atomic_load(&pc_cons[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t out_a = (prod_a - (c & 0xffffu)) & 0xffffu, out_s = (prod_s - (c >> 16)) & 0xffffu;
                const bool do_a = wants_a && out_a + (uint32_t)kRefill <= (uint32_t)kRing;
                const bool do_s = wants_s && out_s + (uint32_t)kRefill <= (uint32_t)kRing;
                if (!__any(do_a || do_s)) { __builtin_amdgcn_s_sleep(HS_PC_SLEEP_P); continue; }
                if (__any(do_a)) {
                    if (do_a) {
                        const uint64_t b0 = gen_a >> 1;
                        const bool odd = (gen_a & 1) != 0;
#pragma unroll
                        for (int i = 0; i < kRefill / 2; ++i) {
                            const uint64_t b = b0 + (uint64_t)i;
                            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), S.asid0, S.asid1, S.key0, S.key1);
                            const double v0 = S.arr_value(res53(o.x, o.y)), v1 = S.arr_value(res53(o.z, o.w));
                            if (!(i == 0 && odd)) { ring_a[slot_a][tid] = v0; slot_a = slot_a + 1 == kRing ? 0 : slot_a + 1; ++prod_a; ++gen_a; }
                            ring_a[slot_a][tid] = v1; slot_a = slot_a + 1 == kRing ? 0 : slot_a + 1; ++prod_a; ++gen_a;
                        }
                    }
                }
                if (__any(do_s)) {
                    if (do_s) {
                        const uint64_t b0 = gen_s >> 1;
                        const bool odd = (gen_s & 1) != 0;
#pragma unroll
                        for (int i = 0; i < kRefill / 2; ++i) {
                            const uint64_t b = b0 + (uint64_t)i;
                            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), S.ssid0, S.ssid1, S.key0, S.key1);
                            const double v0 = S.svc_value(res53(o.x, o.y)), v1 = S.svc_value(res53(o.z, o.w));
                            if (!(i == 0 && odd)) { ring_s[slot_s][tid] = v0; slot_s = slot_s + 1 == kRing ? 0 : slot_s + 1; ++prod_s; ++gen_s; }
                            ring_s[slot_s][tid] = v1; slot_s = slot_s + 1 == kRing ? 0 : slot_s + 1; ++prod_s; ++gen_s;
                        }
                    }
                }
                // the values first, then the counters (release: the ring stores are complete before the count moves)
                __hip_atomic_store(&pc_prod[tid], (prod_a & 0xffffu) | (prod_s << 16), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (live) {
        load_station<C, PF, UNI>(S, P, X, L, lp, n, qmem, ring_a, ring_s, tid);
        S.force_general = (flags & 1) != 0;
    }
    const bool frozen = (mode == HS_MODE_REPLICAS) ? (live && S.last_time > end_ns) : (cur > end_ns);
    bool pre_group = false;
    bool event_order = true;
    if (live && !frozen && S.qn > 0 && S.grp_time <= end_ns) {   // finish a group a previous window stopped inside
        S.run_group_general(S.grp_time);
        S.last_time = S.grp_time;
        pre_group = true;
    }
    if constexpr (C == 1) {
        bool bail_reload = false;
        if (!producer) {
            // (1) request-order loop (hs_station.hpp): one whole request per iteration.  Uniform loops: the
            // wavefront iterates until its slowest lane is done, finished lanes are predicated off.
            const bool elig = live && !frozen && !pre_group && S.qn == 0 && S.req_eligible();
            typename Station<C, PF, UNI>::ReqCursor rc;
            rc.bail = false; rc.done = true;
            if (elig) S.req_begin(rc, end_ns);   // (touches the LP's statistics: eligible lanes only)
#ifdef HS_CYCLES   // tools/cycles.py: where the request-order loop spends its time (never defined in the shipped library)
            unsigned long long cyc_top = 0, cyc_step = 0, n_it = 0;
#endif
            uint32_t seen = 0;                   // PC: produced counters as last read (arrival | service << 16)
            const uint64_t ak0 = S.arr_k, sk0 = S.svc_k;
            for (;;) {
                const bool act = elig && !rc.bail && !rc.done;
                if (!__any(act)) break;
#ifdef HS_CYCLES
                const unsigned long long c0 = __builtin_readcyclecounter();
#endif
                if constexpr (PC) {
                    // every lane that may consume needs one value of each stream; the producer runs ahead, so this rarely waits
                    const bool need_a = act && S.src_kind == 1 && S.A != kInfNs, need_s = act && S.svc_kind == 0;
                    const uint32_t ca = (uint32_t)(S.arr_k - ak0) & 0xffffu, cs = (uint32_t)(S.svc_k - sk0) & 0xffffu;
                    unsigned spins = 0;
                    while (__any((need_a && (seen & 0xffffu) == ca) || (need_s && (seen >> 16) == cs))) {
                        seen = __hip_atomic_load(&pc_prod[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__any((need_a && (seen & 0xffffu) == ca) || (need_s && (seen >> 16) == cs))) __builtin_amdgcn_s_sleep(HS_PC_SLEEP_C);
                        if (++spins > (1u << 22)) {      // bounded (~0.1 s): report instead of hanging the device
                            if (elig) { S.qoverflow = 1; rc.done = true; }
                            break;
                        }
                    }
                    if (spins > (1u << 22)) continue;
                } else {
                    S.top_up(act);             // wave-level refill of the pre-drawn stream values
                }
#ifdef HS_CYCLES
                const unsigned long long c1 = __builtin_readcyclecounter();
#endif
                S.req_step(rc, act);
                if constexpr (PC) {
                    __hip_atomic_store(&pc_cons[tid], ((uint32_t)(S.arr_k - ak0) & 0xffffu) | ((uint32_t)(S.svc_k - sk0) << 16),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    seen = __hip_atomic_load(&pc_prod[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#ifdef HS_CYCLES
                const unsigned long long c2 = __builtin_readcyclecounter();
                cyc_top += c1 - c0; cyc_step += c2 - c1; ++n_it;
#endif
            }
#ifdef HS_CYCLES
            if ((tid & 63) == 0) {
                atomicAdd(&tot->dbg[0], cyc_top); atomicAdd(&tot->dbg[1], cyc_step);
                atomicAdd(&tot->dbg[2], n_it); atomicAdd(&tot->dbg[3], 1ull);
            }
#endif
            if constexpr (PC) {
                if ((tid & 63) == 0) __hip_atomic_store(&pc_done[tid >> 6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (elig && !rc.bail) { S.req_finish(rc); event_order = false; }
            bail_reload = elig && rc.bail;     // same-timestamp hazard: start over in event order
        }
        if constexpr (PC) __syncthreads();     // the producers have left their loop: the rings belong to the consumers again
        if (!producer) {
            if constexpr (PC) {
                // lanes that go on in event order refill their own rings from the consumed counts (the producer's leftovers
                // are values of later indices; dropping them changes nothing)
                S.ra.head = 0; S.ra.n = 0; S.rs.head = 0; S.rs.n = 0;
            }
            if (bail_reload) {
                load_station<C, PF, UNI>(S, P, X, L, lp, n, qmem, ring_a, ring_s, tid);
                S.force_general = (flags & 1) != 0;
            }
        }
        // (2) event-order loop for whatever (1) does not cover
        if (!producer) {
            for (;;) {
                const int64_t t = live && !frozen && S.qn == 0 ? S.next_time() : kInfNs;
                const bool act = live && !frozen && S.qn == 0 && event_order && t <= end_ns;   // t == kInfNs: nothing pending
                if (!__any(act)) break;
                S.top_up(act);
                S.step_c1(t, act);
            }
        }
    } else {
        if (live && !frozen && S.qn == 0) {
            for (;;) {
                S.top_up();
                const int64_t t = S.next_time();
                if (t > end_ns) break;         // also ends on kInfNs: nothing pending
                S.run_group(t);
            }
        }
    }
    if (live) {
        if (!frozen) {
            if (mode == HS_MODE_REPLICAS) overshoot_one<C, PF, UNI>(S);
            else {
                mine = make_candidate<C, PF, UNI>(S);
                mine.rank = cand_rank(P, lp, n, mine.pad);
            }
        }
        store_station<C, PF, UNI>(S, X, lp, n);
    }

    // ---- workgroup reduction of the per-run deltas -> engine totals
    unsigned vals[10];
#pragma unroll
    for (int k = 0; k < 8; ++k) vals[k] = live ? S.ev[k] : 0u;
    // completed / received deltas are the continuation / sink event counts
    vals[8] = vals[6]; vals[9] = vals[7];
    if (!producer) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const unsigned s = wave_sum<unsigned>(vals[k]);
            if ((tid & 63) == 0 && s) atomicAdd(&red[k], (unsigned long long)s);
        }
    }
    if (live) {
        atomicMax(&red_time, (long long)S.last_time);
        if (S.overflow) red_flags[0] = 1;
        if (S.qoverflow) red_flags[1] = 1;
    }
    if (mode == HS_MODE_SINGLE && !producer) {
        const Candidate w = wave_min_cand(mine);
        if ((tid & 63) == 0) wave_c[tid >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8 && red[threadIdx.x]) atomicAdd(&tot->ev[threadIdx.x], red[threadIdx.x]);
    if (threadIdx.x == 8 && red[8]) atomicAdd(&tot->completed, red[8]);
    if (threadIdx.x == 9 && red[9]) atomicAdd(&tot->received, red[9]);
    if (threadIdx.x == 10) {
        if (red_time != INT64_MIN) atomicMax(&tot->final_time, red_time);
        if (red_flags[0]) atomicOr(&tot->overflow, 1);
        if (red_flags[1]) atomicOr(&tot->qoverflow, 1);
    }
    if constexpr (PF) {      // probe events straight to the totals (rare LPs)
        if (live && S.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)S.evp[0]);
        if (live && S.evp[1]) atomicAdd(&tot->ev[14], (unsigned long long)S.evp[1]);
    }
    if (mode != HS_MODE_SINGLE) return;

    // ---- SINGLE mode: elect the globally first event beyond end_ns (last-block pattern)
    if (threadIdx.x == 0) {
        Candidate b = wave_c[0];
        for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
        cands[blockIdx.x] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // state + candidate visible device-wide
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = atomicAdd(&tot->done, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    Candidate best;
    best.valid = 0; best.t = kInfNs; best.t_created = 0; best.lp = 0; best.rank = 0;
    if (!producer) {
        for (int b = tid; b < (int)gridDim.x; b += kBlock) {
            Candidate c;
            c.t = __hip_atomic_load(&cands[b].t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c.t_created = __hip_atomic_load(&cands[b].t_created, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c.lp = __hip_atomic_load(&cands[b].lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c.rank = __hip_atomic_load(&cands[b].rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c.valid = __hip_atomic_load(&cands[b].valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cand_less(c, best)) best = c;
        }
        best = wave_min_cand(best);
        if ((tid & 63) == 0) wave_c[tid >> 6] = best;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Candidate b = wave_c[0];
        for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
        long long new_cur = __hip_atomic_load(&tot->final_time, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur <= end_ns && b.valid) {
            Station<C, PF> W;
            load_station<C, PF>(W, P, X, L, b.lp, n, qmem, ring_a, ring_s, 0);
            W.force_general = false;
            overshoot_one<C, PF>(W);
            store_station<C, PF>(W, X, b.lp, n);
            for (int k = 0; k < 8; ++k) if (W.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)W.ev[k]);
            if (W.ev[6]) atomicAdd(&tot->completed, (unsigned long long)W.ev[6]);
            if (W.ev[7]) atomicAdd(&tot->received, (unsigned long long)W.ev[7]);
            if constexpr (PF) {
                if (W.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)W.evp[0]);
                if (W.evp[1]) atomicAdd(&tot->ev[14], (unsigned long long)W.evp[1]);
            }
            if (W.overflow) atomicOr(&tot->overflow, 1);
            new_cur = b.t;
            atomicMax(&tot->final_time, new_cur);
        }
        if (cur <= end_ns) tot->cur_time = new_cur;
        tot->done = 0;   // self-resetting ticket
    }
}

// ---------------------------------------------------------------------------------------------
// The prologue (hs_exact.hpp): lane 0 runs the reference's heap loop until the run has constructed as many events as
// were constructed before it; then the wavefront hands the state to the parallel engine.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) hs_exact_run(StationParams P, NetParams NP, StationState X, NetState NX, RecordLogs L,
                                                   Totals *tot, XState *xs, XInit I, int n, int C, int net, int n_links,
                                                   int64_t start_ns, int64_t end_ns) {
    __shared__ int s_hand;
    if (I.per_lp) {
        // HS_MODE_REPLICAS: every LP is its own Simulation -- one lane per LP, each with its own slice of the buffers
        const int lp = blockIdx.x * 64 + threadIdx.x;
        if (lp >= n) return;
        XState &S = xs[1 + lp];
        if (S.phase == 0 && S.heap == nullptr) {
            const XState &B = xs[0];                       // slice the shared buffers
            S.heap = B.heap + (size_t)lp * I.heap_cap_lp; S.heap_cap = I.heap_cap_lp; S.heap_len = 0;
            S.qhead = B.qhead; S.qtail = B.qtail;
            S.pnext = B.pnext + (size_t)lp * I.pool_cap_lp; S.pidx = B.pidx + (size_t)lp * I.pool_cap_lp;
            S.pool_n = 0; S.pool_cap = I.pool_cap_lp;
            S.init_t = B.init_t + (size_t)lp * I.init_cap_lp;
            S.err = 0; S.processed = 0;
        }
        // (the pool's list cells are addressed relative to the LP's slice: qhead / qtail hold slice-local indices)
        if (exact_loop(P, NP, X, NX, L, tot, S, I, n, C, false, start_ns, end_ns, lp)) {
            X.seq[lp] = (uint32_t)S.G;
            S.phase = 2;
        }
        return;
    }
    if (threadIdx.x == 0) s_hand = exact_loop(P, NP, X, NX, L, tot, *xs, I, n, C, net != 0, start_ns, end_ns) ? 1 : 0;
    __syncthreads();
    if (!s_hand) return;
    // every LP's creation counter continues from the global one: whatever it constructs from now on follows every
    // pending event (pre-run events keep their own, smaller indices as stamps)
    const uint32_t g = (uint32_t)xs->G;
    for (int lp = threadIdx.x; lp < n; lp += 64) X.seq[lp] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (net) {
            // requests in transit: the pending link continuations become messages in the destination's bag
            int over = 0;
            for (int64_t i = 0; i < xs->heap_len; ++i) {
                const XEvent &e = xs->heap[i];
                if (e.code != XE_LINKCONT) continue;
                const int b = NX.bag_cnt[e.lp];
                if (b >= NX.bag_cap) { over = 1; continue; }
                const size_t d = (size_t)e.lp * NX.bag_cap + b;
                NX.bag_t[d] = e.t; NX.bag_ts[d] = e.ts; NX.bag_cr[d] = e.cr; NX.bag_link[d] = (int32_t)e.aux;
                NX.bag_cnt[e.lp] = b + 1;
            }
            if (over) atomicOr(&tot->overflow, 2);
            if (NX.aq_tail != nullptr)                    // asynchronous engine: the link queues continue behind them
                for (int l = 0; l < n_links; ++l) {
                    NX.aq_ea[l] = pk_pack(start_ns, (unsigned long long)NX.link_sent[l], NX.pk_base);
                    NX.aq_head[l] = (unsigned long long)NX.link_sent[l];
                }
        }
        xs->phase = 2;
    }
    __syncthreads();
    if (net) {
        for (int lp = threadIdx.x; lp < n; lp += 64) {
            int64_t t = X.A[lp];
            for (int i = 0; i < C; ++i) { const int64_t d = X.D[(size_t)i * n + lp]; t = d < t ? d : t; }
            if (X.PA != nullptr) {
                for (int j = 0; j < kMaxProbes; ++j) {
                    const size_t o = (size_t)j * n + lp;
                    if (P.probe_metric[o] != kProbeNone && X.PA[o] < t) t = X.PA[o];
                }
                if (P.sched_off != nullptr && X.sched_i[lp] < P.sched_off[lp + 1]) {
                    const int64_t sa = P.sched_t[X.sched_i[lp]];
                    t = sa < t ? sa : t;
                }
            }
            if (X.XA != nullptr)                              // pending ticks of the LP's further Sources
                for (int j = 0; j < kMaxXSrc; ++j) { const int64_t xa = X.XA[(size_t)j * n + lp]; t = xa < t ? xa : t; }
            const int bn = NX.bag_cnt[lp];
            for (int i = 0; i < bn; ++i) { const int64_t bt = NX.bag_t[(size_t)lp * NX.bag_cap + i]; t = bt < t ? bt : t; }
            NX.next_time[lp] = t;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// network engine (hs_netstation.hpp): one launch per conservative time window
// ---------------------------------------------------------------------------------------------
namespace {

template <int C, bool FAST = false, bool PF = !FAST, bool UNI = false>
__device__ __forceinline__ void load_net(NetStation<C, FAST, PF, UNI> &S, const StationParams &P, const NetParams &NP,
                                         const StationState &X, const NetState &NX, const RecordLogs &L, int lp, int n,
                                         uint8_t (*qmem)[kBlock], int64_t (*enqpay)[kBlock], int tid, int send_idx,
                                         const ShardCtl &SC) {
    S.lp = lp; S.n = n;
    S.sc = &SC; S.sent_min = kInfNs; S.sent_async = false;
    S.src_kind = P.src_kind[lp]; S.svc_kind = P.svc_kind[lp]; S.egress = NP.egress[lp];
    S.conc = P.conc[lp]; S.rt0 = NP.rt0[lp]; S.rt1 = NP.rt1[lp]; S.rt2 = NP.rt2[lp]; S.rt3 = NP.rt3[lp]; S.rtk = NP.rt_cnt[lp];
    S.link_of = NP.link_of[lp];
    S.rate = P.src_rate[lp];
    const double mean = P.svc_mean[lp];
    S.svc_lambda = __ddiv_rn(1.0, mean);
    S.svc_const_s = seconds_from_ns(ns_from_seconds(mean));
    S.svc_const_ns = ns_from_seconds(S.svc_const_s);
    S.stop_ns = P.src_stop[lp]; S.qcap = P.qcap[lp];
    S.seed = P.seed[lp]; S.route_base = NP.route_base[lp];
    S.A = X.A[lp]; S.seqA = X.seqA[lp]; S.crtA = X.crtA[lp]; S.arr_time = X.arr_time[lp];
    S.buf = X.buf[lp]; S.active = X.active[lp]; S.seq = X.seq[lp];
    S.generated = X.generated[lp]; S.accepted = X.accepted[lp]; S.dropped = X.dropped[lp];
    S.completed = X.completed[lp]; S.rejected = X.rejected[lp]; S.started = X.started[lp];
    S.received = X.received[lp]; S.routed = NX.routed[lp];
    S.total_service = X.total_service[lp];
    S.last_time = X.last_time[lp];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        S.D[i] = X.D[(size_t)i * n + lp]; S.seqD[i] = X.seqD[(size_t)i * n + lp];
        S.crtD[i] = X.crtD[(size_t)i * n + lp]; S.svc_s[i] = X.svc_s[(size_t)i * n + lp];
        S.crt[i] = X.crt[(size_t)i * n + lp];
    }
    const uint64_t base = P.stream_base[lp];
    S.arr.init(S.seed, stream_id(base, kStreamArrival), X.arr_k[lp]);
    S.svc.init(S.seed, stream_id(base, kStreamService), X.svc_k[lp]);
    S.rte.init(S.seed, stream_id(S.route_base, kStreamRoute), NX.route_k[lp]);
#pragma unroll
    for (int k = 0; k < 11; ++k) S.ev[k] = 0;
    S.adm = L.adm + lp;
    S.sink_t = L.sink_t + lp;
    S.sink_created = L.sink_created + lp;
    S.cap = L.cap; S.ls = n;
    S.overflow = 0; S.qoverflow = 0; S.bagoverflow = 0;
    S.np = &NP; S.ns = &NX; S.send_idx = send_idx;
    S.tid = tid;
    S.inc_const = __ddiv_rn(1.0, S.rate);
    S.n_probes = 0; S.evp[0] = S.evp[1] = 0; S.pcap = 0; S.probe_t = nullptr; S.probe_v = nullptr;
#pragma unroll
    for (int j = 0; j < kMaxProbes; ++j) { S.p_metric[j] = kProbeNone; S.PA[j] = kInfNs; S.seqP[j] = 0; S.crtP[j] = 0; S.p_arr[j] = 0; S.p_n[j] = 0; S.p_rate[j] = 1.0; }
    S.prof_kind = kProfConstant; S.prof_p0 = S.prof_p1 = S.prof_p2 = S.prof_p3 = 0.0;
    S.SA = kInfNs; S.sc_i = S.sc_end = 0; S.sc_t = P.sched_t; S.sc_idx = P.sched_idx;
    S.n_xsrc = 0; S.x_base = P.stream_base[lp]; S.xs_min = kInfNs; S.xp = &P; S.xx = &X; S.x_n_lp = n;
    if constexpr (PF) {
        if (P.xsrc_kind != nullptr) {      // further Sources of this station's Server (their state stays in X)
            for (int j = 0; j < kMaxXSrc; ++j) {
                const size_t o = (size_t)j * n + lp;
                if (P.xsrc_kind[o] != 0) { S.n_xsrc = j + 1; const int64_t a = X.XA[o]; S.xs_min = a < S.xs_min ? a : S.xs_min; }
            }
        }
        if (P.sched_off != nullptr) {
            S.sc_i = X.sched_i[lp]; S.sc_end = P.sched_off[lp + 1];
            S.SA = S.sc_i < S.sc_end ? P.sched_t[S.sc_i] : kInfNs;
        }
        if (P.prof_kind[lp] != kProfConstant) {
            S.prof_kind = P.prof_kind[lp];
            S.prof_p0 = P.prof_p[lp]; S.prof_p1 = P.prof_p[(size_t)n + lp]; S.prof_p2 = P.prof_p[(size_t)2 * n + lp];
            S.prof_p3 = P.prof_p[(size_t)3 * n + lp];
        }
        if (X.PA != nullptr) {      // probes on a network (PF instantiations of both engines)
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) {
                const size_t o = (size_t)j * n + lp;
                S.p_metric[j] = P.probe_metric[o];
                if (S.p_metric[j] != kProbeNone) {
                    S.n_probes = j + 1;
                    S.p_rate[j] = P.probe_rate[o];
                    S.PA[j] = X.PA[o]; S.seqP[j] = X.seqP[o]; S.crtP[j] = X.crtP[o]; S.p_arr[j] = X.p_arr[o]; S.p_n[j] = X.p_n[o];
                }
            }
            S.probe_t = L.probe_t + lp; S.probe_v = L.probe_v + lp; S.pcap = L.pcap;
        }
    }
    S.ha = S.na = S.hs_ = S.nsv = S.hj = S.nj = S.rn = 0; S.rbits = 0; S.fl_link = -1; S.fi_link = -1; S.fi_packets = 0;
    S.fl_remote = false; S.fi_head = 0; S.win_hi = S.started;
    S.bh = 0; S.tail_hint = 0ull;
    S.presend = false; S.early_upto = S.completed; S.D_pre = S.last_time; S.fl_q = 0; S.end_ns = kInfNs;
    S.bag_n = NX.bag_cnt[lp];
    if constexpr (FAST) {
        // (S.fl was set by the caller.)  The bag moves into LDS for the lifetime of the kernel ...
        int nb = S.bag_n;
        if (nb > kLBag) { S.bagoverflow = 1; nb = kLBag; }             // (a fresh run starts with empty bags)
        S.bag_n = 0; S.bh = 0; S.bmin = kInfNs;
        for (int i = 0; i < nb; ++i) {                                 // ... sorted by arrival time (NetStation::bag_insert)
            const size_t b = (size_t)lp * NX.bag_cap + i;
            S.bag_insert(NX.bag_t[b], NX.bag_ts[b], NX.bag_cr[b], NX.bag_link[b]);
        }
        // ... and the LP's outgoing link into registers when there is exactly one
        int32_t l = -1;
        if (S.egress == EG_LINK) l = S.link_of;
        else if (S.egress == EG_ROUTER) {          // exactly one link among the router's targets
            int cnt = 0;
            for (int k = 0; k < S.rtk; ++k) { const int32_t t = S.rt_target(k); if (t >= 0) { l = t; ++cnt; } }
            if (cnt != 1) l = -1;
        }
        if (l >= 0) {
            S.fl_link = l; S.fl_dst = NP.link_dst[l]; S.fl_jit = NP.link_jit_kind[l];
            S.fl_remote = SC.wend_slots != nullptr && SC.link_rank[l] != SC.rank;
            S.fl_delay0 = seconds_from_ns(ns_from_seconds(NP.link_lat_min[l]));
            S.fl_lam = __ddiv_rn(1.0, NP.link_jit_mean[l]);
            S.fl_loss = NP.link_loss[l];
            S.fl_in = NX.link_in[l]; S.fl_sent = NX.link_sent[l];
            S.fl_q = (NX.aq_tail != nullptr && !S.fl_remote) ? (int64_t)pk_tail(NX.aq_ea[l], NX.aq_head[l]) : S.fl_sent;
            S.jit.init(S.seed, stream_id(NP.link_base[l], kStreamLink), NX.link_k[l]);
        }
        if (C == 1 && l >= 0 && S.conc == 1 && S.fl_loss == 0.0 && NX.early_upto != nullptr) {
            S.presend = true;                                   // departures are pre-sent (hs_netstation.hpp `early_upto`)
            const int64_t eu = NX.early_upto[lp];
            if (eu > S.completed) { S.early_upto = eu; S.D_pre = NX.d_pre[lp]; }
        }
        if (NP.in_off[lp + 1] - NP.in_off[lp] == 1) {          // ... and the counter of its only incoming link
            S.fi_link = NP.in_links[NP.in_off[lp]];
            S.fi_packets = NX.link_packets[S.fi_link];
            S.fi_head = NX.aq_head[S.fi_link];
        }
        // the created_at window: the next kNRing requests to start, as far as they are admitted (read from the log)
        for (int i = 0; i < kNRing; ++i) {
            const int64_t k = S.started + i;
            if (k < S.accepted) { S.fl.crc[k & (kNRing - 1)][tid] = k < S.cap ? S.adm[k * S.ls] : 0; S.win_hi = k + 1; }
        }
    }
    S.bmin = S.bag_scan_min();
    S.qmem = qmem; S.enqpay = enqpay; S.qh = 0; S.qn = 0; S.ph = 0; S.pn = 0;
}

template <int C, bool FAST = false, bool PF = !FAST, bool UNI = false>
__device__ __forceinline__ void store_net(NetStation<C, FAST, PF, UNI> &S, const StationState &X, const NetState &NX, int lp, int n) {
    X.A[lp] = S.A; X.seqA[lp] = S.seqA; X.crtA[lp] = S.crtA; X.arr_time[lp] = S.arr_time;
    X.buf[lp] = S.buf; X.active[lp] = S.active; X.seq[lp] = S.seq;
    X.generated[lp] = S.generated; X.accepted[lp] = S.accepted; X.dropped[lp] = S.dropped;
    X.completed[lp] = S.completed; X.rejected[lp] = S.rejected; X.started[lp] = S.started;
    X.received[lp] = S.received; X.sink_w[lp] = S.received; NX.routed[lp] = S.routed;
    X.total_service[lp] = S.total_service;
    X.last_time[lp] = S.last_time;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = S.D[i]; X.seqD[(size_t)i * n + lp] = S.seqD[i];
        X.crtD[(size_t)i * n + lp] = S.crtD[i]; X.svc_s[(size_t)i * n + lp] = S.svc_s[i];
        X.crt[(size_t)i * n + lp] = S.crt[i];
    }
    // draws CONSUMED (pre-drawn values still in the FAST rings are dropped: pure functions of the index)
    X.arr_k[lp] = S.arr_consumed(); X.svc_k[lp] = S.svc_consumed(); NX.route_k[lp] = S.rte_consumed();
    if constexpr (FAST) {
        for (int i = 0; i < S.bag_n; ++i) {
            const size_t b = (size_t)lp * NX.bag_cap + i;
            NX.bag_t[b] = S.bg_t(i); NX.bag_ts[b] = S.bg_ts(i); NX.bag_cr[b] = S.bg_cr(i); NX.bag_link[b] = S.bg_link(i);
        }
        if (S.fl_link >= 0) {
            NX.link_in[S.fl_link] = S.fl_in; NX.link_sent[S.fl_link] = S.fl_sent;
            NX.link_k[S.fl_link] = S.jit.k - (uint64_t)S.nj;
        }
        if (S.fi_link >= 0) NX.link_packets[S.fi_link] = S.fi_packets;
        if (S.presend) { NX.early_upto[lp] = S.early_upto; NX.d_pre[lp] = S.D_pre; }
    }
    NX.bag_cnt[lp] = S.bag_n;
    NX.next_time[lp] = S.next_time();
    uint32_t tot = 0;
    if constexpr (PF) {
        if (X.PA != nullptr) {
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j < S.n_probes) {
                const size_t o = (size_t)j * n + lp;
                X.PA[o] = S.PA[j]; X.seqP[o] = S.seqP[j]; X.crtP[o] = S.crtP[j]; X.p_arr[o] = S.p_arr[j]; X.p_n[o] = S.p_n[j];
            }
            X.ev_probe[lp] += S.evp[0]; X.ev_probe[(size_t)n + lp] += S.evp[1];
            tot += S.evp[0] + S.evp[1];
            if (S.sc_t != nullptr && X.sched_i != nullptr) X.sched_i[lp] = S.sc_i;
        }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) { if (S.ev[k]) X.ev_kind[(size_t)k * n + lp] += S.ev[k]; tot += S.ev[k]; }
    X.events[lp] += tot;
}

}  // namespace

// One conservative window: every LP merges the messages sent to it during the previous window, then
// processes all of its timestamp groups with time <= wend.  flags bit 0: force the general path;
// bit 1: FINAL launch (after the last window): elect and process the single overshoot event.
template <int C>
__global__ void __launch_bounds__(kBlock) hs_net_window(StationParams P, NetParams NP, StationState X, NetState NX,
                                                        RecordLogs L, Totals *tot, Candidate *cands, int n,
                                                        int64_t wend, int win, int flags, ShardCtl SC) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ int64_t enqpay[kEnqPay][kBlock];
    __shared__ unsigned long long red[14];
    __shared__ long long red_time;
    __shared__ int red_flags[3];
    __shared__ Candidate wave_c[kBlock / 64];
    __shared__ int is_last;

    const int tid = threadIdx.x;
    const int lp = blockIdx.x * kBlock + tid;
    const bool live = lp < n;
    const bool final_launch = (flags & 2) != 0;
    const long long cur0 = tot->cur_time;     // beyond end_ns only when the prologue (hs_exact.hpp) already ran the whole run
    const int merge_idx = (win + 1) & 1, send_idx = win & 1;
    __shared__ long long red_gvt;
    if (tid < 14) red[tid] = 0;
    if (tid == 0) { red_time = INT64_MIN; red_flags[0] = red_flags[1] = red_flags[2] = 0; red_gvt = kInfNs; }
    if (SC.wend_slots != nullptr) {
        // sharded network: every workgroup derives the same window end from the global virtual time
        const int64_t prev = SC.wend_slots[(win + 1) & 1];
        const int64_t gvt = *SC.gvt_in;
        const int64_t base = gvt > prev + 1 ? gvt : prev + 1;
        wend = (base > SC.end_ns - (SC.W - 1)) ? SC.end_ns : base + (SC.W - 1);
        if (final_launch) wend = SC.end_ns;
        if (blockIdx.x == 0 && tid == 0) SC.wend_slots[win & 1] = wend;
    }
    __syncthreads();

    int64_t nt = kInfNs;
    int merge_overflow = 0;
    if (live && (flags & 8)) {
        // after hs_net_async: whatever is still in this LP's link queues arrives beyond end_ns (it only matters to the
        // election of the one event beyond end_ns); nothing is appended any more, plain bookkeeping
        nt = NX.next_time[lp];
        int bn = NX.bag_cnt[lp];
        for (int q = NP.in_off[lp]; q < NP.in_off[lp + 1]; ++q) {
            const int l = NP.in_links[q];
            unsigned long long head = NX.aq_head[l];
            const unsigned long long tail = pk_tail(ag_load(&NX.aq_ea[l]), head);
            for (; head < tail; ++head) {
                if (bn >= NX.bag_cap) { merge_overflow = 1; break; }
                const size_t slot = (size_t)l * NX.aq_cap + (size_t)(head & (unsigned long long)(NX.aq_cap - 1));
                const size_t dst = (size_t)lp * NX.bag_cap + bn;
                const int64_t t = ag_load(&NX.aq_t[slot]);
                NX.bag_t[dst] = t; NX.bag_ts[dst] = ag_load(&NX.aq_ts[slot]); NX.bag_cr[dst] = ag_load(&NX.aq_cr[slot]);
                NX.bag_link[dst] = l;
                nt = t < nt ? t : nt;
                ++bn;
            }
            NX.aq_head[l] = head;
        }
        NX.bag_cnt[lp] = bn;
        NX.next_time[lp] = nt;
    } else if (live) {
        nt = NX.next_time[lp];
        const size_t cs = (size_t)merge_idx * n + lp;
        const int c = NX.in_cnt[cs];
        if (c > 0) {   // EXCHANGE: take delivery of the messages sent during the previous window
            int bn = NX.bag_cnt[lp];
            const int cc = c < NX.bag_cap ? c : NX.bag_cap;
            if (c > NX.bag_cap) merge_overflow = 1;
            for (int i = 0; i < cc; ++i) {
                if (bn >= NX.bag_cap) { merge_overflow = 1; break; }
                const size_t src = cs * NX.bag_cap + i, dst = (size_t)lp * NX.bag_cap + bn;
                const int64_t t = NX.in_t[src];
                NX.bag_t[dst] = t; NX.bag_ts[dst] = NX.in_ts[src]; NX.bag_cr[dst] = NX.in_cr[src];
                NX.bag_link[dst] = NX.in_link[src];
                nt = t < nt ? t : nt;
                ++bn;
            }
            NX.bag_cnt[lp] = bn;
            NX.in_cnt[cs] = 0;
            NX.next_time[lp] = nt;
        }
    }
    const bool act = live && (nt <= wend || final_launch);
    if (!__syncthreads_or((int)act | merge_overflow)) {         // nothing happens in this workgroup's window
        if (SC.wend_slots != nullptr) {                         // ... but its pending work still bounds the GVT
            if (live) atomicMin(&red_gvt, (long long)nt);
            __syncthreads();
            if (tid == 0 && red_gvt != kInfNs) atomicMin((long long *)SC.gvt_out, red_gvt);
        }
        return;
    }

    NetStation<C> S;
    Candidate mine;
    mine.valid = 0; mine.t = kInfNs; mine.t_created = 0; mine.lp = lp; mine.rank = lp; mine.pad = 0;
    if (act) {
        load_net<C>(S, P, NP, X, NX, L, lp, n, qmem, enqpay, tid, send_idx, SC);
        for (;;) {
            const int64_t t = S.next_time();
            if (t > wend) break;
            S.run_group(t, (flags & 1) != 0);
        }
        nt = S.next_time();
        nt = S.sent_min < nt ? S.sent_min : nt;
        if (final_launch) {
            const int64_t t = S.next_time();
            if (t != kInfNs) {
                const int w = S.pick_root(t);
                mine.t = t; mine.valid = 1;
                if (w == 1) { mine.t_created = S.crtA; mine.pad = 2; }
                else if (w >= 64) mine.t_created = NX.bag_ts[(size_t)lp * NX.bag_cap + (w - 64)];
                else if (w >= 56 && w < 56 + kMaxProbes) {
#pragma unroll
                    for (int j = 0; j < kMaxProbes; ++j) if (j == w - 56) mine.t_created = S.crtP[j];
                    mine.pad = 1;                                 // a Probe's tick ranks behind every Source (as in the station engine)
                }
                else if (w >= 48 && w < 48 + kMaxXSrc) { mine.t_created = X.crtX[(size_t)(w - 48) * n + lp]; mine.pad = 3 + (w - 48); }
                else if (w == 62) mine.t_created = INT64_MIN;     // constructed before run()
                else {
#pragma unroll
                    for (int i = 0; i < C; ++i) if (i == w - 2) mine.t_created = S.crtD[i];
                }
                mine.rank = cand_rank(P, lp, n, mine.pad);        // ties on (time, creation time): `sources=[...]` order
            }
        }
        store_net<C>(S, X, NX, lp, n);
    }

    unsigned vals[13];
#pragma unroll
    for (int k = 0; k < 11; ++k) vals[k] = act ? S.ev[k] : 0u;
    vals[11] = vals[6]; vals[12] = vals[7];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const unsigned s = wave_sum<unsigned>(vals[k]);
        if ((tid & 63) == 0 && s) atomicAdd(&red[k], (unsigned long long)s);
    }
    if (act) {
        atomicMax(&red_time, (long long)S.last_time);
        if (S.overflow) red_flags[0] = 1;
        if (S.qoverflow) red_flags[1] = 1;
        if (S.bagoverflow) red_flags[2] = 1;
    }
    if (merge_overflow) red_flags[2] = 1;
    if (act && S.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)S.evp[0]);    // probe events straight to the totals (rare LPs)
    if (act && S.evp[1]) atomicAdd(&tot->ev[14], (unsigned long long)S.evp[1]);
    if (SC.wend_slots != nullptr && live) atomicMin(&red_gvt, (long long)nt);
    if (final_launch) {
        const Candidate w = wave_min_cand(mine);
        if ((tid & 63) == 0) wave_c[tid >> 6] = w;
    }
    __syncthreads();
    if (SC.wend_slots != nullptr && tid == 14 && red_gvt != kInfNs) atomicMin((long long *)SC.gvt_out, red_gvt);
    if (tid < 11 && red[tid]) atomicAdd(&tot->ev[tid], red[tid]);
    if (tid == 11 && red[11]) atomicAdd(&tot->completed, red[11]);
    if (tid == 12 && red[12]) atomicAdd(&tot->received, red[12]);
    if (tid == 13) {
        if (red_time != INT64_MIN) atomicMax(&tot->final_time, red_time);
        if (red_flags[0]) atomicOr(&tot->overflow, 1);
        if (red_flags[1]) atomicOr(&tot->qoverflow, 1);
        if (red_flags[2]) atomicOr(&tot->overflow, 2);
    }
    if (!final_launch) return;

    if (tid == 0) {
        Candidate b = wave_c[0];
        for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
        cands[blockIdx.x] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = atomicAdd(&tot->done, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    Candidate best;
    best.valid = 0; best.t = kInfNs; best.t_created = 0; best.lp = 0; best.rank = 0;
    for (int b = tid; b < (int)gridDim.x; b += kBlock) {
        Candidate c;
        c.t = __hip_atomic_load(&cands[b].t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.t_created = __hip_atomic_load(&cands[b].t_created, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.lp = __hip_atomic_load(&cands[b].lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.rank = __hip_atomic_load(&cands[b].rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.valid = __hip_atomic_load(&cands[b].valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cand_less(c, best)) best = c;
    }
    best = wave_min_cand(best);
    if ((tid & 63) == 0) wave_c[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        Candidate b = wave_c[0];
        for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
        long long new_cur = __hip_atomic_load(&tot->final_time, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flags & 4) {
            // sharded network: the election continues across the ranks on the host (hs_engine_shard_overshoot runs
            // the winner); publish this rank's candidate
            SC.cand_out[0] = b.valid; SC.cand_out[1] = b.t; SC.cand_out[2] = b.t_created;
            SC.cand_out[3] = SC.lp_base + b.lp;
            tot->cur_time = new_cur;
            tot->done = 0;
            return;
        }
        if (cur0 > wend) new_cur = cur0;              // ... its one event beyond end_time included
        else if (b.valid) {
            // the one event beyond end_time (core/simulation.py:472): first micro-event of the winner's next group
            NetStation<C> W;
            load_net<C>(W, P, NP, X, NX, L, b.lp, n, qmem, enqpay, 0, send_idx, SC);
            const int64_t t = W.next_time();
            const int w = W.pick_root(t);
            if (w == 1) (void)W.do_tick(t);
            else if (w >= 64) (void)W.do_msg(w - 64, t);
            else if (w >= 56 && w < 56 + kMaxProbes) W.root_probe(w - 56, t);   // the SourceEvent of the Probe; its probe_event stays unprocessed
            else if (w >= 48 && w < 48 + kMaxXSrc) W.root_xsrc(w - 48, t);      // the SourceEvent of a further Source; its Request stays unprocessed
            else if (w == 62) W.root_sched(t);                 // the injected Request@Server; its QUEUE_NOTIFY stays unprocessed
            else (void)W.do_cont_core(w - 2, t);
            W.last_time = t;
            store_net<C>(W, X, NX, b.lp, n);
            for (int k = 0; k < 11; ++k) if (W.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)W.ev[k]);
            if (W.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)W.evp[0]);
            if (W.ev[6]) atomicAdd(&tot->completed, (unsigned long long)W.ev[6]);
            new_cur = b.t;
            atomicMax(&tot->final_time, new_cur);
        }
        tot->cur_time = new_cur;
        tot->done = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Asynchronous conservative engine: the whole run of a station network in ONE launch, every LP resident.
//
// The windowed engine above advances all LPs in lock-step windows of W = min link latency (one launch per window:
// 60 002 launches for 60 s of a ring with 1 ms links).  But an LP only has to wait for the LPs that can SEND to it, and
// with counter-based streams a sender knows a lot about its future: its next completion is `min D` when all workers
// are busy, and otherwise no earlier than (its next arrival, or the earliest thing its own senders may still deliver)
// + the duration of the next service to start -- which is service draw number `svc.k`, already determined.  Every LP
// publishes, per outgoing link, a lower bound `aq_ea` on the arrival time of any message it has not appended yet
// (next completion bound + the link's transit floor) and processes its own events strictly below the minimum of its
// incoming links' bounds (Chandy-Misra-Bryant null messages, carried by one 8-byte word per link).  Bounds grow by
// at least the transit floor per hop, so the ring cannot deadlock; results are those of the windowed engine and of the
// reference's single heap (same per-LP code, same message order), independent of timing.
//
// All LPs must be co-resident (they spin on each other): the host launches this kernel cooperatively and falls back to
// the windowed engine when the grid does not fit.  Every spin is bounded (kAsyncMaxIter idle iterations): a wave that gives up raises
// overflow bit 8 and the host reports an error instead of hanging the device.
// ---------------------------------------------------------------------------------------------
constexpr unsigned kAsyncMaxIter = 1u << 23;   // consecutive iterations in which a wavefront processed nothing (~25 s)
constexpr unsigned kAsyncBlockedMax = 1u << 17;   // ... while a lane waits for buffer space: a buffer deadlock (~1 s)
#ifndef HS_GROUP_CAP
#define HS_GROUP_CAP 4
#endif
#ifndef HS_LOOK
#define HS_LOOK 4
#endif
#ifndef HS_TOPUP_NEED
#define HS_TOPUP_NEED 4
#endif
constexpr int kTopUpNeed = HS_TOPUP_NEED;
constexpr int kAsyncGroupCap = HS_GROUP_CAP;   // event groups per LP per iteration of hs_net_async (debug flags bits 8..15 override)

template <int C, bool PF, bool UNI = false>
__global__ void __launch_bounds__(kBlock) hs_net_async(StationParams P, NetParams NP, StationState X, NetState NX,
                                                       RecordLogs L, Totals *tot, int n, int64_t end_ns, int flags,
                                                       ShardCtl SC, int lanes, int max_iters) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ int64_t enqpay[kEnqPay][kBlock];
    __shared__ double ring_a[kNRing][kBlock], ring_s[kNRing][kBlock], ring_j[kNRing][kBlock];   // pre-drawn E values
    __shared__ int64_t lbag_t[kLBag][kBlock], lbag_ts[kLBag][kBlock], lbag_cr[kLBag][kBlock];   // the bags, in LDS
    __shared__ int32_t lbag_link[kLBag][kBlock];
    __shared__ int64_t crc[kNRing][kBlock];                                                     // created_at of the FIFO's tail
    __shared__ unsigned long long red[14];
    __shared__ long long red_time;
    __shared__ int red_flags[4];
    const int tid = threadIdx.x;
    // `lanes` LPs per wavefront (64 by default; 32 / 16 for experiments, debug flag 32).  Measured on the 65 536-station
    // ring: 16 LPs per wavefront (4 wavefronts per SIMD, 128 VGPRs with spills) is 2.4x SLOWER than 64 -- progress is bound
    // by how fast bounds travel from LP to LP, and neighbours inside one wavefront exchange them once per iteration
    const int lane = tid & 63;
    const int lp = (blockIdx.x * (kBlock / 64) + (tid >> 6)) * lanes + lane;
    const bool live = lane < lanes && lp < n;
    if (tid < 14) red[tid] = 0;
    if (tid == 0) { red_time = INT64_MIN; red_flags[0] = red_flags[1] = red_flags[2] = red_flags[3] = 0; }
    __syncthreads();

    NetStation<C, true, PF, UNI> S;
    S.fl = NetFastLds{ring_a, ring_s, ring_j, lbag_t, lbag_ts, lbag_cr, lbag_link, crc};
    bool done = !live;
    int gave_up = 0;
    if (live) {
        load_net<C, true, PF, UNI>(S, P, NP, X, NX, L, lp, n, qmem, enqpay, tid, 0, SC);
        S.end_ns = end_ns;
#ifdef HS_CYC2
        S.cy2[0] = S.cy2[1] = S.cy2[2] = S.cy2[3] = 0;
#endif
        // this LP's outgoing links (router targets in constructor order, or the single link) and what it last published
        int32_t out_l[2] = {-1, -1};
        if (S.egress == EG_LINK) out_l[0] = S.link_of;
        else if (S.egress == EG_ROUTER) {          // at most two links among the router's targets (hs_engine_set_network)
            int no = 0;
            for (int k = 0; k < S.rtk; ++k) { const int32_t t = S.rt_target(k); if (t >= 0 && no < 2) out_l[no++] = t; }
        }
        constexpr int kOut = UNI ? 1 : 2;                     // (UNI: one NetworkLink per station)
        int64_t out_pub[2] = {INT64_MIN, INT64_MIN};
        const int64_t out_lat[2] = {out_l[0] >= 0 ? NP.link_lat_ns[out_l[0]] : 0, out_l[1] >= 0 ? NP.link_lat_ns[out_l[1]] : 0};
        unsigned long long head_seen[2] = {0ull, 0ull};
        const bool force_general = !UNI && (flags & 1) != 0;   // (UNI is not launched with debug flag 1)
        // Groups per LP per iteration.  The loop below is divergent: a lane with a long stretch of ready groups would keep
        // the other 63 idle, and their bounds only move at iteration boundaries -- so every lane takes a few groups, then
        // the wavefront exchanges bounds again and (measured) many more lanes are ready in the next trip.
        const int group_cap = ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) : kAsyncGroupCap;
        // a lane with fewer pre-drawn values than this makes the wavefront refill (debug flags bits 16..19 override): refills are
        // the expensive part (4 values = 2 Philox blocks + logs + divisions per stream), a dry ring costs one general-path group
        const int topup_need = ((flags >> 16) & 0xf) ? ((flags >> 16) & 0xf) : (group_cap < kTopUpNeed ? group_cap : kTopUpNeed);
        unsigned n_groups = 0, n_iter = 0, groups_before = 0, idle_iters = 0;
        bool aborted = false, blocked = false;
        unsigned blocked_iters = 0;
        // In-wavefront chains.  When this LP's only incoming link comes from the LP in the previous lane, its bound need
        // not wait for that neighbour's next publication: a sender's bound is a (min, +) map of its own input bound,
        //     ea_j(H) = min(a_j, H + b_j),   a_j = min(min D, [idle worker] next own event + dur) + transit floor,
        //                                     b_j = [idle worker] dur + transit floor, else infinity,
        // and such maps compose associatively -- a 6-step prefix scan over the lanes gives every lane the bound it would
        // reach after up to 63 publish / poll round trips, from the senders' CURRENT states (valid: a bound computed from a
        // state covers everything that state can still send).  Chains are cut where the previous lane is not the sender.
        int32_t next_l = -1;                                  // my link to the LP in the next lane, if any
        bool out_remote[2] = {false, false};                  // a shard: links that leave it go to an outbox row, no queue
#pragma unroll
        for (int o = 0; o < kOut; ++o) {
            if (out_l[o] < 0) continue;
            out_remote[o] = !UNI && SC.wend_slots != nullptr && SC.link_rank[out_l[o]] != SC.rank;
            if (!out_remote[o] && NP.link_dst[out_l[o]] == (int32_t)SC.lp_base + lp + 1) next_l = out_l[o];
        }
        const int in_deg = NP.in_off[lp + 1] - NP.in_off[lp];
        const int32_t my_in = in_deg == 1 ? NP.in_links[NP.in_off[lp]] : -1;
        const int32_t prev_next = __shfl_up(next_l, 1, 64);
        const bool chain = (flags & 64) == 0 && lane > 0 && my_in >= 0 && prev_next == my_in;
        const int64_t next_lat = next_l >= 0 ? NP.link_lat_ns[next_l] : 0;
        auto sat = [](int64_t a, int64_t b) { return (a == kInfNs || b == kInfNs) ? kInfNs : a + b; };
        int c_kind = -1;                                      // the sender map kept across the iteration boundary (bound_map)
        int64_t c_A = kInfNs, c_B = kInfNs, c_D = INT64_MIN, c_sdl = 0;
#ifdef HS_CYCLES   // tools/cycles.py --ring: cycles in receive / bound scan / group processing / publication
        unsigned long long cyc[4] = {0, 0, 0, 0};
#endif
        for (unsigned iter = 0;; ++iter) {
            n_iter = iter + 1;
            // debug flag 1024: pseudo-random per-wavefront delays -- results must not depend on timing (tests/test_gpu_ring.py)
            if ((flags & 1024) && ((((iter + 1u) * 2654435761u + (blockIdx.x * 4u + (tid >> 6)) * 40503u) >> 7) & 3u) == 0)
                __builtin_amdgcn_s_sleep(127);
            const int64_t w_peek = done ? 0 : S.async_peek();         // the incoming link's word: its latency hides behind the refills
            S.window_fill(!done);                                     // created_at of what entered the window from a deep queue
            S.top_up(!done, topup_need);                              // whole wavefront: refill the pre-drawn values
            int64_t H = kInfNs;
#ifdef HS_CYCLES
            const unsigned long long q0 = __builtin_readcyclecounter();
#endif
            {   // (a chain lane's only incoming link is the previous lane's next_l: async_receive_one)
                const long long q_next = next_l >= 0 ? (long long)S.link_sent_of(next_l) : 0ll;
                const long long q_prev = shfl_up_ll(q_next, 1);
                S.tail_hint = chain ? (unsigned long long)q_prev : 0ull;
            }
            if (!done) H = S.async_receive(w_peek);                         // messages below H are all in the bag now
#ifdef HS_CYCLES
            const unsigned long long q1 = __builtin_readcyclecounter();
#endif
            // The map of this LP as the sender on next_l, from its state before this iteration's processing:
            //     ea(H) = min(mA, max(H + mB, mD))
            // mA: what it can still send whatever its input bound is; H + mB: an admission that only a message from upstream
            // can cause (arrives >= H); mD: ... which, behind a backlog whose departures are already fixed, cannot leave before
            // that backlog has (pre-sending stations, hs_netstation.hpp `early_upto`).  The family is closed under composition:
            //     (A2,B2,D2) o (A1,B1,D1) = (min(A2, max(A1 + B2, D2)), B1 + B2, max(D1 + B2, D2)).
            int64_t mA = kInfNs, mB = kInfNs, mD = INT64_MIN;
            if (!done && next_l >= 0) {
                if (c_kind >= 0) {         // the map evaluated at the end of the previous iteration; only the receive has happened since
                    mB = c_B; mD = c_D;
                    if (c_kind == 0) mA = c_A;
                    else if (c_kind == 1) { const int64_t a = S.next_admission(); mA = sat(a > S.D_pre ? a : S.D_pre, c_sdl); }
                    else mA = sat(S.next_time(), c_sdl);
                } else S.bound_map(next_l, next_lat, mA, mB, mD);
            }
            auto satd = [](int64_t d, int64_t b) { return d == INT64_MIN ? INT64_MIN : (b == kInfNs ? kInfNs : d + b); };
            if (!chain) {                                             // head of a chain: its input bound is known
                if (!done && S.undrained < H) H = S.undrained;
                int64_t v = sat(H, mB);
                v = v > mD ? v : mD;
                mA = v < mA ? v : mA;
                mB = kInfNs; mD = INT64_MIN;
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {                        // prefix composition m_i o m_{i-1} o ... (Kogge-Stone)
                const int64_t pA = __shfl_up(mA, o, 64), pB = __shfl_up(mB, o, 64), pD = __shfl_up(mD, o, 64);
                if (lane >= o) {
                    int64_t v = sat(pA, mB);
                    v = v > mD ? v : mD;
                    mA = v < mA ? v : mA;
                    const int64_t d2 = satd(pD, mB);
                    mD = d2 > mD ? d2 : mD;
                    mB = sat(pB, mB);
                }
            }
            {
                const int64_t hp = __shfl_up(mA, 1, 64);              // the previous lane's bound towards me
                if (chain && hp > H) H = hp;
                if (!done && S.undrained < H) H = S.undrained;        // ... never beyond what is still sitting in a queue
            }
#ifdef HS_CYCLES
            const unsigned long long q2 = __builtin_readcyclecounter();
            unsigned long long q3 = q2;
#endif
            blocked = false;
            {
                // The group loop is UNIFORM: every lane of the wavefront walks through the same trips and a lane without a ready
                // group is predicated off (`act`).  (A divergent loop -- lanes breaking out one by one -- made the compiler keep
                // a dozen exec masks and ~70 loop-carried register copies per trip alive: half of the VALU work of a trip.)
                const int64_t limit = done ? INT64_MIN : ((H - 1) < end_ns ? (H - 1) : end_ns);
                if (!done) blocked = S.undrained != kInfNs;           // the bag is full and a queue still holds messages
                bool stop = done;
                for (int g = 0; g < group_cap; ++g) {
#ifdef HS_MARK
                    asm volatile("; HSMARK loop_top" ::: "memory");
#endif
                    const int64_t t = S.next_time();
                    bool act = !stop && t <= limit;
                    if (act && !((out_remote[0] || S.async_can_send(out_l[0], head_seen[0])) &&
                                 (UNI || out_remote[1] || S.async_can_send(out_l[1], head_seen[1])))) { blocked = true; act = false; }   // a consumer is behind: wait
                    stop = stop || !act;
                    if (!__any(act)) break;
#ifdef HS_RINGSTAT
                    S.stat_gl = 0; S.stat_slow = 0;
#endif
#ifdef HS_MARK
                    asm volatile("; HSMARK before_step1" ::: "memory");
#endif
                    if constexpr (C == 1) S.step1(t, act, force_general);
                    else { if (act) S.run_group(t, force_general); }
#ifdef HS_MARK
                    asm volatile("; HSMARK after_step1" ::: "memory");
#endif
                    n_groups += act ? 1u : 0u;
#ifdef HS_RINGSTAT
                    {   // one trip of the wavefront: how many lanes ran it, did any take the general path / a global read
                        const unsigned long long m = __ballot(act);
                        const bool leader = (unsigned)lane == (unsigned)__builtin_ctzll(__ballot(1));
                        const int any_gl = __any(S.stat_gl), any_slow = __any(S.stat_slow);
                        if (leader) {
                            atomicAdd(&tot->dbg[0], 1ull); atomicAdd(&tot->dbg[1], (unsigned long long)any_gl);
                            atomicAdd(&tot->dbg[2], (unsigned long long)any_slow); atomicAdd(&tot->dbg[3], (unsigned long long)__builtin_popcountll(m));
                        }
                    }
#endif
                }
            }
            if (!done) {
#ifdef HS_CYCLES
                q3 = __builtin_readcyclecounter();
#endif
                const int64_t t2 = S.next_time();
                const int64_t base = t2 < H ? t2 : H;                 // nothing happens here before `base`
                // payloads complete (one drain), THEN the link's word (bound, tail): a consumer that sees the word sees every
                // message below its tail, and no bound it can read -- from memory or through the in-wavefront scan of the next
                // iteration (after this drain) -- covers less than the messages that are visible behind it
                // (the bounds are computed first -- registers and LDS only -- so that the drain overlaps with that arithmetic)
                int64_t vo[2] = {INT64_MIN, INT64_MIN};
#pragma unroll
                for (int o = 0; o < kOut; ++o) {
                    const int32_t l = out_l[o];
                    if (l < 0) continue;
                    // the bound of everything this LP has NOT appended to link l yet: its map evaluated at `base`
                    int64_t bA, bB, bD, sdl = 0;
                    int kind = -1;
                    S.bound_map(l, out_lat[o], bA, bB, bD, &kind, &sdl);
                    if (l == next_l) { c_kind = kind; c_A = bA; c_B = bB; c_D = bD; c_sdl = sdl; }
                    int64_t v = sat(base, bB);
                    v = v > bD ? v : bD;
                    v = v < bA ? v : bA;
                    vo[o] = v > out_pub[o] ? v : out_pub[o];
                }
                if (S.sent_async) drain_stores();
#pragma unroll
                for (int o = 0; o < kOut; ++o) {
                    const int32_t l = out_l[o];
                    if (l < 0) continue;
                    if (S.sent_async || vo[o] > out_pub[o]) {
                        ag_store(&NX.aq_ea[l], pk_pack(vo[o], (unsigned long long)S.link_sent_of(l), NX.pk_base));
                        out_pub[o] = vo[o];
                    }
                }
                S.sent_async = false;
                done = base > end_ns;                                 // nothing at or before end_ns can happen any more
            }
#ifdef HS_CYCLES
            {
                const unsigned long long q4 = __builtin_readcyclecounter();
                cyc[0] += q1 - q0; cyc[1] += q2 - q1; cyc[2] += q3 - q2;
                cyc[3] += q4 - q3;
            }
#endif
            if (__all(done)) break;
            if (!UNI && max_iters > 0 && (int)(iter + 1) >= max_iters) break; // a shard's exchange round is over
            // a fatal condition anywhere ends the launch everywhere: nobody spins on a dead neighbour
            if (__any(aborted) || ((iter & 31u) == 31u && __hip_atomic_load(&tot->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) break;
            // the spin bound counts iterations in which the whole wavefront processed nothing (with at most kAsyncGroupCap
            // groups per LP per iteration, the number of WORKING iterations grows with the run and is not a sign of a hang)
            const bool wave_idle = !__any(n_groups != groups_before);
            idle_iters = wave_idle ? idle_iters + 1 : 0;
            if (idle_iters >= kAsyncMaxIter) { gave_up = 1; atomicOr(&tot->overflow, 8); break; }
            // Bounded buffers + time-ordered processing can deadlock on a cycle: every LP waits for room in its outgoing
            // queue while its own bag is full of messages it may not process yet.  A wavefront that makes no progress for
            // a long time while one of its lanes waits for buffer space reports that (raise bag_capacity) instead of spinning.
            if (wave_idle && __any(!done && blocked)) {
                if (++blocked_iters >= kAsyncBlockedMax) { atomicOr(&tot->overflow, 2); aborted = true; }
            } else blocked_iters = 0;
            if ((flags & 128) && wave_idle) __builtin_amdgcn_s_sleep(64);   // experiment: back off when idle
            groups_before = n_groups;
        }
        store_net<C, true, PF, UNI>(S, X, NX, lp, n);
        if (!UNI && max_iters > 0 && !done) atomicAdd(&tot->not_done, 1ull);
        if constexpr (PF) {      // probe events straight to the totals (rare LPs)
            if (S.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)S.evp[0]);
            if (S.evp[1]) atomicAdd(&tot->ev[14], (unsigned long long)S.evp[1]);
        }
#ifdef HS_CYC2
        if ((tid & 63) == 0) for (int k = 0; k < 4; ++k) atomicAdd(&tot->dbg[k], S.cy2[k]);
#elif defined(HS_CYCLES)
        if ((tid & 63) == 0) for (int k = 0; k < 4; ++k) atomicAdd(&tot->dbg[k], cyc[k]);
#elif defined(HS_RINGSTAT)
        (void)n_iter;
#else
        atomicAdd(&tot->dbg[2], (unsigned long long)n_groups);
        if ((tid & 63) == 0) {
            atomicAdd(&tot->dbg[0], (unsigned long long)n_iter); atomicMax(&tot->dbg[1], (unsigned long long)n_iter);
            atomicAdd(&tot->dbg[3], 1ull);
        }
#endif
    }

    // ---- workgroup reduction of the run's deltas -> engine totals (as in hs_net_window)
    unsigned vals[13];
#pragma unroll
    for (int k = 0; k < 11; ++k) vals[k] = live ? S.ev[k] : 0u;
    vals[11] = vals[6]; vals[12] = vals[7];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const unsigned sm = wave_sum<unsigned>(vals[k]);
        if ((tid & 63) == 0 && sm) atomicAdd(&red[k], (unsigned long long)sm);
    }
    if (live) {
        atomicMax(&red_time, (long long)S.last_time);
        if (S.overflow) red_flags[0] = 1;
        if (S.qoverflow) red_flags[1] = 1;
        if (S.bagoverflow) red_flags[2] = 1;
        if (gave_up) red_flags[3] = 1;
    }
    __syncthreads();
    if (tid < 11 && red[tid]) atomicAdd(&tot->ev[tid], red[tid]);
    if (tid == 11 && red[11]) atomicAdd(&tot->completed, red[11]);
    if (tid == 12 && red[12]) atomicAdd(&tot->received, red[12]);
    if (tid == 13) {
        if (red_time != INT64_MIN) atomicMax(&tot->final_time, red_time);
        if (red_flags[0]) atomicOr(&tot->overflow, 1);
        if (red_flags[1]) atomicOr(&tot->qoverflow, 1);
        if (red_flags[2]) atomicOr(&tot->overflow, 2);
        if (red_flags[3]) atomicOr(&tot->overflow, 8);
    }
}

// Sharded network, after the host exchanged the outbox rows: append the messages other ranks sent to this rank's
// stations to the incoming bags of parity `send_idx` (the window that just ran), so that the next launch merges
// them together with the locally sent ones.  Also clears this rank's outbox counters and re-arms the GVT slot
// the NEXT window will accumulate into.
__global__ void hs_shard_inject(NetState NX, const int64_t *inbox, int64_t *outbox, int world, int msg_cap, int row,
                                int n, int64_t lp_base, int send_idx, const int32_t *gid2local, int64_t n_gid,
                                int64_t *gvt_next, Totals *tot) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) *gvt_next = kInfNs;
    if (idx < world) outbox[(size_t)idx * row] = 0;
    if (idx >= (int64_t)world * msg_cap) return;
    const int r = (int)(idx / msg_cap), i = (int)(idx % msg_cap);
    const int64_t *rowp = inbox + (size_t)r * row;
    const int64_t cnt = rowp[0];
    if (cnt > msg_cap && i == 0) atomicOr(&tot->overflow, 2);
    if (i >= cnt) return;
    const int64_t *m = rowp + 1 + 4 * (size_t)i;
    const int64_t dst = (m[3] >> 32) - lp_base, gid = m[3] & 0xffffffffll;
    if (dst < 0 || dst >= n || gid >= n_gid || gid2local[gid] < 0) { atomicOr(&tot->overflow, 4); return; }
    const size_t cslot = (size_t)send_idx * n + (size_t)dst;
    const int pos = atomicAdd(&NX.in_cnt[cslot], 1);
    if (pos < NX.bag_cap) {
        const size_t b = cslot * NX.bag_cap + pos;
        NX.in_t[b] = m[0]; NX.in_ts[b] = m[1]; NX.in_cr[b] = m[2]; NX.in_link[b] = gid2local[gid];
    } else atomicOr(&tot->overflow, 2);
}

// ---- asynchronous shard rounds ------------------------------------------------------------------------------------
// Each shard runs hs_net_async for a bounded number of iterations (a ROUND); between rounds the host moves the outbox
// rows (all-to-all) and the lower bounds of the cross-shard links (all-reduce MAX).  A cross link behaves like any other
// link of the asynchronous engine -- messages in its queue, a bound in aq_ea -- only that queue and bound are refilled
// between launches instead of by a concurrently running producer.  Exchange rounds therefore follow the boundary LPs'
// lookahead (tens of ms of simulated time), not the 1 ms link floor of the windowed protocol.
// What this rank publishes after a round: the bounds of the cross links that START here, and whether it still has work.
__global__ void hs_shard_bounds_out(const int64_t *aq_ea, int64_t pk_base, const int32_t *cross_local, const uint8_t *cross_role, int n_cross,
                                    int64_t *bounds, const Totals *tot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cross) bounds[i] = (cross_role[i] & 1) ? pk_ea(aq_ea[cross_local[i]], pk_base) : INT64_MIN;
    if (i == n_cross) bounds[n_cross] = tot->not_done ? 1 : 0;
}
// After the exchange: the received messages go to the queues of their links (row r holds what rank r sent here, in send
// order; a link has one producer, so one thread per row keeps every queue in order), the all-reduced bounds become the
// aq_ea of the cross links that END here.  One workgroup; the engine's kernel is not running.
__global__ void hs_shard_inject_async(NetState NX, const int64_t *inbox, int64_t *outbox, int world, int msg_cap, int row,
                                      int n, int64_t lp_base, const int32_t *gid2local, int64_t n_gid,
                                      const int32_t *cross_local, const uint8_t *cross_role, int n_cross,
                                      const int64_t *bounds, Totals *tot) {
    const int tid = threadIdx.x;
    for (int r = tid; r < world; r += blockDim.x) {
        const int64_t *rowp = inbox + (size_t)r * row;
        int64_t cnt = rowp[0];
        if (cnt > msg_cap) { atomicOr(&tot->overflow, 2); cnt = msg_cap; }
        for (int64_t i = 0; i < cnt; ++i) {
            const int64_t *m = rowp + 1 + 4 * (size_t)i;
            const int64_t dst = (m[3] >> 32) - lp_base, gid = m[3] & 0xffffffffll;
            if (dst < 0 || dst >= n || gid >= n_gid || gid2local[gid] < 0) { atomicOr(&tot->overflow, 4); continue; }
            const int l = gid2local[gid];
            const unsigned long long head = NX.aq_head[l], tail = pk_tail(NX.aq_ea[l], head);
            if (tail - head >= (unsigned long long)NX.aq_cap) { atomicOr(&tot->overflow, 2); continue; }
            const size_t slot = (size_t)l * NX.aq_cap + (size_t)(tail & (unsigned long long)(NX.aq_cap - 1));
            NX.aq_t[slot] = m[0]; NX.aq_ts[slot] = m[1]; NX.aq_cr[slot] = m[2];
            NX.aq_ea[l] = pk_pack(pk_ea(NX.aq_ea[l], NX.pk_base), tail + 1, NX.pk_base);
        }
        outbox[(size_t)r * row] = 0;
    }
    __syncthreads();          // the tails first (one thread per row), then the bounds (one thread per link): the same words
    for (int i = tid; i < n_cross; i += blockDim.x)
        if (cross_role[i] & 2) {
            const int l = cross_local[i];
            NX.aq_ea[l] = pk_pack(bounds[i], pk_tail(NX.aq_ea[l], NX.aq_head[l]), NX.pk_base);
        }
    if (tid == 0) tot->not_done = 0;
}

// Sharded network: this rank owns the globally first event beyond end_ns -- process it (core/simulation.py:472).
template <int C>
__global__ void hs_shard_overshoot(StationParams P, NetParams NP, StationState X, NetState NX, RecordLogs L,
                                   Totals *tot, int n, int lp, int win, ShardCtl SC) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ int64_t enqpay[kEnqPay][kBlock];
    if (threadIdx.x != 0) return;
    NetStation<C> W;
    load_net<C>(W, P, NP, X, NX, L, lp, n, qmem, enqpay, 0, win & 1, SC);
    const int64_t t = W.next_time();
    if (t == kInfNs) return;
    const int w = W.pick_root(t);
    if (w == 1) (void)W.do_tick(t);
    else if (w >= 64) (void)W.do_msg(w - 64, t);
    else if (w >= 56 && w < 56 + kMaxProbes) W.root_probe(w - 56, t);
    else if (w >= 48 && w < 48 + kMaxXSrc) W.root_xsrc(w - 48, t);
    else if (w == 62) W.root_sched(t);
    else (void)W.do_cont_core(w - 2, t);
    W.last_time = t;
    store_net<C>(W, X, NX, lp, n);
    for (int k = 0; k < 11; ++k) if (W.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)W.ev[k]);
    if (W.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)W.evp[0]);
    if (W.ev[6]) atomicAdd(&tot->completed, (unsigned long long)W.ev[6]);
    atomicMax(&tot->final_time, (long long)t);
    tot->cur_time = t;
}

// Read-back of the record logs.  The logs are [cap][n_lp] (record k of LP lp at k * n + lp) so that the run kernels'
// appends coalesce; the C ABI hands records out per LP, concatenated in LP order.  64 x 64 tiles through LDS: coalesced
// reads along lp, coalesced writes along k.
__global__ void __launch_bounds__(256) hs_gather_logs(const int64_t *__restrict__ log, const int64_t *__restrict__ cnt,
                                                      const int64_t *__restrict__ off, int64_t *__restrict__ out, int n,
                                                      int64_t cap) {
    __shared__ int64_t tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int lp0 = blockIdx.x * 64;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    for (int kk = ty; kk < 64; kk += 4) {
        const int64_t k = k0 + kk;
        const int lp = lp0 + tx;
        tile[kk][tx] = (k < cap && lp < n) ? log[(size_t)k * n + lp] : 0;
    }
    __syncthreads();
    for (int l = ty; l < 64; l += 4) {
        const int lp = lp0 + l;
        if (lp >= n) continue;
        int64_t c = cnt[lp];
        c = c > cap ? cap : c;
        const int64_t k = k0 + tx;
        if (k < c) out[off[lp] + k] = tile[tx][l];
    }
}
// one LP's records (hs_engine_read_sink)
__global__ void hs_gather_one(const int64_t *__restrict__ log, int64_t *__restrict__ out, int n, int lp, int64_t cnt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) out[k] = log[(size_t)k * n + lp];
}

__global__ void hs_debug_draws_kernel(uint64_t seed, uint64_t sid, uint64_t k0, int64_t n, double rate, double *u,
                                      double *e, int64_t *ns) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Stream s;
    s.init(seed, sid, k0 + (uint64_t)i);
    const double uu = s.next_uniform();
    const double ee = exp1_from_uniform(uu);
    u[i] = uu; e[i] = ee;
    ns[i] = ns_from_seconds(__ddiv_rn(ee, rate));
}

// test hook: constant-divisor quotients (ConstDiv, seconds_from_ns) next to the IEEE division
__global__ void hs_debug_const_div_kernel(double b, int64_t n, const double *a, double *q_fast, double *q_ieee,
                                          double *q_ns) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ConstDiv d;
    d.init(b);
    q_fast[i] = d.div(a[i]);
    q_ieee[i] = __ddiv_rn(a[i], b);
    q_ns[i] = seconds_from_ns((int64_t)a[i]);
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_global_error;

struct hs_engine {
    hs_config cfg{};
    int C = 1;                 // departure slots compiled for (>= max concurrency)
    bool have_stations = false;
    bool initialised = false;  // reset done
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::vector<void *> allocs;
    StationParams P{};
    StationState X{};
    RecordLogs L{};
    Totals *tot = nullptr;
    Candidate *cands = nullptr;
    bool is_net = false;
    bool any_xsrc = false;     // some LP has more than one Source (general path + prologue)
    bool uni_stations = false; // every LP: Poisson Source, exponential single-worker Server, unbounded queue, no stop_after
    bool uni_grid = false;     // ... and a Sink behind every Server: hs_station_run<1, false, true, true>
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
    bool async_ok = false;     // the network can run on hs_net_async (whole network on this engine, queues allocated)
    int round_iters = 0;       // > 0: hs_net_async runs one exchange round of a shard (that many iterations), not a whole run
    // asynchronous shard rounds (hs_engine_shard_async_*): the network's cross-shard links
    int n_cross = 0;
    const int32_t *cross_local = nullptr;   // [n_cross] local link index (-1: this shard does not touch the link)
    const uint8_t *cross_role = nullptr;    // [n_cross] bit 0: the source station is here, bit 1: the destination is
    int64_t *cross_bounds = nullptr;        // device int64[n_cross + 1], owned by the caller (all-reduced with MAX)
    bool shard_async = false;
    bool net_pf = false;       // the network has probes / profiles / scheduled Requests: the PF instantiation of hs_net_async
    int round_iters_cfg = 0;
    int async_fit = -1;        // -1 unknown, 0 the grid is not co-resident (windowed engine), 1 it is
    int async_lanes = 64;      // LPs per wavefront in hs_net_async
    int n_blocks = 0;
    int flags = 0;
    // prologue (hs_exact.hpp): SINGLE mode with probes / scheduled Requests
    bool exact = false;
    XState *xs = nullptr;
    XState xs_host{};          // the device pointers / capacities of *xs (phase etc. are reset from it)
    XInit XI{};
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
    return HS_OK;
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
void launch_run(hs_engine *h, int64_t end_ns) {
    if constexpr (C == 1) {
        if (!h->any_profile && !(h->flags & 512)) {     // producer / consumer wavefronts (debug flag 512: the one-role kernel)
            if (h->uni_grid && (h->flags & (1 << 20)) == 0)   // uniform entity kinds: compile-time predicates (hs_station.hpp HSG)
                hipLaunchKernelGGL((hs_station_run<1, false, true, true>), dim3(h->n_blocks), dim3(2 * kBlock), 0, h->stream, h->P,
                                   h->X, h->L, h->tot, h->cands, h->cfg.n_lp, end_ns, h->cfg.mode, h->flags);
            else
                hipLaunchKernelGGL((hs_station_run<1, false, true>), dim3(h->n_blocks), dim3(2 * kBlock), 0, h->stream, h->P, h->X,
                                   h->L, h->tot, h->cands, h->cfg.n_lp, end_ns, h->cfg.mode, h->flags);
            return;
        }
    }
    if (h->any_profile)
        hipLaunchKernelGGL((hs_station_run<C, true>), dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot,
                           h->cands, h->cfg.n_lp, end_ns, h->cfg.mode, h->flags);
    else
        hipLaunchKernelGGL((hs_station_run<C, false>), dim3(h->n_blocks), dim3(kBlock), 0, h->stream, h->P, h->X, h->L, h->tot,
                           h->cands, h->cfg.n_lp, end_ns, h->cfg.mode, h->flags);
}

void launch_run_dispatch(hs_engine *h, int64_t end_ns) {
    switch (h->C) {
        case 1: launch_run<1>(h, end_ns); break;
        case 2: launch_run<2>(h, end_ns); break;
        case 4: launch_run<4>(h, end_ns); break;
        case 8: launch_run<8>(h, end_ns); break;
        case 16: launch_run<16>(h, end_ns); break;
        default: launch_run<32>(h, end_ns); break;     // (departure slots beyond 16 live in scratch: correct, not fast)
    }
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

template <int C>
hipError_t launch_async(hs_engine *h, int64_t end_ns, NetState NX) {
    int n = h->cfg.n_lp, flags = h->flags & (1 | 64 | 128 | 1024 | 0xff00), lanes = h->async_lanes;
    const int per_block = (kBlock / 64) * lanes;
    int max_iters = h->round_iters;
    void *args[] = {&h->P, &h->NP, &h->X, &NX, &h->L, &h->tot, &n, &end_ns, &flags, &h->SC, &lanes, &max_iters};
    const void *fn = h->net_pf ? (const void *)hs_net_async<C, true> : (const void *)hs_net_async<C, false>;
    if constexpr (C == 1) {    // uniform entity kinds: the specialised instantiation (debug flag 1 << 20 keeps the generic one)
        if (h->net_uni && !h->net_pf && (h->flags & ((1 << 20) | 1 | 32)) == 0 && h->round_iters == 0 && lanes == 64)
            fn = (const void *)hs_net_async<1, false, true>;
    }
    return hipLaunchCooperativeKernel(fn, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), args, 0, h->stream);
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
        h->async_fit = 0;
        for (int lanes = (h->flags & 32) ? 16 : 64; lanes <= 64; lanes *= 2) {
            const int per_block = (kBlock / 64) * lanes;
            if (((long long)h->cfg.n_lp + per_block - 1) / per_block <= resident) { h->async_lanes = lanes; h->async_fit = 1; break; }
        }
    }
    return h->async_fit;
}
int try_run_net_whole(hs_engine *h, int64_t end_ns) {
    if (!h->async_ok || (h->flags & 16)) return 0;
    if (!ensure_async_fit(h)) return 0;
    NetState NX = h->NX;
    NX.aq_on = 1;
    hipError_t e = h->C == 1 ? launch_async<1>(h, end_ns, NX) : h->C == 2 ? launch_async<2>(h, end_ns, NX) : launch_async<4>(h, end_ns, NX);
    if (e != hipSuccess) { (void)hipGetLastError(); h->async_fit = 0; return 0; }   // not co-resident after all: windows
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
    {
        const int whole = try_run_net_whole(h, end_ns);
        if (whole != 0) return whole < 0 ? whole : HS_OK;
    }
    const int64_t W = h->window_ns;
    int64_t t0 = h->cfg.start_ns;
    int win = 0;
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

int do_reset_async(hs_engine *h) {
    if (h->any_profile) {      // the arrival-time inversion's budget flag (hs_profile.hpp): cleared BEFORE the bootstrap draws
        static const unsigned long long zero = 0ull;
        HS_HIP(h, hipMemcpyToSymbolAsync(HIP_SYMBOL(hs_prof_budget_hit), &zero, sizeof zero, 0, hipMemcpyHostToDevice, h->stream));
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
    h->initialised = true;
    h->net_ran = false;
    return HS_OK;
}

// the prologue of a run (hs_exact.hpp); a no-op launch once it has handed over
int launch_prologue(hs_engine *h, int64_t end_ns) {
    if (!h->exact || (h->flags & 256)) return HS_OK;
    hipLaunchKernelGGL(hs_exact_run, dim3(h->XI.per_lp ? (unsigned)((h->cfg.n_lp + 63) / 64) : 1u), dim3(64), 0, h->stream, h->P, h->NP, h->X, h->NX, h->L, h->tot, h->xs, h->XI,
                       h->cfg.n_lp, h->C, h->is_net ? 1 : 0, h->is_net ? h->NP.n_links : 0, h->cfg.start_ns, end_ns);
    HS_HIP(h, hipGetLastError());
    h->launches++;
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
        if (eg != HS_EGRESS_NONE && eg != HS_EGRESS_SINK)
            return fail(h, HS_E_UNSUPPORTED, "LP %d: egress kind %d is not lowered", i, eg);
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
    if (st->source_order) {   // cross-LP ties go to the Source the reference constructed first (cand_rank, hs_station.hpp)
        std::vector<int32_t> tr((size_t)n * (kMaxXSrc + 2) + 1, -1);
        int32_t *sr = tr.data() + n;
        for (size_t q = 0; q < so.size(); ++q) {
            sr[(size_t)sslot[q] * n + (size_t)so[q]] = (int32_t)q;               // a tick: its own Source's position
            if (tr[(size_t)so[q]] < 0) tr[(size_t)so[q]] = (int32_t)q;           // anything else: the LP's first-listed Source
        }
        for (int i = 0; i < n; ++i) if (tr[(size_t)i] < 0) tr[(size_t)i] = (int32_t)so.size() + i;   // sourceless LPs after them
        tr[(size_t)n * (kMaxXSrc + 2)] = (int32_t)so.size() + n;                 // Probes behind all of them
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
    }
    if (h->cfg.mode == HS_MODE_SINGLE && (n_sched > 0 || h->any_probe || h->any_xsrc)) {
        // The prologue (hs_exact.hpp): the reference's pre-run events in the order it constructs them
        std::vector<int32_t> po, sl((size_t)n_sched);
        std::vector<uint8_t> pslot;
        std::vector<int64_t> se((size_t)n_sched);
        {   // probes in `probes=[...]` order: (LP, slot) pairs; default = LP-major, slot-minor
            std::vector<uint8_t> taken((size_t)n * kMaxProbes, (uint8_t)0);
            for (int64_t k = 0; k < n_prb_total; ++k) {
                int lp = -1, slot = 0;
                if (st->probe_order) { lp = st->probe_order[k]; slot = st->probe_slot_order ? st->probe_slot_order[k] : 0; }
                else {
                    int64_t seen_k = 0;
                    for (int i = 0; i < n && lp < 0; ++i)
                        for (int j = 0; j < kMaxProbes; ++j)
                            if (pm[(size_t)j * n + i] != 255) { if (seen_k == k) { lp = i; slot = j; break; } ++seen_k; }
                }
                if (lp < 0 || lp >= n || slot < 0 || slot >= kMaxProbes || pm[(size_t)slot * n + lp] == 255 || taken[(size_t)slot * n + lp])
                    return fail(h, HS_E_INVALID, "probe_order / probe_slot_order must list every probe exactly once");
                taken[(size_t)slot * n + lp] = 1;
                po.push_back(lp); pslot.push_back((uint8_t)slot);
            }
        }
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
        h->xs_host = XState{};
        h->xs_host.heap_cap = n_init + (int64_t)n * (h->C + 16) + 1024;
        h->xs_host.pool_cap = 2 * n_init + 16 * (int64_t)n + 1024;
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
    if (h->any_profile) {      // the general-path instantiation of the run kernel loads / stores the probe state of every LP
        AL(PA, N * kMaxProbes); AL(seqP, N * kMaxProbes); AL(crtP, N * kMaxProbes); AL(p_arr, N * kMaxProbes);
        AL(p_n, N * kMaxProbes); AL(ev_probe, N * 2); AL(sched_i, N);
    }
    if (h->any_xsrc) {
        AL(XA, N * kMaxXSrc); AL(crtX, N * kMaxXSrc); AL(x_arr, N * kMaxXSrc); AL(x_n, N * kMaxXSrc); AL(seqX, N * kMaxXSrc);
        AL(x_k, N * kMaxXSrc);
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
    if ((rc = dev_alloc(h, &h->cands, (size_t)h->n_blocks))) return rc;
    HS_HIP(h, hipMemset(h->tot, 0, sizeof(Totals)));
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
    ALN(bag_cnt, N); ALN(bag_t, NB); ALN(bag_ts, NB); ALN(bag_cr, NB); ALN(bag_link, NB);
    ALN(in_cnt, 2 * N); ALN(in_t, 2 * NB); ALN(in_ts, 2 * NB); ALN(in_cr, 2 * NB); ALN(in_link, 2 * NB);
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
        ALN(aq_t, NQ); ALN(aq_ts, NQ); ALN(aq_cr, NQ); ALN(aq_tail, NL); ALN(aq_head, NL); ALN(aq_ea, NL);
        ALN(early_upto, (size_t)n); ALN(d_pre, (size_t)n);
        // the whole network in one cooperative launch (shards: hs_engine_shard_round); with probes, time-varying profiles
        // or scheduled Requests the PF instantiation of the kernel (a profile's next arrival and the next scheduled Request
        // are part of next_admission() / next_time(), which is all the bounds are made of)
        // (the links' packed (bound, tail) words hold 44 bits of nanoseconds: 4.9 hours of simulated time)
        const bool fits = h->cfg.horizon_ns - h->cfg.start_ns < (int64_t)kPkNever - 2 && aqc < (1 << (kPkTailBits - 2));
        h->async_ok = !global && fits;
        h->net_pf = h->any_probe || h->any_timevarying || h->any_sched || h->any_xsrc;
        // the network's entity kinds are uniform: the specialised instantiation (hs_netstation.hpp HSU)
        h->net_uni = h->uni_stations && !global && !h->net_pf && h->C == 1;
        for (int i = 0; i < n && h->net_uni; ++i) {
            if (net->egress_kind[i] != HS_EGRESS_ROUTER) { h->net_uni = false; break; }
            int n_link = 0, l1 = -1;
            const int32_t tg[4] = {rt0[(size_t)i], rt1[(size_t)i], rt2[(size_t)i], rt3[(size_t)i]};
            for (int q = 0; q < (int)rtk[(size_t)i]; ++q) if (tg[q] >= 0) { ++n_link; l1 = tg[q]; }
            if (n_link != 1 || jk[(size_t)l1] != HS_LAT_EXPONENTIAL || lloss[(size_t)l1] != 0.0) h->net_uni = false;
            if (h->net_uni && in_deg_h[(size_t)i] != 1) h->net_uni = false;        // exactly one incoming link per station
        }
    }
#undef ALN
    if (!h->L.sink_created_own) {   // not every completion reaches the Sink any more: explicit created_at column
        if ((rc = dev_alloc(h, &h->L.sink_created_own, N * (size_t)h->L.cap))) return rc;
    }
    h->L.sink_created = h->L.sink_created_own;
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
    h->SC.msg_cap = sh->msg_capacity; h->SC.row = 1 + 4 * sh->msg_capacity;
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
    h->SC.end_ns = end_ns;
    const int64_t init_slots[2] = {h->cfg.start_ns - 1, h->cfg.start_ns - 1};
    const int64_t init_gvt[2] = {kInfNs, h->cfg.start_ns};       // window 0 reads gvt[1], accumulates into gvt[0]
    HS_HIP(h, hipMemcpyAsync(h->SC.wend_slots, init_slots, sizeof init_slots, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemcpyAsync(h->shard_gvt, init_gvt, sizeof init_gvt, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemsetAsync(h->SC.outbox, 0, (size_t)h->SC.world * h->SC.row * 8, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));                   // the host arrays above are stack memory
    h->launches = 1;
    h->net_ran = true;
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
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a message bag, a link queue or an exchange row overflowed; raise bag_capacity / msg_capacity");
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_shard_final(hs_engine *h, int64_t k) {
    if (!h || !h->SC.wend_slots) return fail(h, HS_E_STATE, "hs_engine_shard_final: no shard attached");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    h->SC.gvt_in = h->shard_gvt + ((k + 1) & 1);
    h->SC.gvt_out = h->shard_gvt + (k & 1);
    launch_net_dispatch(h, h->SC.end_ns, (int)(k & 0x3fffffff), (h->flags & 1) | 2 | 4 | (h->shard_async ? 8 : 0));
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

int hs_engine_get_net_stats(hs_engine *h, const hs_net_stats *o) {
    if (!h || !o) return fail(h, HS_E_INVALID, "hs_engine_get_net_stats: null argument");
    if (!h->is_net) return fail(h, HS_E_STATE, "no network set");
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

int hs_engine_reset(hs_engine *h) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "hs_engine_reset: stations not set");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    int rc = do_reset_async(h);
    if (rc) return rc;
    HS_HIP(h, hipStreamSynchronize(h->stream));
    return HS_OK;
}

int hs_engine_run_until_async(hs_engine *h, int64_t end_ns) {
    if (!h || !h->have_stations) return fail(h, HS_E_STATE, "hs_engine_run_until: stations not set");
    if (end_ns > h->cfg.horizon_ns)
        return fail(h, HS_E_INVALID, "end_ns %lld beyond the configured horizon %lld", (long long)end_ns,
                    (long long)h->cfg.horizon_ns);
    HS_HIP(h, hipSetDevice(h->cfg.device));
    h->launches = 0;
    HS_HIP(h, hipEventRecord(h->ev_a, h->stream));
    if (!h->initialised) { int rc = do_reset_async(h); if (rc) return rc; h->launches++; }
    HS_HIP(h, hipEventRecord(h->ev_k0, h->stream));
    if (h->is_net) {
        if (h->net_global) return fail(h, HS_E_STATE, "a shard of a partitioned network is driven with hs_engine_shard_*");
        if (h->net_ran) return fail(h, HS_E_STATE, "network engine: one hs_engine_run_until per hs_engine_reset");
        int rc = launch_prologue(h, end_ns);
        if (rc) return rc;
        rc = run_net_async(h, end_ns);
        if (rc) return rc;
    } else {
        int rc = launch_prologue(h, end_ns);
        if (rc) return rc;
        launch_run_dispatch(h, end_ns);
        HS_HIP(h, hipGetLastError());
        h->launches++;
    }
    HS_HIP(h, hipEventRecord(h->ev_k1, h->stream));
    HS_HIP(h, hipEventRecord(h->ev_b, h->stream));
    return HS_OK;
}

int hs_engine_synchronize(hs_engine *h) {
    if (!h) return fail(h, HS_E_INVALID, "null handle");
    HS_HIP(h, hipSetDevice(h->cfg.device));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) h->last_run_ms = ms;
    if (hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1) == hipSuccess) h->last_kernel_ms = ms;
    return HS_OK;
}

int hs_engine_run_until(hs_engine *h, int64_t end_ns) {
    int rc = hs_engine_run_until_async(h, end_ns);
    if (rc) return rc;
    rc = hs_engine_synchronize(h);
    if (rc) return rc;
    Totals t;
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (h->any_profile) {
        unsigned long long hit = 0ull;
        HS_HIP(h, hipMemcpyFromSymbol(&hit, HIP_SYMBOL(hs_prof_budget_hit), sizeof hit, 0, hipMemcpyDeviceToHost));
        if (hit != 0ull)
            return fail(h, HS_E_UNSUPPORTED, "LP %lld: one arrival of its time-varying Source needs more than %lld adaptive-Simpson "
                        "intervals (the reference's own integrator needs minutes for such an arrival: a ramp that starts near zero "
                        "rate; check with tools/profile_cost.py) -- refused instead of stalling a lane",
                        (long long)hit - 2, (long long)kProfBudget);
    }
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 16)
        return fail(h, HS_E_OVERFLOW, "the prologue (csrc/hs_exact.hpp) ran out of heap / payload-pool space");
    if (t.overflow & 8)
        return fail(h, HS_E_HIP, "the asynchronous network engine gave up waiting for a neighbour (bounded spin); "
                                 "set debug flag 16 to use the windowed engine");
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
    HS_HIP(h, hipEventRecord(ev[(size_t)repeats * 2], h->stream));
    for (int r = 0; r < repeats; ++r) {
        int rc = do_reset_async(h);
        if (rc) return rc;
        HS_HIP(h, hipEventRecord(ev[(size_t)2 * r], h->stream));
        { int rc1 = launch_prologue(h, end_ns); if (rc1) return rc1; }
        if (h->is_net) { h->launches = 0; int rc2 = run_net_async(h, end_ns); if (rc2) return rc2; }
        else launch_run_dispatch(h, end_ns);
        HS_HIP(h, hipGetLastError());
        HS_HIP(h, hipEventRecord(ev[(size_t)2 * r + 1], h->stream));
    }
    HS_HIP(h, hipEventRecord(ev[(size_t)repeats * 2 + 1], h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < repeats; ++r) {
        float ms = 0.f;
        HS_HIP(h, hipEventElapsedTime(&ms, ev[(size_t)2 * r], ev[(size_t)2 * r + 1]));
        if (kernel_ms_out) kernel_ms_out[r] = ms;
        h->last_kernel_ms = ms;
    }
    float tot_ms = 0.f;
    HS_HIP(h, hipEventElapsedTime(&tot_ms, ev[(size_t)repeats * 2], ev[(size_t)repeats * 2 + 1]));
    if (total_ms_out) *total_ms_out = tot_ms;
    h->last_run_ms = tot_ms / (float)repeats;
    if (!h->is_net) h->launches = 2;
    for (auto &e : ev) hipEventDestroy(e);
    Totals t;                                                       // a timed run that overflowed is not a result
    HS_HIP(h, hipMemcpy(&t, h->tot, sizeof t, hipMemcpyDeviceToHost));
    if (t.qoverflow) return fail(h, HS_E_UNSUPPORTED, "a same-timestamp event cascade exceeded the in-group queue");
    if (t.overflow & 8) return fail(h, HS_E_HIP, "the asynchronous network engine gave up waiting for a neighbour (bounded spin)");
    if (t.overflow & 2) return fail(h, HS_E_OVERFLOW, "a station's in-flight message bag overflowed (capacity %d); raise bag_capacity", (int)h->NX.bag_cap);
    if (t.overflow) return fail(h, HS_E_OVERFLOW, "a per-LP record log overflowed (capacity %lld records)", (long long)h->L.cap);
    return HS_OK;
}

int hs_engine_get_summary(hs_engine *h, hs_summary *out) {
    if (!h || !out) return fail(h, HS_E_INVALID, "hs_engine_get_summary: null argument");
    if (!h->have_stations) return fail(h, HS_E_STATE, "stations not set");
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
                               h->cfg.n_lp, lp, cnt);
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
        hipLaunchKernelGGL(hs_gather_logs, grid, dim3(256), 0, h->stream, cols[c], h->X.received, d_off, d_out, (int)n, cap);
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
                               h->cfg.n_lp, lp, cnt);
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
    for (void *p : h->allocs) hipFree(p);
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
