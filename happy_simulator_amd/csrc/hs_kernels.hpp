// hs_kernels.hpp -- the engine's kernels (gfx950).  Included by hs_engine.hip (host side + the C ABI; HS_KERNELS_MAIN: it
// also defines the non-template kernels) and by hs_inst.hip, which is compiled once per group of template instantiations
// (-DHS_INST=k) so that the big kernels build in parallel; hs_engine.hip only DECLARES those instantiations (extern template).
// Replaces `Simulation._execute_until` (happysimulator/core/simulation.py:449-505) for station LPs and station networks;
// kernels, data layout and rooflines are described in DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hs_engine.h"
#include "hs_netstation.hpp"
#ifdef HS_KERNELS_MAIN
#include "hs_exact.hpp"
#endif

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
    // the reference's (time, _sort_index): of two events on one nanosecond the one created first; of two created in one
    // nanosecond the one fewer steps from the root of the group that created it, then the one whose root was created first
    // (the heap is a FIFO inside a nanosecond: StationState lineage); then the construction order
    if (a.t_created != b.t_created) return a.t_created < b.t_created;
    if (a.depth != b.depth) return a.depth < b.depth;
    if (a.rcrt != b.rcrt) return a.rcrt < b.rcrt;
    return a.rank < b.rank;
}
__device__ __forceinline__ Candidate cand_load_agent(const Candidate *p) {
    Candidate c;
    c.t = __hip_atomic_load(&p->t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.t_created = __hip_atomic_load(&p->t_created, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.rcrt = __hip_atomic_load(&p->rcrt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.depth = __hip_atomic_load(&p->depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.lp = __hip_atomic_load(&p->lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.rank = __hip_atomic_load(&p->rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.valid = __hip_atomic_load(&p->valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.pad = __hip_atomic_load(&p->pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (what the candidate is: the shards' network-wide re-ranking reads it, ShardCtl::cand_out[7])
    c.pad2 = __hip_atomic_load(&p->pad2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (hs_net_window: the workgroup's equal-key peers of its best candidate)
    return c;
}
__device__ __forceinline__ Candidate cand_none(int lp) {
    Candidate c;
    c.valid = 0; c.t = kInfNs; c.t_created = 0; c.rcrt = INT64_MIN; c.depth = 0; c.lp = lp; c.rank = lp; c.pad = 0; c.pad2 = 0;
    return c;
}

__device__ __forceinline__ Candidate wave_min_cand(Candidate c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Candidate d;
        d.t = shfl_xor_ll(c.t, o);
        d.t_created = shfl_xor_ll(c.t_created, o);
        d.rcrt = shfl_xor_ll(c.rcrt, o);
        d.depth = __shfl_xor(c.depth, o, 64);
        d.pad = __shfl_xor(c.pad, o, 64); d.pad2 = 0;
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
    S.prof_kind = kProfConstant; S.tab_a = nullptr; S.tab_cap = 0;
    S.n_probes = 0; S.evp[0] = S.evp[1] = 0;
#pragma unroll
    for (int j = 0; j < kMaxProbes; ++j) { S.p_metric[j] = kProbeNone; S.PA[j] = kInfNs; S.seqP[j] = 0; S.crtP[j] = 0; S.p_arr[j] = 0; S.p_n[j] = 0; S.tab_p[j] = nullptr; }
    S.SA = kInfNs; S.sc_i = S.sc_end = 0; S.sc_t = P.sched_t; S.sc_idx = P.sched_idx;
    S.n_xsrc = 0; S.x_base = P.stream_base[lp];
#pragma unroll
    for (int j = 0; j < kMaxXSrc; ++j) { S.x_kind[j] = 0; S.XA[j] = kInfNs; S.seqX[j] = 0; S.crtX[j] = 0; S.x_arr[j] = 0; S.x_n[j] = 0; S.x_k[j] = 0; S.x_rate[j] = 1.0; S.x_stop[j] = -1; S.dpX[j] = 0; S.rcX[j] = INT64_MIN; }
#pragma unroll
    for (int j = 0; j < kMaxProbes; ++j) S.rcP[j] = INT64_MIN;
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
                    S.x_k[j] = X.x_k[o]; S.dpX[j] = X.dpX[o]; S.rcX[j] = X.rcX[o];
                }
            }
        }
        if (P.sched_off != nullptr) {
            S.sc_i = X.sched_i[lp]; S.sc_end = P.sched_off[lp + 1];
            S.SA = S.sc_i < S.sc_end ? P.sched_t[S.sc_i] : kInfNs;
        }
        S.prof_kind = P.prof_kind[lp];
        if (P.tabs != nullptr) {
            S.tab_cap = P.tabs->cap;
            const int32_t row = P.tabs->src_row[lp];
            S.tab_a = row >= 0 ? P.tabs->times + (size_t)row * (size_t)S.tab_cap : nullptr;
        }
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) {
            const size_t o = (size_t)j * n + lp;
            S.p_metric[j] = P.probe_metric[o];
            if (S.p_metric[j] != kProbeNone) {
                S.n_probes = j + 1;
                S.tab_p[j] = P.tabs->times + (size_t)P.tabs->probe_row[o] * (size_t)S.tab_cap;
                S.PA[j] = X.PA[o]; S.seqP[j] = X.seqP[o]; S.crtP[j] = X.crtP[o]; S.p_arr[j] = X.p_arr[o]; S.p_n[j] = X.p_n[o];
                S.rcP[j] = X.rcP[o];
            }
        }
        S.probe_t = L.probe_t + lp; S.probe_v = L.probe_v + lp; S.pcap = L.pcap;
    }
    S.trk = false; S.t_start = INT64_MIN; S.n_up = 0; S.cur_pay = 0; S.undecided = 0; S.wk = 0; S.rk_wk = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) S.wkD[i] = PF ? (int32_t)X.wkD[(size_t)i * n + lp] : 1;
#pragma unroll
    for (int u = 0; u < kMaxUp; ++u) { S.U[u].up = -1; S.U[u].i = 0; S.U[u].n = 0; S.U[u].IA = kInfNs; S.U[u].mask = 0; }
    S.rk_dp = 0; S.rk_rank = 0; S.rk_rc = INT64_MIN; S.tie_rank_p = P.tie_rank; S.n_rank = n;
    S.lsrc = 255u; S.rs_qp = nullptr; S.rs_dp = nullptr;
#pragma unroll
    for (int i = 0; i < C; ++i) S.lsrcD[i] = 255u;
    if constexpr (PF) {
        if (P.tabs != nullptr && P.tabs->rs_dep != nullptr) {            // several Sources per Server: the lineage's Source (TickTables::rs_dep)
            S.rs_qp = P.tabs->rs_q + lp; S.rs_dp = P.tabs->rs_dep + lp;
#pragma unroll
            for (int i = 0; i < C; ++i) S.lsrcD[i] = P.tabs->rs_dep[(size_t)i * n + lp];
        }
        if (P.tabs != nullptr && P.tabs->tandem != nullptr) {            // tandem queues (hs_station.hpp `trk`)
            const TickTables &T = *P.tabs;
            S.trk = true; S.t_start = T.t_start;
            S.fw_rc = T.fw_rc + lp; S.fw_rrc = T.fw_rrc + lp; S.fw_rdr = T.fw_rdr + lp; S.fw_dep = T.fw_dep + lp;
            S.q_rrc = T.q_rrc + lp; S.q_rdr = T.q_rdr + lp; S.q_pay = T.q_pay + lp;
#pragma unroll
            for (int u = 0; u < kMaxUp; ++u) {
                const int up = T.tandem[(size_t)u * n + lp];
                if (up < 0) continue;
                S.n_up = u + 1;                                          // (the lists are filled from 0)
                auto &Lu = S.U[u];
                Lu.up = up;
                Lu.i_p = T.inj_i + (size_t)u * n + lp; Lu.i = *Lu.i_p;
                Lu.n = X.received[up];                                   // forwards it has published (its pass is over)
                Lu.n = Lu.n < L.cap ? Lu.n : L.cap;
                Lu.t = L.sink_t + up; Lu.created = ((C > 1) ? L.sink_created : L.adm) + up;
                Lu.rc = T.fw_rc + up; Lu.rrc = T.fw_rrc + up; Lu.rdr = T.fw_rdr + up; Lu.dep = T.fw_dep + up;
                Lu.IA = Lu.i < Lu.n ? Lu.t[(size_t)Lu.i * (size_t)n] : kInfNs;
            }
        }
    }
    S.A = X.A[lp]; S.seqA = X.seqA[lp]; S.crtA = X.crtA[lp]; S.arr_time = X.arr_time[lp];
    S.buf = X.buf[lp]; S.active = X.active[lp]; S.seq = X.seq[lp];
    S.generated = X.generated[lp]; S.accepted = X.accepted[lp]; S.dropped = X.dropped[lp];
    S.completed = X.completed[lp]; S.rejected = X.rejected[lp]; S.started = X.started[lp];
    S.received = X.received[lp]; S.sink_w = X.sink_w[lp];
    S.total_service = X.total_service[lp];
    S.last_time = X.last_time[lp]; S.grp_time = X.grp_time[lp];
    S.dpA = X.dpA[lp]; S.rcA = X.rcA[lp]; S.cd = 0; S.cr = INT64_MIN;
    S.qdep = X.qdep + lp; S.qrc = X.qrc + lp;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        S.D[i] = X.D[(size_t)i * n + lp]; S.seqD[i] = X.seqD[(size_t)i * n + lp];
        S.crtD[i] = X.crtD[(size_t)i * n + lp]; S.svc_s[i] = X.svc_s[(size_t)i * n + lp];
        S.crt[i] = (C > 1) ? X.crt[(size_t)i * n + lp] : 0;
        S.dpD[i] = X.dpD[(size_t)i * n + lp]; S.rcD[i] = X.rcD[(size_t)i * n + lp];
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
    // (a group that an earlier window stopped inside: the lineage of its <= 2 waiting events is already in slots 0, 1 of the
    // lineage columns -- store_station put it there)
    for (int i = 0; i < qn; ++i) { S.qmem[i][tid] = (uint8_t)((q >> (8 * i)) & 0xffu); }
    S.qn = qn;
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
    X.dpA[lp] = (uint8_t)S.dpA; X.rcA[lp] = S.rcA;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = S.D[i]; X.seqD[(size_t)i * n + lp] = S.seqD[i];
        X.crtD[(size_t)i * n + lp] = S.crtD[i]; X.svc_s[(size_t)i * n + lp] = S.svc_s[i];
        if (C > 1) X.crt[(size_t)i * n + lp] = S.crt[i];
        X.dpD[(size_t)i * n + lp] = (uint8_t)S.dpD[i]; X.rcD[(size_t)i * n + lp] = S.rcD[i];
    }
    if constexpr (PF) {
        if (S.rs_dp != nullptr) {
#pragma unroll
            for (int i = 0; i < C; ++i) S.rs_dp[(size_t)i * n] = (uint8_t)S.lsrcD[i];
        }
    }
    X.arr_k[lp] = S.arr_k; X.svc_k[lp] = S.svc_k;   // draws CONSUMED; pre-drawn values still in the rings are dropped
    uint32_t q = 0;
    int qn = S.qn > 2 ? 2 : S.qn;   // an overshoot root leaves at most two in-group events
    if (qn > 0 && S.qh != 0) {      // ... whose lineage moves to slots 0, 1 (where load_station expects it)
        uint8_t d0 = S.qdep[(size_t)(S.qh % kQCap) * S.ls], d1 = S.qdep[(size_t)((S.qh + 1) % kQCap) * S.ls];
        int64_t r0 = S.qrc[(size_t)(S.qh % kQCap) * S.ls], r1 = S.qrc[(size_t)((S.qh + 1) % kQCap) * S.ls];
        S.qdep[0] = d0; S.qrc[0] = r0;
        if (qn > 1) { S.qdep[(size_t)S.ls] = d1; S.qrc[(size_t)S.ls] = r1; }
        if constexpr (PF) {
            if (S.trk) {
                int64_t *cols[3] = {S.q_rrc, S.q_rdr, S.q_pay};
                for (int c = 0; c < 3; ++c) {
                    const int64_t v0 = cols[c][(size_t)(S.qh % kQCap) * S.ls], v1 = cols[c][(size_t)((S.qh + 1) % kQCap) * S.ls];
                    cols[c][0] = v0;
                    if (qn > 1) cols[c][(size_t)S.ls] = v1;
                }
            }
        }
    }
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
            X.rcP[o] = S.rcP[j];
        }
        X.ev_probe[lp] += S.evp[0]; X.ev_probe[(size_t)n + lp] += S.evp[1];
        tot += S.evp[0] + S.evp[1];
        if (S.sc_t != nullptr) X.sched_i[lp] = S.sc_i;
#pragma unroll
        for (int i = 0; i < C; ++i) X.wkD[(size_t)i * n + lp] = (uint8_t)S.wkD[i];
        if (S.trk) {                                       // (mask == 0 between groups: a run of forwards is consumed whole)
#pragma unroll
            for (int u = 0; u < kMaxUp; ++u) if (u < S.n_up) *S.U[u].i_p = S.U[u].i;
        }
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) if (j < S.n_xsrc) {
            const size_t o = (size_t)j * n + lp;
            X.XA[o] = S.XA[j]; X.seqX[o] = S.seqX[j]; X.crtX[o] = S.crtX[j]; X.x_arr[o] = S.x_arr[j]; X.x_n[o] = S.x_n[j];
            X.x_k[o] = S.x_k[j]; X.dpX[o] = (uint8_t)S.dpX[j]; X.rcX[o] = S.rcX[j];
        }
    }
    X.events[lp] += tot;
}

// first pending event of an LP: time, creation time, lineage, which root
template <int C, bool PF, bool UNI = false>
__device__ __forceinline__ Candidate make_candidate(const Station<C, PF, UNI> &S) {
    Candidate c = cand_none(S.lp);
    if (S.qn > 0) {   // a group already in progress keeps the floor
        c.t = S.grp_time; c.t_created = S.grp_time; c.valid = 1;
        return c;
    }
    const int64_t t = S.next_time();
    if (t == kInfNs) return c;
    const int w = S.pick_root(t);
    c.t = t; c.valid = 1;
    c.t_created = S.root_crt(w);
    if (w == 0) { c.depth = S.dpA; c.rcrt = S.rcA; c.pad = 2; }
    else if (PF && w >= kRootXSrc && w < kRootInj) {
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) if (j == w - kRootXSrc) { c.depth = S.dpX[j]; c.rcrt = S.rcX[j]; }
        c.pad = 3 + (w - kRootXSrc);
    }
    else if (PF && w >= kRootProbe && w < kRootInj) {
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) if (j == w - kRootProbe) c.rcrt = S.rcP[j];
        c.depth = 1;            // a Probe's next tick is created by its tick, always a root
        c.pad = 8 + (w - kRootProbe);   // ranked by the Probe's own position in `probes=[...]`, behind every Source (cand_rank)
    }
    else if (PF && w == kRootSched) { c.depth = 0; c.rcrt = INT64_MIN; }   // constructed before run()
    else {
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == w - 1) { c.depth = S.dpD[i]; c.rcrt = S.rcD[i]; if (PF && S.rs_dp != nullptr && S.lsrcD[i] != 255u) c.pad = 2 + (int)S.lsrcD[i]; }
        c.pad2 = 1;     // a DEPARTURE: whatever Source its rank borrows (the lineage's: a rule of thumb, tools/election_rules.py), a tie of
                        // the whole lineage key with another LP's candidate goes to the single heap (hs_station_run's tie check; round 6:
                        // multi_source case 130100 of tools/gpu_random_sweep.py is a counter-example to the rule)
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
        tot->overflow = 0; tot->qoverflow = 0; tot->done = 0; tot->undecided = 0;
        tot->dbg[0] = tot->dbg[1] = tot->dbg[2] = tot->dbg[3] = 0; tot->not_done = 0;
        tot->pend_lp = -1; tot->no_resume = 0;
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
            if constexpr (PF) arr_time = P.tabs->times[(size_t)P.tabs->src_row[lp] * (size_t)P.tabs->cap];   // tick 0 of its table
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
    X.dpA[lp] = 0; X.rcA[lp] = INT64_MIN;     // lineage: the first tick was constructed before run()
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = kInfNs; X.seqD[(size_t)i * n + lp] = 0; X.crtD[(size_t)i * n + lp] = start_ns;
        X.svc_s[(size_t)i * n + lp] = 0.0; X.crt[(size_t)i * n + lp] = 0;
        X.dpD[(size_t)i * n + lp] = 0; X.rcD[(size_t)i * n + lp] = INT64_MIN; X.wkD[(size_t)i * n + lp] = 1;
        if constexpr (PF) { if (P.tabs != nullptr && P.tabs->rs_dep != nullptr) P.tabs->rs_dep[(size_t)i * n + lp] = 255; }
    }
    for (int k = 0; k < 11; ++k) X.ev_kind[(size_t)k * n + lp] = 0;
    if constexpr (!PF) return;
    if constexpr (PF) {                                    // tandem: no forward consumed yet
        if (P.tabs != nullptr && P.tabs->tandem != nullptr) for (int u = 0; u < kMaxUp; ++u) P.tabs->inj_i[(size_t)u * n + lp] = 0;
    }
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
            X.dpX[o] = 0; X.rcX[o] = INT64_MIN;
            if (NX.next_time != nullptr && XA < NX.next_time[lp]) NX.next_time[lp] = XA;   // network engine: first pending event
        }
    }
    if (X.PA != nullptr) {   // probes start after the sources (core/simulation.py:156-160): first tick from start_ns
        uint32_t stamp = 1 + kMaxXSrc;
        for (int j = 0; j < kMaxProbes; ++j) {
            const size_t o = (size_t)j * n + lp;
            int64_t PA = kInfNs, p_arr = start_ns;
            if (P.probe_metric[o] != kProbeNone) {
                if constexpr (PF) PA = P.tabs->times[(size_t)P.tabs->probe_row[o] * (size_t)P.tabs->cap];     // tick 0 of its table
                p_arr = 0;                                                                                     // ... whose index it is
            }
            if (NX.next_time != nullptr && PA < NX.next_time[lp]) NX.next_time[lp] = PA;   // network engine: first pending event
            X.PA[o] = PA; X.seqP[o] = stamp++; X.crtP[o] = start_ns; X.p_arr[o] = p_arr; X.p_n[o] = 0; X.rcP[o] = INT64_MIN;
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
    mine = cand_none(lp);
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
                S.prof_kind = kProfConstant;
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
        S.force_general = (flags & 1) != 0 || (PF && S.trk && (S.n_up > 0 || S.egress == kEgressServer));   // (tandem LPs: event order)
    }
    const bool ended = (mode == HS_MODE_REPLICAS) ? (live && S.last_time > end_ns) : (cur > end_ns);
    // tandem queues run in passes, upstream Servers first (hs_station.hpp `trk`): flags bits 28..30 = pass + 1; an LP of another
    // pass keeps its state in this launch, and still names its first event beyond end_ns in the launch that elects
    bool other_pass = false;
    if constexpr (PF) {
        const int pass1 = (flags >> 28) & 7;
        if (pass1 != 0 && live && P.tabs != nullptr && P.tabs->tandem != nullptr) other_pass = P.tabs->tandem[(size_t)kMaxUp * n + lp] != pass1 - 1;
    }
    const bool frozen = ended || other_pass;
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
                S.force_general = (flags & 1) != 0 || (PF && S.trk && (S.n_up > 0 || S.egress == kEgressServer));   // (tandem LPs: event order)
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
        if (!ended) {
            if (mode == HS_MODE_REPLICAS) overshoot_one<C, PF, UNI>(S);
            else {
                mine = make_candidate<C, PF, UNI>(S);
                mine.rank = cand_rank(P, lp, n, mine.pad);
            }
        }
        store_station<C, PF, UNI>(S, X, lp, n);
        if constexpr (PF) {
            if (P.tabs != nullptr && P.tabs->cand_key != nullptr) {
                // tandem queues / several Sources per Server: what the election's tie check reads (TickTables::cand_key); bit 33: the
                // candidate's construction rank is a stand-in (a departure or injected Request of an LP with several Sources)
                int64_t *ck = P.tabs->cand_key + (size_t)lp * 4;
                ck[0] = mine.t; ck[1] = mine.t_created; ck[2] = mine.rcrt;
                ck[3] = (int64_t)(uint32_t)mine.depth | ((int64_t)(mine.valid ? 1 : 0) << 32) |
                        ((int64_t)((mine.valid && (mine.pad < 2 || mine.pad2 != 0) && (S.n_xsrc > 0 || P.tabs->standin_sched != 0)) ? 1 : 0) << 33);
            }
            if (S.undecided) atomicOr(&tot->undecided, S.undecided);
        }
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
    best = cand_none(0);
    if (!producer) {
        for (int b = tid; b < (int)gridDim.x; b += kBlock) {
            const Candidate c = cand_load_agent(&cands[b]);
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
        wave_c[0] = b;   // (for the tie check below)
    }
    if constexpr (PF) {
        // Tandem queues: did the election rest on the construction rank?  Then another LP's candidate shares the winner's whole
        // lineage key, and only the two events' ancestry decides which the reference pops first (Totals::undecided).
        // Several Sources per Server (no tandem queues): the same, but only where one of the two tying candidates ranks with a stand-in
        // (bit 33 of its key word: hs_engine_set_stations).
        if (P.tabs != nullptr && P.tabs->cand_key != nullptr && cur <= end_ns) {
            __syncthreads();
            const Candidate b = wave_c[0];
            const bool any_tie = P.tabs->tandem != nullptr;
            bool tie = false;
            if (b.valid) {
                const int64_t bv = __hip_atomic_load(&P.tabs->cand_key[(size_t)b.lp * 4 + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int q = tid; q < n; q += kBlock) {
                    const int64_t *ck = P.tabs->cand_key + (size_t)q * 4;
                    const int64_t t = __hip_atomic_load(&ck[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t cr = __hip_atomic_load(&ck[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t rc = __hip_atomic_load(&ck[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t dv = __hip_atomic_load(&ck[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (q != b.lp && ((dv >> 32) & 1) != 0 && t == b.t && cr == b.t_created && rc == b.rcrt && (int)(uint32_t)dv == b.depth &&
                        (any_tie || (((dv | bv) >> 33) & 1) != 0)) tie = true;
                }
            }
            if (tie) atomicOr(&tot->undecided, 2);
        }
    }
}

#ifdef HS_KERNELS_MAIN   // (non-template kernels: defined by hs_engine.hip only)
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
                NX.bag_lin[d] = lin_pack(e.dep, e.rcrt, e.ts);
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


#endif  // HS_KERNELS_MAIN

// ---------------------------------------------------------------------------------------------
// network engine (hs_netstation.hpp): one launch per conservative time window
// ---------------------------------------------------------------------------------------------
namespace {

template <int C, bool FAST = false, bool PF = !FAST, bool UNI = false>
__device__ __forceinline__ void load_net(NetStation<C, FAST, PF, UNI> &S, const StationParams &P, const NetParams &NP,
                                         const StationState &X, const NetState &NX, const RecordLogs &L, int lp, int n,
                                         uint8_t (*qmem)[kBlock], int64_t *enq, size_t enq_stride, int tid, int send_idx,
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
    S.A = X.A[lp]; S.seqA = X.seqA[lp]; S.crtA = X.crtA[lp]; S.arr_time = X.arr_time[lp]; S.arr_d = (double)S.arr_time;
    S.buf = X.buf[lp]; S.active = X.active[lp]; S.seq = X.seq[lp];
    S.generated = X.generated[lp]; S.accepted = X.accepted[lp]; S.dropped = X.dropped[lp];
    S.completed = X.completed[lp]; S.rejected = X.rejected[lp]; S.started = X.started[lp];
    S.received = X.received[lp]; S.routed = NX.routed[lp];
    S.total_service = X.total_service[lp];
    S.last_time = X.last_time[lp];
    S.dpA = X.dpA[lp]; S.rcA = X.rcA[lp]; S.cd = 0; S.cr = INT64_MIN;
    S.qdep = X.qdep + lp; S.qrc = X.qrc + lp;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        S.D[i] = X.D[(size_t)i * n + lp]; S.seqD[i] = X.seqD[(size_t)i * n + lp];
        S.crtD[i] = X.crtD[(size_t)i * n + lp]; S.svc_s[i] = X.svc_s[(size_t)i * n + lp];
        S.crt[i] = X.crt[(size_t)i * n + lp];
        S.dpD[i] = X.dpD[(size_t)i * n + lp]; S.rcD[i] = X.rcD[(size_t)i * n + lp];
    }
    const uint64_t base = P.stream_base[lp];
    S.arr.init(S.seed, stream_id(base, kStreamArrival), X.arr_k[lp]);
    S.svc.init(S.seed, stream_id(base, kStreamService), X.svc_k[lp]);
    S.rte.init(S.seed, stream_id(S.route_base, kStreamRoute), NX.route_k[lp]);
#pragma unroll
    for (int k = 0; k < 11; ++k) S.ev[k] = 0;
    // (network engines keep their record logs LP-major, RecordLogs::lp_major: record k of this LP at [lp * cap + k])
    const size_t log0 = L.lp_major ? (size_t)lp * (size_t)L.cap : (size_t)lp;
    S.adm = L.adm + log0;
    S.sink_t = L.sink_t + log0;
    S.sink_created = L.sink_created + log0;
    S.cap = L.cap; S.ls = n; S.lgs = L.lp_major ? 1 : n;
    S.overflow = 0; S.qoverflow = 0; S.bagoverflow = 0;
    S.np = &NP; S.ns = &NX; S.send_idx = send_idx;
    S.tid = tid;
    S.inc_const = __ddiv_rn(1.0, S.rate);
    S.n_probes = 0; S.evp[0] = S.evp[1] = 0; S.pcap = 0; S.probe_t = nullptr; S.probe_v = nullptr;
#pragma unroll
    for (int j = 0; j < kMaxProbes; ++j) { S.p_metric[j] = kProbeNone; S.PA[j] = kInfNs; S.seqP[j] = 0; S.crtP[j] = 0; S.p_arr[j] = 0; S.p_n[j] = 0; S.tab_p[j] = nullptr; S.rcP[j] = INT64_MIN; }
    S.prof_kind = kProfConstant; S.tab_a = nullptr; S.tab_cap = P.tabs != nullptr ? P.tabs->cap : 0;
    S.undecided = 0;
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
            S.tab_a = P.tabs->times + (size_t)P.tabs->src_row[lp] * (size_t)S.tab_cap;
        }
        if (X.PA != nullptr) {      // probes on a network (PF instantiations of both engines)
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) {
                const size_t o = (size_t)j * n + lp;
                S.p_metric[j] = P.probe_metric[o];
                if (S.p_metric[j] != kProbeNone) {
                    S.n_probes = j + 1;
                    S.tab_p[j] = P.tabs->times + (size_t)P.tabs->probe_row[o] * (size_t)S.tab_cap;
                    S.PA[j] = X.PA[o]; S.seqP[j] = X.seqP[o]; S.crtP[j] = X.crtP[o]; S.p_arr[j] = X.p_arr[o]; S.p_n[j] = X.p_n[o];
                    S.rcP[j] = X.rcP[o];
                }
            }
            S.probe_t = L.probe_t + lp; S.probe_v = L.probe_v + lp; S.pcap = L.pcap;
        }
    }
    S.ha = S.na = S.hs_ = S.nsv = S.hj = S.nj = S.rn = 0; S.rbits = 0; S.fl_link = -1; S.fi_link = -1; S.fi_packets = 0;
    S.fl_remote = false; S.fi_head = 0; S.fi_unpub = 0; S.fl_local = false; S.fi_local = false; S.win_hi = S.started;
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
            S.bag_insert(NX.bag_t[b], NX.bag_ts[b], NX.bag_cr[b], NX.bag_link[b], NX.bag_lin[b]);
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
            // jitter = ConstantLatency(m): + Duration.from_seconds(m).to_seconds(), no draw (link.py:195-200); m == 0: none
            if (S.fl_jit != 0) S.fl_delay0 = __dadd_rn(S.fl_delay0, seconds_from_ns(ns_from_seconds(NP.link_jit_mean[l])));
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
            if (k < S.accepted) { S.fl.crc[k & (kNRing - 1)][tid] = k < S.cap ? S.adm[k * S.lgs] : 0; S.win_hi = k + 1; }
        }
    }
    S.bmin = S.bag_scan_min();
    S.qmem = qmem; S.enq = enq; S.enq_stride = enq_stride; S.qh = 0; S.qn = 0; S.ph = 0; S.pn = 0;
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
    X.dpA[lp] = (uint8_t)S.dpA; X.rcA[lp] = S.rcA;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        X.D[(size_t)i * n + lp] = S.D[i]; X.seqD[(size_t)i * n + lp] = S.seqD[i];
        X.crtD[(size_t)i * n + lp] = S.crtD[i]; X.svc_s[(size_t)i * n + lp] = S.svc_s[i];
        X.crt[(size_t)i * n + lp] = S.crt[i];
        X.dpD[(size_t)i * n + lp] = (uint8_t)S.dpD[i]; X.rcD[(size_t)i * n + lp] = S.rcD[i];
    }
    // draws CONSUMED (pre-drawn values still in the FAST rings are dropped: pure functions of the index)
    X.arr_k[lp] = S.arr_consumed(); X.svc_k[lp] = S.svc_consumed(); NX.route_k[lp] = S.rte_consumed();
    if constexpr (FAST) {
        for (int i = 0; i < S.bag_n; ++i) {
            const size_t b = (size_t)lp * NX.bag_cap + i;
            NX.bag_t[b] = S.bg_t(i); NX.bag_ts[b] = S.bg_ts(i); NX.bag_cr[b] = S.bg_cr(i); NX.bag_link[b] = S.bg_link(i);
            NX.bag_lin[b] = S.bg_lin(i);
        }
        if (S.fl_link >= 0) {
            NX.link_in[S.fl_link] = S.fl_in; NX.link_sent[S.fl_link] = S.fl_sent;
            NX.link_k[S.fl_link] = S.jit.k - (uint64_t)S.nj;
        }
        if (S.fi_link >= 0) { NX.link_packets[S.fi_link] = S.fi_packets; NX.aq_head[S.fi_link] = S.fi_head; }   // (head_taken publishes lazily)
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
                X.rcP[o] = S.rcP[j];
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
                const int64_t t = ag_load(&NX.aq_rec[4 * slot]);
                NX.bag_t[dst] = t; NX.bag_ts[dst] = ag_load(&NX.aq_rec[4 * slot + 1]); NX.bag_cr[dst] = ag_load(&NX.aq_rec[4 * slot + 2]);
                NX.bag_lin[dst] = ag_load(&NX.aq_rec[4 * slot + 3]);
                NX.bag_link[dst] = l;
                nt = t < nt ? t : nt;
                ++bn;
            }
            NX.aq_head[l] = head;
        }
        NX.bag_cnt[lp] = bn;
        NX.next_time[lp] = nt;
        // (windows: the next run_until continues from this state when hs_net_async can take every bag back into LDS)
        if (bn > kLBag) __hip_atomic_fetch_or(&tot->no_resume, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                NX.bag_t[dst] = t; NX.bag_ts[dst] = NX.in_ts[src]; NX.bag_cr[dst] = NX.in_cr[src]; NX.bag_lin[dst] = NX.in_lin[src];
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
    mine = cand_none(lp);
    if (act) {
        load_net<C>(S, P, NP, X, NX, L, lp, n, qmem, &enqpay[0][tid], (size_t)kBlock, tid, send_idx, SC);
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
                mine.t_created = S.root_crt(w);
                if (w == 1) { mine.depth = S.dpA; mine.rcrt = S.rcA; mine.pad = 2; }
                else if (w >= 64) { const int64_t lin = S.bg_lin(w - 64); mine.depth = lin_steps(lin); mine.rcrt = lin_root(lin, mine.t_created); }
                else if (w >= 56 && w < 56 + kMaxProbes) {
#pragma unroll
                    for (int j = 0; j < kMaxProbes; ++j) if (j == w - 56) mine.rcrt = S.rcP[j];
                    mine.depth = 1;
                    mine.pad = 8 + (w - 56);                      // a Probe's tick: by the Probe's own position in `probes=[...]`
                }
                else if (w >= 48 && w < 48 + kMaxXSrc) {
                    const size_t o = (size_t)(w - 48) * n + lp;
                    mine.depth = X.dpX[o]; mine.rcrt = X.rcX[o]; mine.pad = 3 + (w - 48);
                }
                else if (w == 62) { mine.depth = 0; mine.rcrt = INT64_MIN; }   // constructed before run()
                else {
#pragma unroll
                    for (int i = 0; i < C; ++i) if (i == w - 2) { mine.depth = S.dpD[i]; mine.rcrt = S.rcD[i]; }
                }
                mine.rank = cand_rank(P, lp, n, mine.pad);        // ties on (time, creation time): `sources=[...]` order
            }
        }
        store_net<C>(S, X, NX, lp, n);
        if (S.undecided) atomicOr(&tot->undecided, S.undecided);
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

    // The workgroup's best candidate, and (for the election's tie check below) whether ANOTHER station of this workgroup has a
    // candidate with the same (time, creation time, steps from the root, root's creation time): Candidate::pad2 bit 0 -- any such
    // peer, bit 1 -- one that ranks with a stand-in (pad < 2: a departure, a message, an injected Request).
    Candidate bb = wave_c[0];
    for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], bb)) bb = wave_c[w];
    {
        const bool peer = bb.valid && mine.valid && mine.lp != bb.lp && mine.t == bb.t && mine.t_created == bb.t_created &&
                          mine.rcrt == bb.rcrt && mine.depth == bb.depth;
        const int peer_any = __syncthreads_or((int)peer), peer_standin = __syncthreads_or((int)(peer && mine.pad < 2));
        bb.pad2 = (peer_any ? 1 : 0) | (peer_standin ? 2 : 0);
    }
    if (tid == 0) {
        cands[blockIdx.x] = bb;
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
    best = cand_none(0);
    for (int b = tid; b < (int)gridDim.x; b += kBlock) {
        const Candidate c = cand_load_agent(&cands[b]);
        if (cand_less(c, best)) best = c;
    }
    best = wave_min_cand(best);
    if ((tid & 63) == 0) wave_c[tid >> 6] = best;
    __syncthreads();
    Candidate b = wave_c[0];
    for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
    // Did the election rest on the construction rank of a STAND-IN (round 5)?  Another station's candidate shares
    // the winner's (time, creation time, steps from the root, root's creation time), and one of the two ranks with its station's
    // first-listed Source instead of the Source its lineage goes back to (a departure, a message, an injected Request): only the
    // two events' ancestry decides which the reference pops first.  Totals::undecided bit 1 -- hs_engine_run_until refuses the
    // result by name (lock-step constant arrivals, services and link latencies; never seen on a random workload) instead of
    // guessing; a shard reports it in bit 1 of its candidate's first word.
    // (round 6: from the workgroups' summaries -- a station with the winner's key is its workgroup's best candidate or one of that
    //  candidate's peers; until then the last workgroup read every station's key, 65 536 x 32 bytes through 256 lanes: most of this launch)
    bool tie = false;
    if (b.valid && cur0 <= wend) {
        const bool b_standin = b.pad < 2;
        for (int k = tid; k < (int)gridDim.x; k += kBlock) {
            const Candidate c = cand_load_agent(&cands[k]);
            if (!c.valid || c.t != b.t || c.t_created != b.t_created || c.rcrt != b.rcrt || c.depth != b.depth) continue;
            if (c.lp != b.lp && (b_standin || c.pad < 2)) tie = true;            // another workgroup's best is a peer
            if ((c.pad2 & 2) || (b_standin && (c.pad2 & 1))) tie = true;         // ... or one of that best's own peers is
        }
    }
    const int any_tie = __syncthreads_or((int)tie);
    if (tid == 0) {
        if (any_tie) atomicOr(&tot->undecided, 2);
        long long new_cur = __hip_atomic_load(&tot->final_time, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flags & 4) {
            // sharded network: the election continues across the ranks on the host (hs_engine_shard_overshoot runs
            // the winner); publish this rank's candidate
            SC.cand_out[0] = (b.valid ? 1 : 0) | (any_tie ? 2 : 0); SC.cand_out[1] = b.t; SC.cand_out[2] = b.t_created;
            SC.cand_out[3] = SC.lp_base + b.lp; SC.cand_out[4] = b.depth; SC.cand_out[5] = b.rcrt; SC.cand_out[6] = b.rank; SC.cand_out[7] = b.pad;   // [6] is shard-LOCAL: the host re-ranks by (station, [7]) network-wide
            tot->cur_time = new_cur;
            tot->done = 0;
            return;
        }
        if (cur0 > wend) new_cur = cur0;              // ... its one event beyond end_time included
        else if (b.valid) {
            // the one event beyond end_time (core/simulation.py:472): first micro-event of the winner's next group
            NetStation<C> W;
            load_net<C>(W, P, NP, X, NX, L, b.lp, n, qmem, &enqpay[0][0], (size_t)kBlock, 0, send_idx, SC);
            // (NetStation::root_first: what the event created stays in the in-group FIFO, unprocessed -- and is kept for the next
            //  run_until, which finishes the group before anything else: StationState::q / grp_time, NetState::pend_pay, hs_net_resume)
            const int64_t t = W.next_time();
            const int w = W.pick_root(t);
            const bool egress = W.root_first(w, t);
            W.last_time = t;
            bool fits = true;
            const uint32_t pend = W.pending_pack(egress, egress ? w - 2 : 0, fits);
            X.q[b.lp] = pend; X.grp_time[b.lp] = t;
            if (NX.pend_pay != nullptr) NX.pend_pay[b.lp] = W.pending_payload();
            tot->pend_lp = pend != 0u ? b.lp : -1;
            if (!fits || NX.pend_pay == nullptr) tot->no_resume |= 2;
            store_net<C>(W, X, NX, b.lp, n);
            if (W.undecided) atomicOr(&tot->undecided, W.undecided);           // (a pre-run root beside another root of the nanosecond)
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

// Windows over a network (core/simulation.py:527-541): the previous run_until's election stopped inside a timestamp group -- the one
// event beyond its end_time was the group's first micro-event.  This launch (one lane) finishes that group from what the election
// kept (NetStation::resume_pending); it is the network's earliest pending work, so nothing else can come before it.  Messages it
// sends go where the engine that runs next looks for them: the links' queues (NX.aq_on) or the incoming bags of parity `send_idx`.
template <int C>
__global__ void __launch_bounds__(64) hs_net_resume(StationParams P, NetParams NP, StationState X, NetState NX, RecordLogs L,
                                                    Totals *tot, int n, int send_idx, ShardCtl SC) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ int64_t enqpay[kEnqPay][kBlock];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int lp = tot->pend_lp;
    tot->pend_lp = -1;
    if (lp < 0 || lp >= n) return;
    const uint32_t pend = X.q[lp];
    X.q[lp] = 0;
    if (pend == 0u) return;
    NetStation<C> W;
    load_net<C>(W, P, NP, X, NX, L, lp, n, qmem, &enqpay[0][0], (size_t)kBlock, 0, send_idx, SC);
    const int64_t t = X.grp_time[lp];
    W.resume_pending(pend, t, NX.pend_pay[lp]);
    store_net<C>(W, X, NX, lp, n);
    for (int k = 0; k < 11; ++k) if (W.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)W.ev[k]);
    if (W.evp[0]) atomicAdd(&tot->ev[13], (unsigned long long)W.evp[0]);
    if (W.evp[1]) atomicAdd(&tot->ev[14], (unsigned long long)W.evp[1]);
    if (W.ev[6]) atomicAdd(&tot->completed, (unsigned long long)W.ev[6]);
    if (W.ev[7]) atomicAdd(&tot->received, (unsigned long long)W.ev[7]);
    atomicMax(&tot->final_time, (long long)t);
    if (W.overflow) atomicOr(&tot->overflow, 1);
    if (W.qoverflow) atomicOr(&tot->qoverflow, 1);
    if (W.bagoverflow) atomicOr(&tot->overflow, 2);
    if (W.undecided) atomicOr(&tot->undecided, W.undecided);
    if (NX.aq_on && W.sent_async) {
        // the links' (bound, tail) words: the bound stays (it covered this message), the tail is the link's sequence number
        drain_stores();
        const int nt = W.egress == EG_LINK ? 1 : W.egress == EG_ROUTER ? W.rtk : 0;
        for (int k = 0; k < nt; ++k) {
            const int32_t l = W.egress == EG_LINK ? W.link_of : W.rt_target(k);
            if (l < 0) continue;
            const int64_t w = ag_load(&NX.aq_ea[l]);
            ag_store(&NX.aq_ea[l], (int64_t)(((unsigned long long)w & ~kPkTailMask) | ((unsigned long long)NX.link_sent[l] & kPkTailMask)));
        }
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
    // (the ENQ payloads of the general path live in global memory here -- X.enqpay, [kEnqPay][n_lp]: the LDS they had is the
    // bags' lineage column)
    __shared__ double ring_a[kNRing][kBlock], ring_s[kNRing][kBlock], ring_j[kNRing][kBlock];   // pre-drawn E values
    __shared__ int64_t lbag_t[kLBag][kBlock], lbag_ts[kLBag][kBlock], lbag_cr[kLBag][kBlock];   // the bags, in LDS
    __shared__ int64_t lbag_lin[kLBag][kBlock];
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
    // (generic instantiations: bits 8.. of `lanes` = the first station of this launch in workgroups -- a network with more
    //  stations than one cooperative launch holds runs in SEGMENTS that take turns, hs_engine.hip run_net_segments)
    int lp0 = 0;
#ifndef HS_NO_SEG
    if constexpr (!UNI) { lp0 = (lanes >> 8) * kBlock; lanes &= 0xff; }
#endif
    const int lp = lp0 + (blockIdx.x * (kBlock / 64) + (tid >> 6)) * lanes + lane;
    const bool live = lane < lanes && lp < n;
    if (tid < 14) red[tid] = 0;
    if (tid == 0) { red_time = INT64_MIN; red_flags[0] = red_flags[1] = red_flags[2] = red_flags[3] = 0; }
    __syncthreads();

    NetStation<C, true, PF, UNI> S;
    S.fl = NetFastLds{ring_a, ring_s, ring_j, lbag_t, lbag_ts, lbag_cr, lbag_lin, lbag_link, crc};
    bool done = !live;
    int gave_up = 0;
    if (live) {
        load_net<C, true, PF, UNI>(S, P, NP, X, NX, L, lp, n, qmem, X.enqpay + lp, (size_t)n, tid, 0, SC);
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
        const bool live = !UNI && SC.live != 0;               // ... LIVE exchange: to the queue in the destination rank's memory
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
        // Round 6: the words of a link INSIDE a wavefront stay out of memory while the launch runs.  A chain lane's input bound is
        // the scan's (from the sender's current state: never below what that sender last published) and its tail is the sender's
        // appended count, handed over by a shuffle (tail_hint) -- so it does not load the link's (bound, tail) word, and its sender
        // does not store it until the launch ends (the final launch, the next round / segment and load_net read the tail from
        // memory).  On the 65 536-station ring 63 of 64 links are such links: ~365 polls and publications per link and run were
        // 2.5 GB of the kernel's 5.9 GB of HBM traffic (1.78 GB algorithmic).  Debug flag 1 << 21 keeps every word in memory.
        const bool quiet = (flags & (1 << 21)) == 0;
        const bool next_chain = quiet && next_l >= 0 && lane + 1 < lanes && __shfl_down(chain ? 1 : 0, 1, 64) != 0;
        bool pub_skipped = false;
        // ... and such a link's records move at workgroup scope (hs_netstation.hpp wg_store_rec): both ends are lanes of this wavefront
#ifndef HS_NO_WG_SCOPE   // (scratch build: every record at device scope, as until round 5)
        S.fl_local = next_chain && next_l == S.fl_link;
        S.fi_local = chain && quiet && my_in == S.fi_link;
#endif
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
#ifdef HS_CYCLES
            const unsigned long long qs = __builtin_readcyclecounter();
#endif
            // the incoming link's word: its latency hides behind the refills (a chain lane: "nothing new" -- bound 0, tail = head)
            const int64_t w_peek = done ? 0 : (chain && quiet) ? (int64_t)(S.fi_head & kPkTailMask) : S.async_peek();
            // Three memory round trips of an iteration run behind the wave-level refill of the pre-drawn stream values instead of in
            // front of it: the word of the incoming link (above) and the created_at window's entries behind its first half, the first
            // two waiting messages' payloads behind its second half (round 5, with the two-value top-ups of hs_netstation.hpp: 7.39 -> 6.25 ms on the 65 536-station ring).
            int64_t wf0, wf1;                                         // created_at of what entered the window from a deep queue:
            const int wf_n = S.window_issue(!done, wf0, wf1);         // loaded before, stored behind the refill
            {   // (a chain lane's only incoming link is the previous lane's next_l: async_receive_one)
                const long long q_next = next_l >= 0 ? (long long)S.link_sent_of(next_l) : 0ll;
                const long long q_prev = __double_as_longlong(dpp_or0<0x138, 0xf>(__longlong_as_double(q_next)));   // wave_shr:1
                S.tail_hint = chain ? (unsigned long long)q_prev : 0ull;
            }
            S.top_up(!done, topup_need, 1);                           // whole wavefront: refill the pre-drawn values (arrivals, services)
            const bool split_rx = !done && S.fi_link >= 0;            // one incoming link, held in registers (FAST)
            int64_t rx[8];
            unsigned long long rx_tail = 0ull;
            const int rx_n = split_rx ? S.async_receive_issue(w_peek, rx, rx_tail) : 0;
            S.top_up(!done, topup_need, 2);                           // ... jitter, routing decisions
            S.window_commit(!done, wf_n, wf0, wf1);
            int64_t H = kInfNs;
#ifdef HS_CYCLES
            const unsigned long long q0 = __builtin_readcyclecounter();
#endif
            if (split_rx) H = S.async_receive_commit(w_peek, rx_n, rx, rx_tail);
            else if (!done) H = S.async_receive(w_peek);                    // messages below H are all in the bag now
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
            if (!chain) {                                             // head of a chain: its input bound is known
                if (!done && S.undrained < H) H = S.undrained;
                int64_t v = sat(H, mB);
                v = v > mD ? v : mD;
                mA = v < mA ? v : mA;
                mB = kInfNs; mD = INT64_MIN;
            }
#ifdef HS_SCAN_I64   // (scratch build: the scan on int64 through ds_bpermute shuffles, as until round 5)
            auto satd = [](int64_t d, int64_t b) { return d == INT64_MIN ? INT64_MIN : (b == kInfNs ? kInfNs : d + b); };
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
#else
            // Round 6: the prefix composition in whole-ns binary64 on the VALU's own lane movement (DPP row shifts / row broadcasts
            // with the identity map (+inf, 0, -inf) where a lane has no source: no per-lane selects).  Times travel as offsets from
            // pk_base -- whole numbers below 2^51, exact in binary64, +-infinity for "never" / "no floor" -- so a saturating 64-bit
            // add is ONE v_add_f64 and a 64-bit min / max one instruction instead of a compare and two selects; the six steps were
            // 36 ds_bpermute round trips and ~200 integer instructions per iteration on a wavefront that is alone on its SIMD.
            {
                constexpr double kTwo51 = 2251799813685248.0, kTwo52 = 4503599627370496.0;
                const double kPosInf = __longlong_as_double(0x7ff0000000000000ll), kNegInf = __longlong_as_double((long long)0xfff0000000000000ull);
                // (clamping a time up to pk_base and a finite value down to 2^51 - 1 is conservative: nothing happens before the start)
                auto whole_d = [&](int64_t r) {
                    r = r < 0 ? 0 : (r > (int64_t)kTwo51 - 1 ? (int64_t)kTwo51 - 1 : r);
                    return __longlong_as_double(r | 0x4330000000000000ll) - kTwo52;
                };
                double fA = mA == kInfNs ? kPosInf : whole_d(mA - NX.pk_base);
                double fB = mB == kInfNs ? kPosInf : whole_d(mB);
                double fD = mD == INT64_MIN ? kNegInf : mD == kInfNs ? kPosInf : whole_d(mD - NX.pk_base);
                auto after = [&](double pA, double pB, double pD) {      // (fA, fB, fD) o (pA, pB, pD)
                    const double v = __builtin_fmax(pA + fB, fD);
                    fA = __builtin_fmin(v, fA);
                    const double d2 = pD == kNegInf ? kNegInf : pD + fB;   // (-inf + inf: the floor that does not exist stays none)
                    fD = __builtin_fmax(d2, fD);
                    fB = pB + fB;
                };
                after(dpp_orpos<0x111, 0xf>(fA), dpp_or0<0x111, 0xf>(fB), dpp_orneg<0x111, 0xf>(fD));   // row_shr:1, 2, 4, 8
                after(dpp_orpos<0x112, 0xf>(fA), dpp_or0<0x112, 0xf>(fB), dpp_orneg<0x112, 0xf>(fD));
                after(dpp_orpos<0x114, 0xf>(fA), dpp_or0<0x114, 0xf>(fB), dpp_orneg<0x114, 0xf>(fD));
                after(dpp_orpos<0x118, 0xf>(fA), dpp_or0<0x118, 0xf>(fB), dpp_orneg<0x118, 0xf>(fD));
                after(dpp_orpos<0x142, 0xa>(fA), dpp_or0<0x142, 0xa>(fB), dpp_orneg<0x142, 0xa>(fD));   // rows 1, 3 <- the row before
                after(dpp_orpos<0x143, 0xc>(fA), dpp_or0<0x143, 0xc>(fB), dpp_orneg<0x143, 0xc>(fD));   // rows 2, 3 <- rows 0-1
                const double hp_d = dpp_orpos<0x138, 0xf>(fA);        // wave_shr:1: the previous lane's bound towards me
                if (chain) {
                    const int64_t hp = hp_d == kPosInf ? kInfNs : NX.pk_base + i64_from_whole_d(__builtin_fmin(hp_d, kTwo51));
                    if (hp > H) H = hp;
                }
                if (!done && S.undrained < H) H = S.undrained;        // ... never beyond what is still sitting in a queue
            }
#endif
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
                    // (a link that leaves the shard: an outbox row takes what a round sends -- LIVE exchange: the queue in the peer's memory)
                    if (act && !(((out_remote[0] && !live) || S.async_can_send(out_l[0], head_seen[0], out_remote[0])) &&
                                 (UNI || (out_remote[1] && !live) || S.async_can_send(out_l[1], head_seen[1], out_remote[1])))) { blocked = true; act = false; }   // a consumer is behind: wait
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
                        if (l == next_l && next_chain) pub_skipped = true;      // (its reader is the next lane: the scan and tail_hint)
                        else if (live && out_remote[o]) S.live_publish(l, pk_pack(vo[o], (unsigned long long)S.link_sent_of(l), NX.pk_base));
                        else ag_store(&NX.aq_ea[l], pk_pack(vo[o], (unsigned long long)S.link_sent_of(l), NX.pk_base));
                        out_pub[o] = vo[o];
                    }
                }
                S.sent_async = false;
                done = base > end_ns;                                 // nothing at or before end_ns can happen any more
            }
#ifdef HS_CYCLES
            {
                const unsigned long long q4 = __builtin_readcyclecounter();
                cyc[0] += q0 - qs; cyc[1] += q2 - q0; cyc[2] += q3 - q2;   // refills | receive + bound scan | groups | publication
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
            // (LIVE exchange: the neighbour may be another process's launch that never came -- give up after ~a second of idling)
            if (idle_iters >= (live ? (1u << 19) : kAsyncMaxIter)) { gave_up = 1; atomicOr(&tot->overflow, 8); break; }
            // Bounded buffers + time-ordered processing can deadlock on a cycle: every LP waits for room in its outgoing
            // queue while its own bag is full of messages it may not process yet.  A wavefront that makes no progress for
            // a long time while one of its lanes waits for buffer space reports that (raise bag_capacity) instead of spinning.
            if (wave_idle && __any(!done && blocked)) {
                if (++blocked_iters >= kAsyncBlockedMax) { atomicOr(&tot->overflow, 2); aborted = true; }
            } else blocked_iters = 0;
            if ((flags & 128) && wave_idle) __builtin_amdgcn_s_sleep(64);   // experiment: back off when idle
            groups_before = n_groups;
        }
        // The links' words as the launch leaves them.  (i) an in-wavefront link's word, once (payloads: drained in the loop).  (ii) A
        // later window end continues from these words (hs_engine_run_until_async), and a bound may rest on neighbours that were DONE:
        // in the scan a lane that is done contributes "never" -- right for this launch, not for the next.  What is certain is that
        // nothing has been sent at or before end_ns: every word is capped at end_ns + 1 + the link's transit floor (round 6; found on
        // the full-size ring only: 3 - 16 of 65 536 stations, all at wavefront / DPP-row boundaries, differed in the second window).
#pragma unroll
        for (int o = 0; o < kOut; ++o) {
            const int32_t l = out_l[o];
            if (l < 0 || out_remote[o]) continue;
            const int64_t cap = sat(end_ns + 1, out_lat[o]);
            if (out_pub[o] > cap || (pub_skipped && l == next_l))
                ag_store(&NX.aq_ea[l], pk_pack(out_pub[o] < cap ? out_pub[o] : cap, (unsigned long long)S.link_sent_of(l), NX.pk_base));
        }
        store_net<C, true, PF, UNI>(S, X, NX, lp, n);
        if constexpr (PF) { if (S.undecided) atomicOr(&tot->undecided, S.undecided); }
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

#ifdef HS_KERNELS_MAIN   // (non-template kernels: defined by hs_engine.hip only)
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
    const int64_t *m = rowp + 1 + kMsgWords * (size_t)i;
    const int64_t dst = (m[3] >> 32) - lp_base, gid = m[3] & 0xffffffffll;
    if (dst < 0 || dst >= n || gid >= n_gid || gid2local[gid] < 0) { atomicOr(&tot->overflow, 4); return; }
    const size_t cslot = (size_t)send_idx * n + (size_t)dst;
    const int pos = atomicAdd(&NX.in_cnt[cslot], 1);
    if (pos < NX.bag_cap) {
        const size_t b = cslot * NX.bag_cap + pos;
        NX.in_t[b] = m[0]; NX.in_ts[b] = m[1]; NX.in_cr[b] = m[2]; NX.in_link[b] = gid2local[gid]; NX.in_lin[b] = m[4];
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
            const int64_t *m = rowp + 1 + kMsgWords * (size_t)i;
            const int64_t dst = (m[3] >> 32) - lp_base, gid = m[3] & 0xffffffffll;
            if (dst < 0 || dst >= n || gid >= n_gid || gid2local[gid] < 0) { atomicOr(&tot->overflow, 4); continue; }
            const int l = gid2local[gid];
            const unsigned long long head = NX.aq_head[l], tail = pk_tail(NX.aq_ea[l], head);
            if (tail - head >= (unsigned long long)NX.aq_cap) { atomicOr(&tot->overflow, 2); continue; }
            const size_t slot = (size_t)l * NX.aq_cap + (size_t)(tail & (unsigned long long)(NX.aq_cap - 1));
            NX.aq_rec[4 * slot] = m[0]; NX.aq_rec[4 * slot + 1] = m[1]; NX.aq_rec[4 * slot + 2] = m[2]; NX.aq_rec[4 * slot + 3] = m[4];
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

// ---- device-side exchange between the rounds (round 5): peers' buffers mapped with hipIpcOpenMemHandle ---------------------
// Instead of an all-to-all of whole outbox rows and an all-reduce of the bounds (staged through the host when the transport
// cannot move device tensors), every rank WRITES what it has for rank d straight into rank d's memory -- its row of d's inbox
// (only the messages that exist) and its row of d's bound table -- over xGMI peer-to-peer stores (the same device when the
// ranks share one).  The buffers are double-buffered by round parity, so ONE barrier per round orders everything: the
// stream-ordered all-reduce of the one "still working" word (RCCL), the only collective left on the path.  Peer memory is
// written with system-scope stores and read with system-scope loads (remote writes reach HBM past the reader's L2).
// One workgroup per destination rank.
__global__ void __launch_bounds__(256) hs_shard_push(const int64_t *outbox, int row, int msg_cap, int world, int rank, int parity,
                                                     int64_t *const *peer_inbox, int64_t *const *peer_bounds,
                                                     const int64_t *bounds, int n_cross, Totals *tot) {
    const int d = blockIdx.x;
    if (d >= world) return;
    const int64_t *src = outbox + (size_t)d * row;
    int64_t cnt = src[0];
    if (cnt > msg_cap) { if (threadIdx.x == 0) atomicOr(&tot->overflow, 2); cnt = msg_cap; }
    int64_t *dst = peer_inbox[d] + ((size_t)parity * world + (size_t)rank) * (size_t)row;
    const int64_t words = 1 + (int64_t)kMsgWords * cnt;
    for (int64_t i = threadIdx.x; i < words; i += blockDim.x)
        __hip_atomic_store(&dst[i], i == 0 ? cnt : src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int64_t *bd = peer_bounds[d] + ((size_t)parity * world + (size_t)rank) * (size_t)(n_cross + 1);
    for (int i = threadIdx.x; i <= n_cross; i += blockDim.x)
        __hip_atomic_store(&bd[i], bounds[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
}
// After the barrier: the element-wise maximum of what the ranks pushed (= the all-reduce(MAX) of the collective path) into the
// engine's own bound vector, which hs_shard_inject_async and hs_engine_shard_async_done read as before.
__global__ void __launch_bounds__(256) hs_shard_combine_bounds(const int64_t *ipc_bounds, int world, int parity, int n_cross, int64_t *bounds) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_cross; i += gridDim.x * blockDim.x) {
        int64_t m = INT64_MIN;
        for (int r = 0; r < world; ++r) {
            const int64_t v = __hip_atomic_load(&ipc_bounds[((size_t)parity * world + (size_t)r) * (size_t)(n_cross + 1) + i],
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            m = v > m ? v : m;
        }
        bounds[i] = m;
    }
}
// ... and the pushed rows copied out of the uncached exchange buffer (system-scope loads) into the engine's own inbox
__global__ void __launch_bounds__(256) hs_shard_fetch_inbox(const int64_t *ipc_inbox, int world, int parity, int row, int msg_cap, int64_t *inbox) {
    const int r = blockIdx.x;
    if (r >= world) return;
    const int64_t *src = ipc_inbox + ((size_t)parity * world + (size_t)r) * (size_t)row;
    __shared__ int64_t s_cnt;
    if (threadIdx.x == 0) s_cnt = __hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    int64_t cnt = s_cnt;
    if (cnt > msg_cap) cnt = msg_cap;
    if (cnt < 0) cnt = 0;
    int64_t *dst = inbox + (size_t)r * row;
    const int64_t words = 1 + (int64_t)kMsgWords * cnt;
    for (int64_t i = threadIdx.x; i < words; i += blockDim.x)
        dst[i] = i == 0 ? s_cnt : __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif  // HS_KERNELS_MAIN

// Sharded network: this rank owns the globally first event beyond end_ns -- process it (core/simulation.py:472).
template <int C>
__global__ void hs_shard_overshoot(StationParams P, NetParams NP, StationState X, NetState NX, RecordLogs L,
                                   Totals *tot, int n, int lp, int win, ShardCtl SC) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ int64_t enqpay[kEnqPay][kBlock];
    if (threadIdx.x != 0) return;
    NetStation<C> W;
    load_net<C>(W, P, NP, X, NX, L, lp, n, qmem, &enqpay[0][0], (size_t)kBlock, 0, win & 1, SC);
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

#ifdef HS_KERNELS_MAIN
// Read-back of the record logs.  The logs are [cap][n_lp] (record k of LP lp at k * n + lp) so that the run kernels'
// appends coalesce; the C ABI hands records out per LP, concatenated in LP order.  64 x 64 tiles through LDS: coalesced
// reads along lp, coalesced writes along k.
__global__ void __launch_bounds__(256) hs_gather_logs(const int64_t *__restrict__ log, const int64_t *__restrict__ cnt,
                                                      const int64_t *__restrict__ off, int64_t *__restrict__ out, int n,
                                                      int64_t cap, int lp_major) {
    __shared__ int64_t tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int lp0 = blockIdx.x * 64;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    if (lp_major) {                 // network engines: every LP's records are contiguous already -- a copy of 64 LPs x 64 records
        for (int l = ty; l < 64; l += 4) {
            const int lp = lp0 + l;
            if (lp >= n) continue;
            int64_t c = cnt[lp];
            c = c > cap ? cap : c;
            const int64_t k = k0 + tx;
            if (k < c) out[off[lp] + k] = log[(size_t)lp * cap + k];
        }
        return;
    }
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
__global__ void hs_gather_one(const int64_t *__restrict__ log, int64_t *__restrict__ out, int n, int lp, int64_t cnt, int64_t lp_major_cap) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // lp_major_cap > 0: an LP-major log of that capacity
    if (k < cnt) out[k] = lp_major_cap > 0 ? log[(size_t)lp * lp_major_cap + k] : log[(size_t)k * n + lp];
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
#endif  // HS_KERNELS_MAIN

#include "hs_kernels_wide.hpp"
#include "hs_kernels_wave.hpp"

// ---------------------------------------------------------------------------------------------
// The template instantiations the host launches, in build groups (hs_inst.hip -DHS_INST=k defines group k).
// ---------------------------------------------------------------------------------------------
#define HS_ARGS_STATION_RUN (StationParams, StationState, RecordLogs, Totals *, Candidate *, int, int64_t, int, int)
#define HS_ARGS_NET_WINDOW (StationParams, NetParams, StationState, NetState, RecordLogs, Totals *, Candidate *, int, int64_t, int, int, ShardCtl)
#define HS_ARGS_NET_ASYNC (StationParams, NetParams, StationState, NetState, RecordLogs, Totals *, int, int64_t, int, ShardCtl, int, int)
#define HS_ARGS_SHARD_OVERSHOOT (StationParams, NetParams, StationState, NetState, RecordLogs, Totals *, int, int, int, ShardCtl)
#define HS_ARGS_NET_RESUME (StationParams, NetParams, StationState, NetState, RecordLogs, Totals *, int, int, ShardCtl)
#define HS_INST_GROUP_0(X) X(hs_station_run<1, false, true, true> HS_ARGS_STATION_RUN) X(hs_station_run<1, false, true> HS_ARGS_STATION_RUN)
#define HS_INST_GROUP_1(X) X(hs_station_run<1, true> HS_ARGS_STATION_RUN) X(hs_station_run<1, false> HS_ARGS_STATION_RUN) \
                           X(hs_station_run<2, true> HS_ARGS_STATION_RUN) X(hs_station_run<2, false> HS_ARGS_STATION_RUN)
#define HS_INST_GROUP_2(X) X(hs_station_run<4, true> HS_ARGS_STATION_RUN) X(hs_station_run<4, false> HS_ARGS_STATION_RUN) \
                           X(hs_station_run<8, true> HS_ARGS_STATION_RUN) X(hs_station_run<8, false> HS_ARGS_STATION_RUN)
#define HS_INST_GROUP_3(X) X(hs_station_run<16, true> HS_ARGS_STATION_RUN) X(hs_station_run<16, false> HS_ARGS_STATION_RUN)
#define HS_INST_GROUP_4(X) X(hs_station_run<32, true> HS_ARGS_STATION_RUN) X(hs_station_run<32, false> HS_ARGS_STATION_RUN)
#define HS_INST_GROUP_5(X) X(hs_net_window<1> HS_ARGS_NET_WINDOW) X(hs_shard_overshoot<1> HS_ARGS_SHARD_OVERSHOOT) X(hs_net_resume<1> HS_ARGS_NET_RESUME)
#define HS_INST_GROUP_6(X) X(hs_net_window<2> HS_ARGS_NET_WINDOW) X(hs_shard_overshoot<2> HS_ARGS_SHARD_OVERSHOOT) X(hs_net_resume<2> HS_ARGS_NET_RESUME)
#define HS_INST_GROUP_7(X) X(hs_net_window<4> HS_ARGS_NET_WINDOW) X(hs_shard_overshoot<4> HS_ARGS_SHARD_OVERSHOOT) X(hs_net_resume<4> HS_ARGS_NET_RESUME)
#define HS_INST_GROUP_8(X) X(hs_net_async<1, false, true> HS_ARGS_NET_ASYNC)
#define HS_INST_GROUP_9(X) X(hs_net_async<1, false> HS_ARGS_NET_ASYNC)
#define HS_INST_GROUP_10(X) X(hs_net_async<1, true> HS_ARGS_NET_ASYNC)
#define HS_INST_GROUP_11(X) X(hs_net_async<2, false> HS_ARGS_NET_ASYNC) X(hs_net_async<2, true> HS_ARGS_NET_ASYNC)
#define HS_INST_GROUP_12(X) X(hs_net_async<4, false> HS_ARGS_NET_ASYNC) X(hs_net_async<4, true> HS_ARGS_NET_ASYNC)
#define HS_ARGS_WIDE (StationParams, StationState, RecordLogs, Totals *, Candidate *, WideCtl *, int32_t *, int, int64_t, int)
#define HS_INST_GROUP_13(X) X(hs_station_wide<4> HS_ARGS_WIDE) X(hs_station_wide<8> HS_ARGS_WIDE)
#define HS_INST_GROUP_14(X) X(hs_station_wide<16> HS_ARGS_WIDE)
#define HS_ARGS_WAVE (StationParams, StationState, RecordLogs, Totals *, Candidate *, WideCtl *, int32_t *, WavePart *, int, int64_t, int, int64_t)
#define HS_INST_GROUP_15(X) X(hs_station_wave<16, false> HS_ARGS_WAVE) X(hs_station_wave<8, false> HS_ARGS_WAVE)
#define HS_INST_GROUP_16(X) X(hs_station_wave<16, true> HS_ARGS_WAVE) X(hs_station_wave<8, true> HS_ARGS_WAVE)
#define HS_INST_GROUPS 17
#define HS_INST_ALL(X) HS_INST_GROUP_0(X) HS_INST_GROUP_1(X) HS_INST_GROUP_2(X) HS_INST_GROUP_3(X) HS_INST_GROUP_4(X) HS_INST_GROUP_5(X) \
    HS_INST_GROUP_6(X) HS_INST_GROUP_7(X) HS_INST_GROUP_8(X) HS_INST_GROUP_9(X) HS_INST_GROUP_10(X) HS_INST_GROUP_11(X) HS_INST_GROUP_12(X) \
    HS_INST_GROUP_13(X) HS_INST_GROUP_14(X) HS_INST_GROUP_15(X) HS_INST_GROUP_16(X)
#define HS_DECLARE_INST(...) extern template __global__ void __VA_ARGS__;
#define HS_DEFINE_INST(...) template __global__ void __VA_ARGS__;
#ifdef HS_KERNELS_MAIN
HS_INST_ALL(HS_DECLARE_INST)
#endif
