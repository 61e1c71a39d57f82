// hs_kernels_wave.hpp -- the headline grid with ONE WAVEFRONT PER LP: the strong shard of `_execute_until`
// (core/simulation.py:449-505) when the metric's 65 536 servers are block-partitioned over 8 GPUs (8 192 LPs per device).
//
// hs_station_wide<K> (hs_kernels_wide.hpp) spreads an LP over K <= 16 lanes and keeps ONE part serial in the request index: the
// arrival chain  a' = from_seconds(to_seconds(a) + E / rate)  (load/arrival_time_provider.py:72-82), walked redundantly by every
// lane of the group -- three roles a step apart, each a chain of dependent instructions: ~76 us at any size.
// Here nothing is serial in the request index but one binary64 addition:
//   V  lane j draws the step's requests 2j and 2j + 1: one Philox4x32-10 block per stream, hs_log, the constant-divisor quotients;
//   C  the arrival chain as an EXACT PREFIX SUM over speculated whole-nanosecond steps (the technique of hs_lb.hip lb_step_encode):
//      with F = RN(inc * 1e9) the three roundings of the reference's step stay below 4e-4 ns while a + F < 2^40, so whenever
//      frac(F) lies in [2^-10, 1 - 2^-10] the tick is exactly a + floor(F) -- a wave-wide inclusive scan of whole numbers held in
//      binary64 (exact below 2^52).  The one increment in ~500 that lies closer to a whole number is resolved one by one: the
//      reference's own ten-instruction step at that position, its difference added to every later position;
//   L  the Lindley recursion  D_k = max(a_k, D_{k-1}) + s_k  as a wave-wide scan in the (max, +) semiring (hs_kernels_wide.hpp), the
//      reference's events of request k counted with ballots exactly as Station::req_step does (hs_station.hpp);
//   T  `_total_service_time += s` in completion order (server/server.py:252-273): one dependent binary64 addition per completed
//      request -- the only serial part left; ONE wavefront of the workgroup sums for all its LPs, a lane each, out of LDS.
// A workgroup is NW wavefronts = NW neighbouring LPs.  The record logs are [record][LP] (a lane-per-LP wavefront appends 512
// contiguous bytes); a wavefront that owns ONE LP would write 128 words 8 n_lp bytes apart, each a partial line shared with 15
// other LPs -- so a step's records are staged in LDS and the workgroup writes them out transposed: 16 neighbouring LPs' k-th
// records are one 128-byte line.
// Same-nanosecond hazards make the LP bail to hs_station_wide_finish (event-order loop), which also adds up the workgroups' event
// partials (no same-address atomics here: 8 192 wavefronts adding to eight words would serialise) and elects the one event beyond
// end_ns.  Bit-identical to hs_station_run and hs_station_wide (tests/test_gpu_wide.py).
#pragma once

constexpr int kWaveR = 128;            // requests of an LP per step: two per lane
constexpr int kWaveRowPad = kWaveR + 2;

namespace {

// lane i <- lane i - 1 across the whole wavefront (v_mov_b32_dpp wave_shr:1, gfx9); lane 0 keeps its own value
__device__ __forceinline__ double wave_shr1(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// (dpp_or0 / dpp_orneg / dpp_orv: hs_device.hpp)
// f(x) = max(p, x + q) after g: hs_kernels_wide.hpp mp_compose with v_max_f64 (the values are never NaN)
__device__ __forceinline__ MaxPlus mp_after(const MaxPlus &f2, const MaxPlus &f1) {
    return MaxPlus{__builtin_fmax(f2.p, f1.p + f2.q), f1.q + f2.q};
}
template <int CTRL, int RM>
__device__ __forceinline__ MaxPlus mp_scan_step(const MaxPlus &F) {   // F after the map of the lane CTRL names (identity where there is none)
    return mp_after(F, MaxPlus{dpp_orneg<CTRL, RM>(F.p), dpp_or0<CTRL, RM>(F.q)});
}
// the value of lane `l` (wave-uniform index) as a wave-uniform value (two v_readlane)
__device__ __forceinline__ double rl64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double rfl64(double v) {     // ... of the first active lane
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// wave-uniform copies of a value every lane read from one LDS address (v_readfirstlane: the value lives in SGPRs afterwards)
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ int64_t uni(int64_t v) { return (int64_t)uni((uint64_t)v); }
__device__ __forceinline__ double uni(double v) { return rfl64(v); }

// What an LP's wavefront starts from (WaveIn) and what it hands to the fold (WaveOut).  Both cross the workgroup through LDS, a lane
// per LP on wavefront 0 at the other end: the bootstrap / the state loads before the first step and the fold + state stores behind
// the last are wave-uniform work -- run by every wavefront for its own LP they cost 64 lanes' issue slots for one lane's worth of
// values (~850 of the ~5 500 VALU instructions of a 60 s window), and sixteen wavefronts' single-lane stores of one field are
// sixteen partial lines where a lane per LP writes one.
struct WaveIn {
    int64_t A, crtA, Dprev, Sprev, accepted, started, sink_w, last_time, rcA, rcD;
    uint64_t ak0, sk0;
    double total_service, svc_s0, carry_fresh, rate_b, rate_y, lam_b, lam_y;
    uint32_t key0, key1, asid0, asid1, ssid0, ssid1, seq, seqA, seqD;
    int32_t dpA, dpD;
    int32_t bits;                       // 1 busy, 2 eligible, 4 div_rate.fast, 8 div_lambda.fast
};
struct WaveOut {
    int64_t r0, lt;                     // arrivals of the window; the latest processed event
    uint32_t n_tick, n_start, n_dep, n_notify, n_poll;
    int32_t bits;                       // 1 pend, 2 pend_new, 4 count (ran, did not bail), 8 bailed, 16 overflow, 32 live
};

// Station::req_finish for one LP (lane `l` of wavefront 0): folds the window into the LP's state and stores it, lists a bailed LP,
// and returns the LP's event counts, latest time and candidate for the one event beyond end_ns.
template <bool FRESH>
__device__ __forceinline__ void wave_fold(const StationParams &P, StationState &X, const RecordLogs &L, WideCtl *ctl, int32_t *bail, int n, int lp,
                                          int64_t start_ns, const WaveIn &I, const double *pendv, const double *tick, const WaveOut &O,
                                          double total_service, unsigned (&ev)[8], long long &lt_out, Candidate &mine) {
    auto to_i64 = [](double d) {                        // exact for whole d in [0, 2^52)
        return (int64_t)((uint64_t)__double_as_longlong(__dadd_rn(d, 4503599627370496.0)) & 0xFFFFFFFFFFFFFull);
    };
    const bool live = (O.bits & 32) != 0, count = (O.bits & 4) != 0, bailed = (O.bits & 8) != 0, pend_new = (O.bits & 2) != 0;
    bool pend = (O.bits & 1) != 0;
    const int64_t A = I.A, crtA0 = I.crtA, accepted = I.accepted, started = I.started, lt = O.lt;
    int64_t pendD = I.Dprev, pendS = I.Sprev, pend_i = 0;
    double pend_s = I.svc_s0, pendA_d = 0.0, pendAp_d = 0.0, pendSp_d = 0.0;
    if (pend_new) {
        pend = true;
        pendD = to_i64(pendv[0]); pendS = to_i64(pendv[1]); pendA_d = pendv[2]; pendAp_d = pendv[3];
        pend_i = to_i64(pendv[4]); pend_s = pendv[5]; pendSp_d = pendv[6];
    }
    const double A_next = tick[0], a_last = tick[1], a_last2 = tick[2];
    const int64_t n_arr_total = O.r0;
    mine = cand_none(live ? lp : 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) ev[k] = 0;
    // what the LP's state becomes (FRESH: for an LP that did not run, or bailed, what the reset leaves)
    uint32_t c_tick = 0, c_start = 0, c_dep = 0, seq = I.seq, seqA = I.seqA, seqD = I.seqD;
    int64_t acc2 = accepted, st2 = started, A_next_i = A, crtA2 = crtA0, rcA = I.rcA, rcD = I.rcD;
    int32_t dpA = I.dpA, dpD = I.dpD;
    if (count) {
        c_tick = O.n_tick; c_start = O.n_start; c_dep = O.n_dep;
        const uint32_t c_notify = O.n_notify, c_poll = O.n_poll;
        ev[0] = c_tick; ev[1] = c_tick; ev[2] = c_notify; ev[3] = c_poll + c_dep; ev[4] = c_start; ev[5] = c_start; ev[6] = c_dep; ev[7] = c_dep;
        acc2 = accepted + c_tick; st2 = started + c_start;
        A_next_i = A != kInfNs ? to_i64(A_next) : A;
        crtA2 = c_tick ? to_i64(a_last) : crtA0;
        if ((c_tick | c_start) != 0u) {                  // creation stamps: only their order matters (Station::req_finish)
            const bool d_first = pend && pendS < crtA2;
            seqA = seq + (d_first ? 1u : 0u); seqD = seq + (d_first ? 0u : 1u); seq += 2u;
        }
        // lineage of what is pending now (Station::req_finish)
        if (c_tick) { dpA = 1; rcA = acc2 >= 2 ? (c_tick >= 2 ? to_i64(a_last2) : L.adm[(size_t)(acc2 - 2) * n + lp]) : crtA0; }
        if (pend && pend_new) {
            const int64_t m = st2 - 1;                   // the request in service: it started at pendS
            if (pendS == to_i64(pendA_d)) {              // ... on arrival: six steps from its tick, which was created at the tick before
                dpD = 6;
                rcD = m >= 1 ? (pend_i >= 1 ? to_i64(pendAp_d) : (m - 1 < L.cap ? L.adm[(size_t)(m - 1) * n + lp] : 0)) : crtA0;
            } else {                                     // ... when request m - 1 left: four steps from that continuation, created when IT started
                dpD = 4; rcD = to_i64(pendSp_d);         // (= pendS - its service time: Station::req_finish draws that again)
            }
        }
        // this LP's candidate for the one event beyond end_ns (make_candidate / pick_root: creation stamps decide a tie)
        const int64_t Dn = pend ? pendD : kInfNs;
        const int64_t tmin = A_next_i < Dn ? A_next_i : Dn;
        if (tmin != kInfNs) {
            const bool tick_first = A_next_i < Dn || (A_next_i == Dn && (int32_t)(seqA - seqD) < 0);
            mine.t = tmin; mine.valid = 1;
            if (tick_first) { mine.t_created = crtA2; mine.depth = dpA; mine.rcrt = rcA; mine.pad = 2; }
            else { mine.t_created = pend ? pendS : 0; mine.depth = dpD; mine.rcrt = rcD; mine.pad = 0; }
            mine.rank = cand_rank(P, lp, n, mine.pad);
        }
    }
    if constexpr (FRESH) {
        if (live) {                                      // the WHOLE state, as hs_station_reset + the fold would have left it
            const bool pd = count && pend;
            uint32_t tot_ev = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { X.ev_kind[(size_t)k * n + lp] = ev[k]; tot_ev += ev[k]; }
#pragma unroll
            for (int k = 8; k < 11; ++k) X.ev_kind[(size_t)k * n + lp] = 0;
            X.events[lp] = tot_ev;
            X.generated[lp] = c_tick; X.accepted[lp] = acc2; X.dropped[lp] = 0; X.completed[lp] = c_dep; X.rejected[lp] = 0;
            X.started[lp] = st2; X.received[lp] = c_dep; X.sink_w[lp] = c_dep;
            X.buf[lp] = (int64_t)c_tick - (int64_t)c_start;
            X.active[lp] = pd ? 1 : 0;
            X.D[lp] = pd ? pendD : kInfNs; X.crtD[lp] = pd ? pendS : start_ns; X.svc_s[lp] = pd ? pend_s : 0.0; X.crt[lp] = 0;
            X.total_service[lp] = count ? total_service : 0.0;
            X.A[lp] = A_next_i; X.arr_time[lp] = A_next_i; X.arr_k[lp] = 1u + (uint64_t)(count ? n_arr_total : 0); X.svc_k[lp] = (uint64_t)c_start;
            X.crtA[lp] = crtA2;
            X.seqA[lp] = seqA; X.seqD[lp] = seqD; X.seq[lp] = seq;
            X.q[lp] = 0; X.grp_time[lp] = start_ns;
            X.last_time[lp] = count ? lt : start_ns;
            X.dpA[lp] = (uint8_t)dpA; X.rcA[lp] = rcA; X.dpD[lp] = (uint8_t)dpD; X.rcD[lp] = rcD; X.wkD[lp] = 1;
        }
    } else if (count) {
        // (the read-modify-write counters: every load before the first store -- the pointers may alias as far as the compiler knows, and
        //  a load behind each store is a memory round trip each)
        const int64_t o_gen = X.generated[lp], o_comp = X.completed[lp], o_recv = X.received[lp], o_events = X.events[lp];
        int64_t o_ev[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o_ev[k] = X.ev_kind[(size_t)k * n + lp];
        X.generated[lp] = o_gen + c_tick; X.accepted[lp] = acc2; X.started[lp] = st2; X.completed[lp] = o_comp + c_dep;
        X.received[lp] = o_recv + c_dep; X.sink_w[lp] = I.sink_w + c_dep;
        X.buf[lp] = (int64_t)c_tick - (int64_t)c_start;
        X.active[lp] = pend ? 1 : 0;
        X.D[lp] = pend ? pendD : kInfNs;
        if (pend) { X.crtD[lp] = pendS; X.svc_s[lp] = pend_s; }
        X.total_service[lp] = total_service;
        X.A[lp] = A_next_i; X.arr_time[lp] = A_next_i; X.arr_k[lp] = I.ak0 + (uint64_t)n_arr_total; X.svc_k[lp] = I.sk0 + (uint64_t)c_start;
        X.crtA[lp] = crtA2;
        if ((c_tick | c_start) != 0u) { X.seqA[lp] = seqA; X.seqD[lp] = seqD; X.seq[lp] = seq; }
        X.last_time[lp] = lt;
        if (c_tick) { X.dpA[lp] = (uint8_t)dpA; X.rcA[lp] = rcA; }
        if (pend && pend_new) { X.dpD[lp] = (uint8_t)dpD; X.rcD[lp] = rcD; }
        uint32_t tot_ev = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { X.ev_kind[(size_t)k * n + lp] = o_ev[k] + ev[k]; tot_ev += ev[k]; }
        X.events[lp] = o_events + tot_ev;
    }
    if (bailed) {
        const unsigned pos = atomicAdd(&ctl->n_bail, 1u);
        bail[pos] = lp;
    }
    lt_out = count ? (long long)lt : INT64_MIN;
}

}  // namespace

#ifndef HS_WAVE_WPE
#define HS_WAVE_WPE 8                  // wavefronts per SIMD the register allocation aims at (<= 64 VGPRs: two workgroups of 16 per CU)
#endif
// FRESH: the engine was reset and nothing has run since -- the kernel performs the bootstrap itself (hs_station_reset: the first
// arrival of every Source from start_ns) instead of loading the state a reset kernel would have written, and STORES the whole
// per-LP state instead of folding deltas into it; hs_station_wide_finish sets the engine totals (round 5: one launch, ~45 scalar
// loads and ~12 read-modify-writes per LP less on the strong shard's step).
template <int NW, bool FRESH = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(HS_WAVE_WPE, HS_WAVE_WPE))) hs_station_wave(StationParams P, StationState X, RecordLogs L, Totals *tot, Candidate *cands,
                                                           WideCtl *ctl, int32_t *bail, WavePart *parts, int n, int64_t end_ns, int flags,
                                                           int64_t start_ns) {
    static_assert(NW == 4 || NW == 8 || NW == 16, "a 128-byte line of a record log holds 16 LPs");
    constexpr int R = kWaveR;
    __shared__ int64_t st_a[NW][kWaveRowPad], st_d[NW][kWaveRowPad];   // the step's admission / completion records, INT64_MIN = none
    // the step's service samples in completion order and how many of them count: double-buffered, wavefront 0 adds up step s - 1
    // while step s is computed (T)
    __shared__ double s_sv[2][NW][R + 1];
    __shared__ int s_cnt[2][NW];
    __shared__ double s_ts[NW];                                        // _total_service_time of the workgroup's LPs
    __shared__ int64_t s_base[NW][2];                                  // index of slot 0's record in adm / sink_t
    // what only the fold at the end reads (kept out of the loop's registers): the request still in service {D, S, arrival, arrival
    // before it, index, service time}, the last two ticks and the pending one, arrivals
    __shared__ double s_pend[NW][7];
    __shared__ double s_tick[NW][3];
    __shared__ Candidate wave_c[NW];
    __shared__ unsigned s_ev[NW][8];
    __shared__ long long s_lt[NW];
    __shared__ int s_ovf[NW];
    __shared__ WaveIn s_in[NW];
    __shared__ WaveOut s_out[NW];

#ifdef HS_WAVE_CYC   // scratch build (tools/wide_timing.py --wave-cycles): where a wavefront of workgroup 0 spends its cycles
    const unsigned long long cyc_k0 = __builtin_readcyclecounter();
    unsigned long long cyc_compute = 0, cyc_phase2 = 0, cyc_loop0 = 0, cyc_bar = 0;
#ifdef HS_WAVE_SPAN
    const unsigned long long span_k0 = wall_clock64();
#endif
#endif
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lp0 = blockIdx.x * NW;
    const int lp = lp0 + w;
    const bool live = lp < n;
    const long long cur = FRESH ? (long long)start_ns : tot->cur_time;
    const int64_t T = end_ns;
    const bool frozen = cur > end_ns;
    const double NEG = -__builtin_huge_val();

    // ---- the LPs' state: wavefront 0, a lane per LP (coalesced loads, ONE bootstrap per LP), handed over in LDS (WaveIn)
    if (w == 0 && lane < NW) {
        const int lpl = lp0 + lane;
        WaveIn I;
        I.A = kInfNs; I.crtA = 0; I.Dprev = INT64_MIN; I.Sprev = INT64_MIN; I.accepted = 0; I.started = 0; I.sink_w = 0; I.last_time = 0;
        I.rcA = 0; I.rcD = 0; I.ak0 = 0; I.sk0 = 0; I.total_service = 0.0; I.svc_s0 = 0.0; I.carry_fresh = 0.0;
        I.key0 = I.key1 = I.asid0 = I.asid1 = I.ssid0 = I.ssid1 = 0; I.seq = I.seqA = I.seqD = 0; I.dpA = I.dpD = 0;
        ConstDiv dr, dl;
        dr.init(1.0); dl.init(1.0);
        bool busy_l = false, elig_l = false;
        if (lpl < n) {
            const uint64_t seed = P.seed[lpl], base = P.stream_base[lpl];
            const double rate_in = P.src_rate[lpl], mean_in = P.svc_mean[lpl];
            I.key0 = (uint32_t)seed; I.key1 = (uint32_t)(seed >> 32);
            const uint64_t sa = stream_id(base, kStreamArrival), ss = stream_id(base, kStreamService);
            I.asid0 = (uint32_t)sa; I.asid1 = (uint32_t)(sa >> 32); I.ssid0 = (uint32_t)ss; I.ssid1 = (uint32_t)(ss >> 32);
            dr.init(rate_in);
            dl.init(__ddiv_rn(1.0, mean_in));
            if constexpr (FRESH) {   // Simulation.__init__ bootstrap (core/simulation.py:145-154, hs_station_reset): the first arrival from start_ns
                const U4 o = philox4x32_10(0u, 0u, I.asid0, I.asid1, I.key0, I.key1);
                I.A = ns_from_seconds(__dadd_rn(seconds_from_ns(start_ns), __ddiv_rn(exp1_from_uniform(res53(o.x, o.y)), rate_in)));
                I.carry_fresh = dr.div(exp1_from_uniform(res53(o.z, o.w)));      // draw 1: the increment behind the first arrival
                I.crtA = start_ns; I.ak0 = 1; I.sk0 = 0; I.last_time = start_ns;
                I.seq = 1; I.rcA = INT64_MIN; I.rcD = INT64_MIN;
                elig_l = (int)!frozen & (int)(I.A >= 0) & (int)(start_ns >= 0) & (int)(end_ns < (1ll << 51)) & (int)(I.A < (1ll << 51));
            } else {
                // every load first, unconditionally (a short-circuited `&&` chain loads one value per memory round trip), then the predicates
                I.A = X.A[lpl]; I.crtA = X.crtA[lpl]; I.ak0 = X.arr_k[lpl]; I.sk0 = X.svc_k[lpl];
                I.accepted = X.accepted[lpl]; I.started = X.started[lpl]; I.sink_w = X.sink_w[lpl]; I.last_time = X.last_time[lpl];
                I.total_service = X.total_service[lpl];
                const int32_t active_in = X.active[lpl];
                const int64_t D_in = X.D[lpl], crtD_in = X.crtD[lpl], buf_in = X.buf[lpl], arr_time_in = X.arr_time[lpl];
                const double svc_s_in = X.svc_s[lpl];
                const uint32_t q_in = X.q[lpl];
                I.seq = X.seq[lpl]; I.seqA = X.seqA[lpl]; I.seqD = X.seqD[lpl];              // (only the fold reads these)
                I.dpA = X.dpA[lpl]; I.dpD = X.dpD[lpl]; I.rcA = X.rcA[lpl]; I.rcD = X.rcD[lpl];
                busy_l = active_in > 0;
                if (busy_l) { I.Dprev = D_in; I.Sprev = crtD_in; I.svc_s0 = svc_s_in; }
                elig_l = (int)!frozen & (int)(q_in == 0) & (int)(buf_in == 0) & (int)(active_in <= 1) & (int)(arr_time_in == I.A) &
                         (int)(I.A >= 0) & (int)(I.crtA >= 0) & (int)(I.last_time >= 0) & (int)(end_ns < (1ll << 51)) &
                         (int)(I.A == kInfNs || I.A < (1ll << 51)) &                        // (exact in binary64)
                         (int)(active_in == 0 || (D_in < (1ll << 51) && crtD_in >= 0));
            }
            if ((flags & (1 << 21)) && (lpl % 97) == 5) elig_l = false;     // debug: force some LPs through the bail path
        }
        I.rate_b = dr.b; I.rate_y = dr.y; I.lam_b = dl.b; I.lam_y = dl.y;
        I.bits = (busy_l ? 1 : 0) | (elig_l ? 2 : 0) | (dr.fast ? 4 : 0) | (dl.fast ? 8 : 0);
        s_in[lane] = I;
    }
    __syncthreads();
    // the LP's wavefront: everything wave-uniform (SGPRs)
    const int32_t in_bits = uni(s_in[w].bits);
    const int64_t A = uni(s_in[w].A), crtA = uni(s_in[w].crtA), accepted = uni(s_in[w].accepted), sink_w = uni(s_in[w].sink_w);
    const int64_t last_time = uni(s_in[w].last_time);
    const uint64_t ak0 = uni(s_in[w].ak0), sk0 = uni(s_in[w].sk0);
    const bool busy = (in_bits & 1) != 0, elig = (in_bits & 2) != 0;
    const int64_t Dprev = uni(s_in[w].Dprev), Sprev = uni(s_in[w].Sprev);
    double total_service = uni(s_in[w].total_service);
    const double svc_s0 = uni(s_in[w].svc_s0);
    uint32_t key0 = uni(s_in[w].key0), key1 = uni(s_in[w].key1);
    const uint32_t asid0 = uni(s_in[w].asid0), asid1 = uni(s_in[w].asid1), ssid0 = uni(s_in[w].ssid0), ssid1 = uni(s_in[w].ssid1);
    ConstDiv div_rate, div_lambda;
    div_rate.b = uni(s_in[w].rate_b); div_rate.y = uni(s_in[w].rate_y); div_rate.fast = (in_bits & 4) != 0;
    div_lambda.b = uni(s_in[w].lam_b); div_lambda.y = uni(s_in[w].lam_y); div_lambda.fast = (in_bits & 8) != 0;
    const double carry_fresh = FRESH ? uni(s_in[w].carry_fresh) : 0.0;
    uint32_t n_dep = 0, n_tick = 0, n_notify = 0, n_poll = 0, n_start = 0;   // wave-uniform counts (ballots)
    int64_t lt = last_time;
    int overflow = 0;
    bool bailed = live && !frozen && !elig;
    const bool run = live && elig;
    if (run && busy && Dprev <= T) {                    // the request already in service departs inside the window (Station::req_begin)
        total_service = __dadd_rn(total_service, svc_s0);
        if (lane == 0) { if (sink_w < L.cap) L.sink_t[(size_t)sink_w * n + lp] = Dprev; else overflow = 1; }
        n_dep = 1;
        lt = Dprev > lt ? Dprev : lt;
    }
    const uint32_t dep0 = n_dep;
    bool pend = run && busy && dep0 == 0;               // a request in service beyond the window: nobody else starts
    bool pend_new = false;                              // ... one that started in this window (s_pend)

    auto sec_d = [](double ns) {                        // to_seconds: float(ns) / 1e9, correctly rounded (hs_device.hpp seconds_from_ns_d)
        const double q0 = __dmul_rn(ns, 1e-9);
        const double r0_ = __fma_rn(-1e9, q0, ns);
        const double q1 = __fma_rn(r0_, 1e-9, q0);
        const double r1 = __fma_rn(-1e9, q1, ns);
        return __fma_rn(r1, 1e-9, q1);
    };
    auto nsd = [](double x) { return __builtin_trunc(__dmul_rn(x, 1e9)); };      // from_seconds, as a binary64 integer
    auto to_i64 = [](double d) {                        // exact for whole d in [0, 2^52)
        return (int64_t)((uint64_t)__double_as_longlong(__dadd_rn(d, 4503599627370496.0)) & 0xFFFFFFFFFFFFFull);
    };

    // ---- how the window's requests map onto Philox blocks.  Slot p = 2 lane + q of step s is request i = 128 s + p of the window; it
    // takes arrival draw ak0 + i (the increment from its own arrival to the next one) and service draw sk0 + i.  A stream whose
    // first draw k0 is even aligns its blocks with the lanes (block k0 / 2 + 64 s + lane = the lane's two slots); an odd k0 puts the
    // FIRST half of block (k0 + 1) / 2 + 64 s + lane into the lane's slot 1 and its second half into the NEXT lane's slot 0 (lane
    // 0: the second half of the block before it, carried from the step before -- at the start, the rest of the block draw k0 - 1
    // was taken from).
    const bool da = (ak0 & 1ull) != 0, ds = (sk0 & 1ull) != 0;
    uint64_t blk_a = ((ak0 + 1) >> 1) + (uint64_t)lane, blk_s = ((sk0 + 1) >> 1) + (uint64_t)lane;
    double carry_inc = 0.0, carry_sv = 0.0;
    if constexpr (FRESH) carry_inc = carry_fresh;
    if (!FRESH && run && da) {
        const uint64_t b = ak0 >> 1;
        const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), asid0, asid1, key0, key1);
        carry_inc = div_rate.div(exp1_from_uniform(res53(o.z, o.w)));
    }
    if (run && ds) {
        const uint64_t b = sk0 >> 1;
        const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), ssid0, ssid1, key0, key1);
        carry_sv = sec_d(nsd(div_lambda.div(exp1_from_uniform(res53(o.z, o.w)))));
    }
    const bool spec = end_ns < (1ll << 39);              // speculated whole-ns steps: a + F < 2^40 (below: F < 2^39)
    const double Td = (double)T;
    double base_a = (double)A;                           // arrival time of the step's slot 0
    double carryD = busy ? (double)Dprev : NEG, carryS = busy ? (double)Sprev : NEG;   // D and S of the request before the step's first
    double carryA = NEG;                                 // its arrival (NEG: none in this window)
    double lt_d = (double)lt;
    bool fin = !run || A > T;
    int64_t r0 = 0;                                      // request index of the step's slot 0 = arrivals so far
    int64_t steps_left = L.cap / R + 2;                  // (more steps than the record log has room for: an overflow, never a hang)
    if (lane == 0) {
        s_ts[w] = total_service;
        s_tick[w][0] = (double)A; s_tick[w][1] = (double)crtA; s_tick[w][2] = (double)crtA;   // pending tick and the two ticks before it
    }
    __syncthreads();
    double ts_mine = (w == 0 && lane < NW) ? s_ts[lane] : 0.0;   // wavefront 0, lane l: _total_service_time of LP lp0 + l
    bool rows_clear = false;                             // this wavefront's staging rows hold no record

#ifdef HS_WAVE_CYC
    cyc_loop0 = __builtin_readcyclecounter();
#endif
    auto service_sum = [&](int b) {                      // T of the step whose samples are in buffer b: wavefront 0, a lane per LP
        if (w == 0 && lane < NW) {
            const int c = s_cnt[b][lane];
            for (int k = 0; k < R && __any(k < c); k += 8) {
                if (k < c) {                             // (the writer zeroed the slots behind the last departure: + 0.0 is exact, the
                    double v[8];                         // chain is the additions alone -- no per-sample select)
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = s_sv[b][lane][k + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) ts_mine = __dadd_rn(ts_mine, v[q]);
                }
            }
        }
    };
    int step = 0;
    for (; __syncthreads_or(!fin); ++step) {
        int cnt_sv = 0;
        bool did = false;
        const int sb = step & 1;
        if (step > 0) service_sum(sb ^ 1);
#ifdef HS_WAVE_CYC
        const unsigned long long c0_ = __builtin_readcyclecounter();
#endif
        if (!fin) {
            if (--steps_left < 0) { overflow = 1; fin = true; }
            // ---- V: the step's stream values.  (The empty asm keeps the ten round keys from being hoisted out of the loop: twenty more
            // live SGPRs spill to VGPR lanes and every round then pays a v_readlane; twenty s_add per step are free next to the VALU.)
            key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key0); key1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key1);   // (wave-uniform by construction: say so)
            asm volatile("" : "+s"(key0), "+s"(key1));
            const U4 oa = philox4x32_10((uint32_t)blk_a, (uint32_t)(blk_a >> 32), asid0, asid1, key0, key1);
            const U4 os = philox4x32_10((uint32_t)blk_s, (uint32_t)(blk_s >> 32), ssid0, ssid1, key0, key1);
            blk_a += 64; blk_s += 64;
            double inc0 = div_rate.div(exp1_from_uniform(res53(oa.x, oa.y))), inc1 = div_rate.div(exp1_from_uniform(res53(oa.z, oa.w)));
            if (da) {
                const double up = dpp_orv<0x138, 0xf>(inc1, carry_inc);
                carry_inc = rl64(inc1, 63);
                inc1 = inc0; inc0 = up;
            }
            const double e0 = div_lambda.div(exp1_from_uniform(res53(os.x, os.y))), e1 = div_lambda.div(exp1_from_uniform(res53(os.z, os.w)));
            double sv0 = sec_d(nsd(e0)), sv1 = sec_d(nsd(e1));                     // Duration.from_seconds(sample).to_seconds()
            if (ds) {
                const double up = dpp_orv<0x138, 0xf>(sv1, carry_sv);
                carry_sv = rl64(sv1, 63);
                sv1 = sv0; sv0 = up;
            }
            const double du0 = nsd(sv0), du1 = nsd(sv1);
            // ---- C: arrival times.  a0 / a1 = arrival of slots 2 lane / 2 lane + 1, an = the tick after slot 2 lane + 1
            double a0, a1, an;
            {
                const double F0 = __dmul_rn(inc0, 1e9), F1 = __dmul_rn(inc1, 1e9);
                const double fl0 = __builtin_floor(F0), fl1 = __builtin_floor(F1);
                const double fr0 = __dsub_rn(F0, fl0), fr1 = __dsub_rn(F1, fl1);
                constexpr double m = 1.0 / 1024.0;
                const bool safe0 = spec && fr0 >= m && fr0 <= 1.0 - m && F0 < 549755813888.0;
                const bool safe1 = spec && fr1 >= m && fr1 <= 1.0 - m && F1 < 549755813888.0;
                const double f0 = safe0 ? fl0 : 0.0, f1 = safe1 ? fl1 : 0.0;
                const double ps = f0 + f1;
                double I = ps;                                                   // inclusive scan over the lanes (whole numbers: exact)
                I += dpp_or0<0x111, 0xf>(I); I += dpp_or0<0x112, 0xf>(I); I += dpp_or0<0x114, 0xf>(I); I += dpp_or0<0x118, 0xf>(I);
                I += dpp_or0<0x142, 0xa>(I);                                     // rows 1 and 3 take the row before them
                I += dpp_or0<0x143, 0xc>(I);                                     // rows 2 and 3 take rows 0-1
                a0 = base_a + (I - ps); a1 = a0 + f0; an = base_a + I;
                unsigned long long ub0 = __ballot(!safe0), ub1 = __ballot(!safe1);
                while ((ub0 | ub1) != 0ull) {                                    // the increments too close to a whole number, in order
                    const int l0 = ub0 ? (int)__builtin_ctzll(ub0) : 64, l1 = ub1 ? (int)__builtin_ctzll(ub1) : 64;
                    const bool at0 = l0 <= l1;                                   // position 2 l0 before position 2 l1 + 1
                    const int lu = at0 ? l0 : l1;
                    const double au = at0 ? rl64(a0, lu) : rl64(a1, lu), iu = at0 ? rl64(inc0, lu) : rl64(inc1, lu);
                    const double dl = nsd(__dadd_rn(sec_d(au), iu)) - au;        // the reference's step, exactly
                    if (at0) { if (lane >= lu) { a1 += dl; an += dl; } if (lane > lu) a0 += dl; ub0 &= ub0 - 1; }
                    else { if (lane >= lu) an += dl; if (lane > lu) { a0 += dl; a1 += dl; } ub1 &= ub1 - 1; }
                }
            }
            const double base_next = rl64(an, 63);
            // ---- L: Lindley recursion as a (max, +) scan (hs_kernels_wide.hpp MaxPlus), two requests per lane
            MaxPlus F = mp_after(MaxPlus{a1 + du1, du1}, MaxPlus{a0 + du0, du0});
            F = mp_scan_step<0x111, 0xf>(F); F = mp_scan_step<0x112, 0xf>(F); F = mp_scan_step<0x114, 0xf>(F); F = mp_scan_step<0x118, 0xf>(F);
            F = mp_scan_step<0x142, 0xa>(F);                                     // rows 1 and 3 take the row before them
            F = mp_scan_step<0x143, 0xc>(F);                                     // rows 2 and 3 take rows 0-1
            const MaxPlus E{dpp_orneg<0x138, 0xf>(F.p), dpp_or0<0x138, 0xf>(F.q)};   // composition of the lanes before this one (lane 0: the identity)
            const double Dp_in = __builtin_fmax(E.p, carryD + E.q);              // D of the request before this lane's first (-inf + q = -inf)
            const double S0 = __builtin_fmax(a0, Dp_in), D0 = S0 + du0;
            const double S1 = __builtin_fmax(a1, D0), D1 = S1 + du1;
            const double Sp_in = dpp_orv<0x138, 0xf>(S1, carryS);                // S of the request before this lane's first
            // which reference events happen (Station::req_step), request by request
            const bool arr0 = a0 <= Td, arr1 = a1 <= Td;
            const bool st0 = arr0 && S0 <= Td, st1 = arr1 && S1 <= Td;
            const bool dp0 = st0 && D0 <= Td, dp1 = st1 && D1 <= Td;
            const bool hz0 = (arr0 && (a0 == Sp_in || a0 == Dp_in || a1 <= a0)) || (st0 && du0 == 0.0);
            const bool hz1 = (arr1 && (a1 == S0 || a1 == D0 || an <= a1)) || (st1 && du1 == 0.0);
            if (__ballot(hz0 || hz1) != 0ull) { bailed = true; fin = true; }
            if (!bailed) {
                const unsigned long long b_arr0 = __ballot(arr0), b_arr1 = __ballot(arr1);
                const unsigned long long b_st0 = __ballot(st0), b_st1 = __ballot(st1), b_dp0 = __ballot(dp0), b_dp1 = __ballot(dp1);
                const int n_arr_l = __popcll(b_arr0) + __popcll(b_arr1);
                n_tick += (uint32_t)n_arr_l;
                n_notify += (uint32_t)(__popcll(__ballot(arr0 && Sp_in < a0)) + __popcll(__ballot(arr1 && S0 < a1)));
                n_poll += (uint32_t)(__popcll(__ballot(arr0 && Dp_in < a0)) + __popcll(__ballot(arr1 && D0 < a1)));
                n_start += (uint32_t)(__popcll(b_st0) + __popcll(b_st1));
                n_dep += (uint32_t)(__popcll(b_dp0) + __popcll(b_dp1));
                auto slot_of = [&](const double &x0, const double &x1, int p) { return (p & 1) ? rl64(x1, p >> 1) : rl64(x0, p >> 1); };
                // the latest processed event: arrivals, starts and departures are each in request order, and each a prefix of the step
                if (n_arr_l > 0) { const double v = slot_of(a0, a1, n_arr_l - 1); lt_d = v > lt_d ? v : lt_d; }
                const int n_st_l = __popcll(b_st0) + __popcll(b_st1), n_dp_l = __popcll(b_dp0) + __popcll(b_dp1);
                if (n_st_l > 0) { const double v = slot_of(S0, S1, n_st_l - 1); lt_d = v > lt_d ? v : lt_d; }
                if (n_dp_l > 0) { const double v = slot_of(D0, D1, n_dp_l - 1); lt_d = v > lt_d ? v : lt_d; }
                // the one request (at most) that started and is still in service at the end of the window: the last one that started
                if (n_st_l > n_dp_l) {
                    const int pp = n_st_l - 1;
                    pend_new = true;
                    if (lane == 0) {
                        s_pend[w][0] = slot_of(D0, D1, pp); s_pend[w][1] = slot_of(S0, S1, pp); s_pend[w][2] = slot_of(a0, a1, pp);
                        s_pend[w][3] = pp >= 1 ? slot_of(a0, a1, pp - 1) : carryA;
                        s_pend[w][4] = (double)(r0 + pp); s_pend[w][5] = slot_of(sv0, sv1, pp);
                        s_pend[w][6] = pp >= 1 ? slot_of(S0, S1, pp - 1) : carryS;  // S of the request before it
                    }
                }
                // ---- the step's records -> LDS (coalesced transposed write below); service samples for T
                st_a[w][2 * lane] = arr0 ? to_i64(a0) : INT64_MIN; st_a[w][2 * lane + 1] = arr1 ? to_i64(a1) : INT64_MIN;
                st_d[w][2 * lane] = dp0 ? to_i64(D0) : INT64_MIN; st_d[w][2 * lane + 1] = dp1 ? to_i64(D1) : INT64_MIN;
                if (n_dp_l > 0) { s_sv[sb][w][2 * lane] = dp0 ? sv0 : 0.0; s_sv[sb][w][2 * lane + 1] = dp1 ? sv1 : 0.0; }
                cnt_sv = n_dp_l;
                did = true; rows_clear = false;
                if (lane == 0) {
                    s_base[w][0] = accepted + r0; s_base[w][1] = sink_w + (int64_t)dep0 + r0;
                    // the pending tick and the two ticks before it (lineage, Station::req_finish)
                    if (n_arr_l > 0) {
                        const double last = slot_of(a0, a1, n_arr_l - 1);
                        s_tick[w][2] = n_arr_l >= 2 ? slot_of(a0, a1, n_arr_l - 2) : s_tick[w][1];
                        s_tick[w][1] = last;
                    }
                    s_tick[w][0] = n_arr_l < R ? slot_of(a0, a1, n_arr_l) : base_next;
                }
                if (n_arr_l < R) fin = true;
                r0 += n_arr_l;
                carryD = rl64(D1, 63); carryS = rl64(S1, 63); carryA = rl64(a1, 63);
                base_a = base_next;
            }
        }
        if (!did && !rows_clear) {                        // nothing to write for this LP in this step (done, or bailed): clear its rows once
            st_a[w][2 * lane] = INT64_MIN; st_a[w][2 * lane + 1] = INT64_MIN; st_d[w][2 * lane] = INT64_MIN; st_d[w][2 * lane + 1] = INT64_MIN;
            rows_clear = true;
        }
        if (lane == 0) s_cnt[sb][w] = (flags & (1 << 18)) ? 0 : cnt_sv;
#ifdef HS_WAVE_CYC
        const unsigned long long c1_ = __builtin_readcyclecounter();
        cyc_compute += c1_ - c0_;
#endif
        // ---- the workgroup writes the step's records: 16 neighbouring LPs' k-th records are one line
        __syncthreads();
#ifdef HS_WAVE_CYC
        const unsigned long long c2_ = __builtin_readcyclecounter();
        cyc_bar += c2_ - c1_;
#endif
        if (!(flags & (1 << 19))) {
            // (the thread's slot / LP / addresses are recomputed every step: hoisted out of the loop they spill to scratch under the
            //  64-VGPR budget, and every reload waits for vmcnt(0), i.e. for the stores before it -- 5 000 cycles per step, measured)
            int tx = (int)threadIdx.x;
            asm volatile("" : "+v"(tx));
            const int ww = tx % NW, s0 = tx / NW;
            const int64_t ba = s_base[ww][0], bd = s_base[ww][1];
            int64_t *const pa = L.adm + (lp0 + ww), *const pd = L.sink_t + (lp0 + ww);
#pragma unroll
            for (int it = 0; it < R / 64; ++it) {
                const int slot = s0 + it * 64;
                const int64_t va = st_a[ww][slot], vd = st_d[ww][slot];
                const int64_t ka = ba + slot, kd = bd + slot;
                if (va != INT64_MIN) { if (ka < L.cap) pa[(size_t)ka * n] = va; else overflow = 1; }
                if (vd != INT64_MIN) { if (kd < L.cap) pd[(size_t)kd * n] = vd; else overflow = 1; }
            }
        }
#ifdef HS_WAVE_CYC
        cyc_phase2 += __builtin_readcyclecounter() - c2_;
#endif
    }
#ifdef HS_WAVE_CYC
    const unsigned long long cyc_loop1 = __builtin_readcyclecounter();
#endif
    if (step > 0) service_sum((step - 1) & 1);           // T of the last step (wavefront 0 holds its LPs' sums in its lanes)
    // ---- fold the window into the LPs' state (Station::req_finish): wavefront 0 again, a lane per LP (wave_fold)
    {
        const int ovf_w = __any(overflow) ? 1 : 0;
        if (lane == 0) {
            WaveOut O;
            O.r0 = r0; O.lt = to_i64(lt_d);
            O.n_tick = n_tick; O.n_start = n_start; O.n_dep = n_dep; O.n_notify = n_notify; O.n_poll = n_poll;
            O.bits = (pend ? 1 : 0) | (pend_new ? 2 : 0) | (run && !bailed ? 4 : 0) | (bailed ? 8 : 0) | (ovf_w ? 16 : 0) | (live ? 32 : 0);
            s_out[w] = O;
        }
    }
    __syncthreads();
    if (w == 0 && lane < NW) {
        unsigned ev[8];
        long long lt_l;
        Candidate mine;
        wave_fold<FRESH>(P, X, L, ctl, bail, n, lp0 + lane, start_ns, s_in[lane], s_pend[lane], s_tick[lane], s_out[lane], ts_mine, ev, lt_l, mine);
#pragma unroll
        for (int k = 0; k < 8; ++k) s_ev[lane][k] = ev[k];
        s_lt[lane] = lt_l;
        s_ovf[lane] = (s_out[lane].bits & 16) ? 1 : 0;
        wave_c[lane] = mine;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        unsigned long long s = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) s += s_ev[q][threadIdx.x];
        parts[blockIdx.x].ev[threadIdx.x] = s;
    }
    if (threadIdx.x == 8) {
        long long mx = INT64_MIN; int o = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) { mx = s_lt[q] > mx ? s_lt[q] : mx; o |= s_ovf[q]; }
        parts[blockIdx.x].lt = mx; parts[blockIdx.x].ovf = o;
    }
    if (threadIdx.x == 64) {
        Candidate b = wave_c[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) if (cand_less(wave_c[q], b)) b = wave_c[q];
        cands[blockIdx.x] = b;
    }
#ifdef HS_WAVE_CYC
    if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == 1)) {     // dbg[0..3]: wavefront 1 {compute, barrier wait, T + writes, before + after the loop}
        const unsigned long long end_ = __builtin_readcyclecounter();
#ifndef HS_WAVE_SPAN
        if (w == 1) { tot->dbg[0] = cyc_compute; tot->dbg[1] = cyc_bar; tot->dbg[2] = cyc_phase2; tot->dbg[3] = (cyc_loop0 - cyc_k0) * 1000000ull + (end_ - cyc_loop1); }
#endif
    }
#ifdef HS_WAVE_SPAN   // (with HS_WAVE_CYC) dbg = {~earliest start, latest start, ~earliest end, latest end} over the workgroups (s_memtime)
    if (lane == 0 && w == 0) {                               // (s_memrealtime, 100 MHz: s_memtime is not synchronised between XCDs)
        const unsigned long long end_ = wall_clock64();
        atomicMax(&tot->dbg[0], ~span_k0); atomicMax(&tot->dbg[1], span_k0); atomicMax(&tot->dbg[2], ~end_); atomicMax(&tot->dbg[3], end_);
    }
#endif
#endif
}
