// hs_kernels_wide.hpp -- the headline grid with K LANES PER LP: strong scaling of `_execute_until` (core/simulation.py:449-505).
//
// hs_station_run gives every LP one lane.  That is the right shape while there are at least as many LPs as lanes (65 536 chains
// on one MI355X: one wavefront per SIMD); when the metric's 65 536 servers are block-partitioned over 8 GPUs a
// GPU holds 8 192 LPs, 7 of 8 lanes idle, and the slowest lane needs the same ~0.4 ms: no speed-up.  Here K lanes share one LP
// (`Source.poisson -> Server(Exp, c = 1, unbounded FIFO) -> Sink`, the host checks it: hs_engine uni_grid) and a wavefront works
// on 2K requests of each of its 64 / K LPs per step:
//   V  the stream values of the step -- E / rate per arrival draw, service time per service draw (Philox4x32-10, hs_log, the
//      constant-divisor quotients, the ns truncations: pure functions of the draw index) -- one Philox block per stream and lane,
//      into per-LP LDS rings (production runs one step ahead, so the parity of the draw counters does not matter);
//   C  the arrival chain  a' = from_seconds(to_seconds(a) + E / rate)  (load/arrival_time_provider.py:72-82) is inherently serial
//      (a rounding and a truncation per step): every lane of the LP's group walks the step's 2K arrivals redundantly -- ~25
//      dependent instructions per request, the one part that does not shrink with K;
//   L  the Lindley recursion  D_k = max(a_k, D_{k-1}) + s_k  is a prefix scan in the (max, +) semiring: f_k(x) = max(p_k, x + q_k),
//      closed under composition -- each lane composes its two requests, a log2(K)-step scan over the group gives every request
//      its predecessor's departure, and the reference's events of request k (SourceEvent, Request@Server, QUEUE_NOTIFY iff the
//      buffer was empty, QUEUE_POLL iff the worker was idle, QUEUE_DELIVER + Request@worker at the start, ProcessContinuation +
//      Request@Sink + the completion's QUEUE_POLL at the departure) are counted by comparing a_k, S_k, D_k with end_ns exactly as
//      Station::req_step does (hs_station.hpp), the record logs are appended lane-parallel;
//   T  `_total_service_time += s` is a binary64 running sum in completion order (server/server.py:252-273): a second short
//      serial walk (one addition per completed request).
// Anything whose outcome depends on the order of two events on the SAME nanosecond (a_k == S_{k-1}, a_k == D_{k-1}, a next tick
// on / before a_k, a zero-length service) makes the LP BAIL: nothing of it is stored, it is put on a list, and
// hs_station_wide_finish re-runs it from its old state with the event-order loop of hs_station.hpp before it elects the one
// event beyond end_ns among all LPs (the last-block part of hs_station_run).  Results are bit-identical to hs_station_run
// (tests/test_gpu_wide.py: every state array, every log record, totals, the elected event).
#pragma once

namespace {

// f(x) = max(p, x + q); (f2 o f1)(x) = max(max(p2, p1 + q2), x + q1 + q2)
// (p, q: whole nanoseconds below 2^53 held exactly in binary64; -infinity = "no predecessor")
struct MaxPlus { double p, q; };
__device__ __forceinline__ MaxPlus mp_compose(const MaxPlus &f2, const MaxPlus &f1) {
    const double a = f1.p + f2.q;
    return MaxPlus{f2.p > a ? f2.p : a, f1.q + f2.q};
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
    int lo = (int)(unsigned)(v & 0xffffffffll), hi = (int)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((int64_t)hi << 32) | (unsigned)lo;
}

}  // namespace

#ifndef HS_WIDE_T_ROLE
#define HS_WIDE_T_ROLE 1          // which wavefront sums the service times: 1 the chain's, 0 the values', 3 one of its own
#endif
constexpr int kWideTRole = HS_WIDE_T_ROLE;
constexpr int kWideBlock = kWideTRole == 3 ? 256 : 192;    // one wavefront per role (values / arrival chain / the rest [/ service-time sum]), one barrier per step

struct WideCtl {                   // device memory: what hs_station_wide leaves for hs_station_wide_finish
    unsigned int n_bail;           // LPs that bailed (same-nanosecond hazards), listed in bail[]
    unsigned int pad;
};

struct WavePart {                  // what one workgroup of hs_station_wave (hs_kernels_wave.hpp) adds to the engine's totals
    unsigned long long ev[8];
    long long lt;                  // latest processed event (INT64_MIN: none)
    int ovf, pad;
};

template <int N>
__device__ __forceinline__ double dpp_shr(double v) {   // lane i <- lane i - N inside its row of 16 lanes (v_mov_b32_dpp row_shr: ~10 cycles;
    const long long b = __double_as_longlong(v);        // a ds_bpermute round trip is ~150); lanes without a source keep their value
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x110 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x110 + N, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

#ifndef HS_WIDE_Q
#define HS_WIDE_Q 2
#endif
#ifndef HS_WIDE_WPE
#define HS_WIDE_WPE (kWideBlock / 64)
#endif
template <int K, int Q = HS_WIDE_Q>
__global__ void __launch_bounds__(kWideBlock) __attribute__((amdgpu_waves_per_eu(HS_WIDE_WPE))) hs_station_wide(StationParams P, StationState X, RecordLogs L, Totals *tot,
                                                              Candidate *cands, WideCtl *ctl, int32_t *bail, int n,
                                                              int64_t end_ns, int flags) {
    static_assert(K == 4 || K == 8 || K == 16, "a group of lanes lives inside one DPP row of 16 lanes");
    constexpr int G = 64 / K;          // LPs per wavefront
    static_assert(Q == 2 || Q == 4, "requests per lane and step: one or two Philox blocks per stream");
    constexpr int R = Q * K;           // requests per LP and step (Q per lane)
    constexpr int RING = 8 * R;        // values buffered per stream and LP: the stages below run on five different steps
    __shared__ double s_inc[G][RING], s_svc[G][RING];
    __shared__ double s_a[2][G][R + 1];    // arrival times of a step (whole ns in binary64), double-buffered by step parity (+ the next step's first)
    __shared__ double s_carry[G][2];       // D and S of the last request of the step before
    __shared__ int s_ndp[2][G];            // completions of a step (for the service-time sum, one iteration later)
    __shared__ double s_ts[G];             // _total_service_time, handed from the summing wavefront to the storing one
    // (Round 3 experiments, all bit-identical, none faster at 8 192 LPs: the service-time sum on the values' wavefront 0.171 ms or on
    //  a fourth wavefront 0.198 (HS_WIDE_T_ROLE); four requests per lane and step 0.228 (HS_WIDE_Q); two wavefronts per EU instead of
    //  three 0.168; the roles decoupled by progress counters in LDS instead of the barrier per step 0.162 -- the roles take about the
    //  same time per step (s_memtime: values 2 200, chain + sum 1 800, Lindley 2 700 per step at K = 4), each is a chain of dependent
    //  instructions on a wavefront of its own, and VALU busy is 51 % per SIMD: the step is latency-bound three ways at once.)
    // Three ROLES, one wavefront each, on the same 64 / K LPs, a step apart (one workgroup barrier per iteration).  In iteration i:
    //   wavefront 0  V: the stream values of step i + 3;
    //   wavefront 1  C: the arrival chain of step i + 1, and T: the service-time sum of step i - 1;
    //   wavefront 2  L: Lindley recursion, event counting and the record logs of step i.
    // As ONE wavefront per group the parts ran one after the other -- ~9 000 cycles of mostly dependent latency per 2K requests at
    // K = 8 (s_memtime: chain 2 300, L 2 200, T 2 900, values 1 600) -- on three SIMDs they overlap.
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int g = lane / K, j = lane % K;
    const int lp = blockIdx.x * G + g;
    const bool live = lp < n;
    const long long cur = tot->cur_time;
    const int64_t T = end_ns;
    const bool frozen = cur > end_ns;
    const unsigned long long gmask = (K == 64 ? ~0ull : ((1ull << K) - 1ull)) << (g * K);   // this LP's lanes

    // ---- the LP's state (every lane of the group holds the same copy)
    int64_t A = kInfNs, crtA = 0, Dprev = INT64_MIN, Sprev = INT64_MIN, accepted = 0, started = 0, sink_w = 0, last_time = 0;
    uint64_t ak0 = 0, sk0 = 0;
    double total_service = 0.0, svc_s0 = 0.0;
    bool busy = false, elig = false;
    ConstDiv div_rate, div_lambda;
    uint32_t key0 = 0, key1 = 0, asid0 = 0, asid1 = 0, ssid0 = 0, ssid1 = 0;
    div_rate.init(1.0); div_lambda.init(1.0);
    if (live) {
        A = X.A[lp]; crtA = X.crtA[lp]; ak0 = X.arr_k[lp]; sk0 = X.svc_k[lp];
        accepted = X.accepted[lp]; started = X.started[lp]; sink_w = X.sink_w[lp]; last_time = X.last_time[lp];
        total_service = X.total_service[lp];
        busy = X.active[lp] > 0;
        if (busy) { Dprev = X.D[lp]; Sprev = X.crtD[lp]; svc_s0 = X.svc_s[lp]; }
        elig = !frozen && X.q[lp] == 0 && X.buf[lp] == 0 && X.active[lp] <= 1 && X.arr_time[lp] == A &&
               A >= 0 && crtA >= 0 && last_time >= 0 && end_ns < (1ll << 51) && (A == kInfNs || A < (1ll << 51)) &&   // (exact in binary64, to_i64)
               (X.active[lp] == 0 || (X.D[lp] < (1ll << 51) && X.crtD[lp] >= 0));
        const uint64_t seed = P.seed[lp], base = P.stream_base[lp];
        key0 = (uint32_t)seed; key1 = (uint32_t)(seed >> 32);
        const uint64_t sa = stream_id(base, kStreamArrival), ss = stream_id(base, kStreamService);
        asid0 = (uint32_t)sa; asid1 = (uint32_t)(sa >> 32); ssid0 = (uint32_t)ss; ssid1 = (uint32_t)(ss >> 32);
        div_rate.init(P.src_rate[lp]);
        div_lambda.init(__ddiv_rn(1.0, P.svc_mean[lp]));
    }
    if ((flags & (1 << 21)) && live && (lp % 97) == 5) elig = false;     // debug: force some LPs through the bail path
    const int64_t crtA0 = crtA;
    // the request already in service departs inside the window (Station::req_begin)
    uint32_t n_dep = 0, n_tick = 0, n_notify = 0, n_poll = 0, n_start = 0;
    int64_t lt = last_time;
    int overflow = 0;
    bool bailed = live && !frozen && !elig;
    const bool run = live && elig;
    if (run && busy && Dprev <= T) {
        total_service = __dadd_rn(total_service, svc_s0);
        if (j == 0 && role == 2) { if (sink_w < L.cap) L.sink_t[(size_t)sink_w * n + lp] = Dprev; else overflow = 1; }
        n_dep = 1;
        lt = Dprev > lt ? Dprev : lt;
    }
    const uint32_t dep0 = n_dep;
    bool pend = run && busy && dep0 == 0;               // a request in service beyond the window: nobody else starts
    int64_t pendD = Dprev, pendS = Sprev;
    double pend_s = svc_s0;
    bool pend_new = false;                              // ... one that started in this window

    // Times are whole nanoseconds below 2^52 (eligibility), so a binary64 holds them EXACTLY: from_seconds(x) = trunc(x * 1e9) is one
    // v_trunc_f64 instead of the multi-instruction f64 -> i64 conversion, to_seconds(ns) starts from the value itself instead of an
    // i64 -> f64 conversion, sums and maxima of such integers are exact -- the arrival chain is 8 dependent fp64 instructions per
    // request (45 with the integer round trip) and nothing converts until a value is stored.
    auto sec_d = [](double ns) {                        // to_seconds: float(ns) / 1e9, correctly rounded (hs_device.hpp seconds_from_ns)
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
    // ---- V: blocks (pa >> 1) + j, two values each
    uint64_t pa = ak0 & ~1ull, ps = sk0 & ~1ull;        // draws produced so far (block aligned)
    auto produce = [&]() {
        const uint64_t ba = (pa >> 1) + (uint64_t)j, bs = (ps >> 1) + (uint64_t)j;
        const U4 oa = philox4x32_10((uint32_t)ba, (uint32_t)(ba >> 32), asid0, asid1, key0, key1);
        const U4 os = philox4x32_10((uint32_t)bs, (uint32_t)(bs >> 32), ssid0, ssid1, key0, key1);
        const double a0 = div_rate.div(exp1_from_uniform(res53(oa.x, oa.y))), a1 = div_rate.div(exp1_from_uniform(res53(oa.z, oa.w)));
        const double e0 = div_lambda.div(exp1_from_uniform(res53(os.x, os.y))), e1 = div_lambda.div(exp1_from_uniform(res53(os.z, os.w)));
        const double v0 = sec_d(nsd(e0)), v1 = sec_d(nsd(e1));   // Duration.from_seconds(sample).to_seconds()
        s_inc[g][(int)((2 * ba) % RING)] = a0; s_inc[g][(int)((2 * ba + 1) % RING)] = a1;
        s_svc[g][(int)((2 * bs) % RING)] = v0; s_svc[g][(int)((2 * bs + 1) % RING)] = v1;
        pa += 2 * K; ps += 2 * K;
    };
    auto produce_step = [&]() { produce(); if constexpr (Q == 4) produce(); };   // the values of one step
    if (role == 0) { produce_step(); produce_step(); produce_step(); }
    const double NEG = -__builtin_huge_val();
    if (role == 2 && j == 0) {
        s_carry[g][0] = busy ? (double)Dprev : NEG; s_carry[g][1] = busy ? (double)Sprev : NEG;
        s_ndp[0][g] = 0; s_ndp[1][g] = 0;
    }
    __syncthreads();

    double a_cur = (double)A;                           // C: arrival time of the next request the chain reaches (draws ak0 + i, sk0 + i)
    int64_t r0 = 0;
    bool fin = !run || A > T;
    int64_t n_arr_total = 0;                            // arrivals processed (requests with a_i <= T)
    double A_next = (double)A, a_last = (double)crtA, a_last2 = (double)crtA;   // pending tick and the two ticks before it
    auto chain = [&](int buf, int64_t first) {          // the step that starts at request `first`, all lanes of the group redundantly
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const double inc = s_inc[g][(int)((ak0 + (uint64_t)(first + m)) % RING)];
            if (j == 0) s_a[buf][g][m] = a_cur;
            a_cur = nsd(__dadd_rn(sec_d(a_cur), inc));
        }
        if (j == 0) s_a[buf][g][R] = a_cur;
    };
    auto service_sum = [&](int n_add, int64_t first) {  // T: the first n_add requests of the step that starts at `first` completed
        constexpr int CH = R < 16 ? R : 16;               // (a chunk's loads first: one after the other each costs an LDS round trip)
#pragma unroll
        for (int c = 0; c < R; c += CH) {
            double sv[CH];
#pragma unroll
            for (int m = 0; m < CH; ++m) sv[m] = s_svc[g][(int)((sk0 + (uint64_t)(first + c + m)) % RING)];
#pragma unroll
            for (int m = 0; m < CH; ++m) total_service = (c + m) < n_add ? __dadd_rn(total_service, sv[m]) : total_service;
        }
    };
    if (role == 1) chain(0, 0);
    __syncthreads();
    int it = 0;
    const double Td = (double)T;
    double lt_d = (double)lt, pendD_d = 0.0, pendS_d = 0.0;
#ifdef HS_WIDE_CYC   // scratch build (tools/_wide_cyc.py): cycles each role works per step vs the loop's total
    unsigned long long cyc_work = 0;
    const unsigned long long cyc_t0 = __builtin_readcyclecounter();
#define HS_WCYC_BEGIN const unsigned long long wc0_ = __builtin_readcyclecounter();
#define HS_WCYC_END cyc_work += __builtin_readcyclecounter() - wc0_;
#else
#define HS_WCYC_BEGIN
#define HS_WCYC_END
#endif
    while (__syncthreads_or(role == 2 && !fin)) {
        const int cb = it & 1;
        HS_WCYC_BEGIN
        if (role == 0) {                                                         // V of step it + 3
            produce_step();
            if (kWideTRole == 0 && it > 0) service_sum(s_ndp[cb ^ 1][g], r0 - R);
            r0 += R; ++it; HS_WCYC_END continue;
        }
        if (role == 1) {
            chain(cb ^ 1, r0 + R);                                               // C of step it + 1 (speculative beyond the LP's last arrival: harmless)
            if (kWideTRole == 1 && it > 0) service_sum(s_ndp[cb ^ 1][g], r0 - R);   // T of step it - 1
            r0 += R; ++it;
            HS_WCYC_END
            continue;
        }
        if (role == 3) {
            if (it > 0) service_sum(s_ndp[cb ^ 1][g], r0 - R);
            r0 += R; ++it;
            HS_WCYC_END
            continue;
        }
        // ---- L: this lane's Q requests, in exact binary64 integer arithmetic
        const int m0 = Q * j;
        const double Dp_c = s_carry[g][0], Sp_c = s_carry[g][1];                 // D and S of the request before the step's first
        double av[Q + 1], sv_[Q], du[Q];
#pragma unroll
        for (int q = 0; q <= Q; ++q) av[q] = s_a[cb][g][m0 + q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { sv_[q] = s_svc[g][(int)((sk0 + (uint64_t)(r0 + m0 + q)) % RING)]; du[q] = nsd(sv_[q]); }
        MaxPlus F{av[0] + du[0], du[0]};
#pragma unroll
        for (int q = 1; q < Q; ++q) F = mp_compose(MaxPlus{av[q] + du[q], du[q]}, F);
        // inclusive scan over the group's lanes (Kogge-Stone with DPP row shifts), then shift by one for the predecessor's map
        if constexpr (K > 1) { MaxPlus Pm{dpp_shr<1>(F.p), dpp_shr<1>(F.q)}; if (j >= 1) F = mp_compose(F, Pm); }
        if constexpr (K > 2) { MaxPlus Pm{dpp_shr<2>(F.p), dpp_shr<2>(F.q)}; if (j >= 2) F = mp_compose(F, Pm); }
        if constexpr (K > 4) { MaxPlus Pm{dpp_shr<4>(F.p), dpp_shr<4>(F.q)}; if (j >= 4) F = mp_compose(F, Pm); }
        if constexpr (K > 8) { MaxPlus Pm{dpp_shr<8>(F.p), dpp_shr<8>(F.q)}; if (j >= 8) F = mp_compose(F, Pm); }
        const MaxPlus E{dpp_shr<1>(F.p), dpp_shr<1>(F.q)};                       // composition of the lanes before this one
        const double xq = Dp_c + E.q;                    // (-inf + q = -inf: nothing before the first request)
        double Dp = j == 0 ? Dp_c : (E.p > xq ? E.p : xq);                      // D of the request before this lane's first
        double Sv[Q], Dv[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { Sv[q] = av[q] > Dp ? av[q] : Dp; Dv[q] = Sv[q] + du[q]; Dp = Dv[q]; }
        const double S_before = dpp_shr<1>(Sv[Q - 1]);
        if (j == K - 1) { s_carry[g][0] = Dv[Q - 1]; s_carry[g][1] = Sv[Q - 1]; }   // the group's last request of the step carries over
        const bool act = !fin;
        // which reference events happen (Station::req_step), request by request
        bool hazard = false, arr_q[Q], dp_q[Q];
        double Dprv = j == 0 ? Dp_c : (E.p > xq ? E.p : xq), Sprv = j == 0 ? Sp_c : S_before, lt_l = NEG;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const double a = av[q];
            const bool arr = act && a <= Td;
            const bool st = arr && Sv[q] <= Td, dp = st && Dv[q] <= Td;
            hazard = hazard || (arr && (a == Sprv || a == Dprv || av[q + 1] <= a)) || (st && du[q] == 0.0);
            n_tick += arr ? 1u : 0u;
            n_notify += (arr && Sprv < a) ? 1u : 0u;
            n_poll += (arr && Dprv < a) ? 1u : 0u;
            n_start += st ? 1u : 0u;
            n_dep += dp ? 1u : 0u;
            // the latest processed event of this request (a <= S < D); a later request may have arrived before this one left
            { const double e = dp ? Dv[q] : st ? Sv[q] : arr ? a : NEG; lt_l = e > lt_l ? e : lt_l; }
            if (st && !dp) { pend_new = true; pendD_d = Dv[q]; pendS_d = Sv[q]; pend_s = sv_[q]; }
            arr_q[q] = arr; dp_q[q] = dp;
            Dprv = Dv[q]; Sprv = Sv[q];
        }
        lt_d = lt_l > lt_d ? lt_l : lt_d;
        // record logs: adm[k] = time of tick k (doubles as created_at of sink record k), sink_t[m] = completion time
        if (!(flags & (1 << 19))) {
            const int64_t w0 = accepted + r0 + m0, z0 = sink_w + (int64_t)dep0 + r0 + m0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (arr_q[q]) { if (w0 + q < L.cap) L.adm[(size_t)(w0 + q) * n + lp] = to_i64(av[q]); else overflow = 1; }
                if (dp_q[q]) { if (z0 + q < L.cap) L.sink_t[(size_t)(z0 + q) * n + lp] = to_i64(Dv[q]); else overflow = 1; }
            }
        }
        // the LP's step: arrivals, completions, hazards over the group -- wavefront ballots, no shuffles
        int n_arr_l = 0, n_dp_l = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { n_arr_l += __popcll(__ballot(arr_q[q]) & gmask); n_dp_l += __popcll(__ballot(dp_q[q]) & gmask); }
        const bool hz = (__ballot(hazard) & gmask) != 0ull;
        if (act && hz) { bailed = true; fin = true; }
        if (j == 0) s_ndp[cb][g] = (act && !bailed) ? n_dp_l : 0;               // (the summing wavefront reads it an iteration later)
        if (act && !bailed) {
            if (n_arr_l > 0) {                           // the pending tick and the two before it (lineage, Station::req_finish)
                const double last = s_a[cb][g][n_arr_l - 1];
                a_last2 = n_arr_l >= 2 ? s_a[cb][g][n_arr_l - 2] : a_last;
                a_last = last;
            }
            n_arr_total += n_arr_l;
            A_next = s_a[cb][g][n_arr_l];                // the first arrival beyond T when the step was not full
            if (n_arr_l < R) fin = true;
        }
        r0 += R;
        ++it;
        HS_WCYC_END
    }
#ifdef HS_WIDE_CYC
    if (lane == 0 && blockIdx.x == 0) {
        atomicAdd(&tot->dbg[role], cyc_work);
        if (role == 2) atomicAdd(&tot->dbg[3], __builtin_readcyclecounter() - cyc_t0);
    }
#endif
    if (role == kWideTRole) {                            // T of the last step, then hand the sum over
        if (it > 0) service_sum(s_ndp[(it & 1) ^ 1][g], r0 - R);
        if (j == 0) s_ts[g] = total_service;
    }
    __syncthreads();
    if (role != 2) return;                               // (the fold, the store and the block's candidate: the third wavefront)
    total_service = s_ts[g];

    // ---- fold the window into the LP's state (Station::req_finish), one lane per LP
    uint32_t c_tick = n_tick, c_notify = n_notify, c_poll = n_poll, c_start = n_start, c_dep = n_dep - dep0;
    int pn = pend_new ? 1 : 0, ovf = overflow;
#pragma unroll
    for (int o = 1; o < K; o <<= 1) {
        c_tick += __shfl_xor(c_tick, o, 64); c_notify += __shfl_xor(c_notify, o, 64); c_poll += __shfl_xor(c_poll, o, 64);
        c_start += __shfl_xor(c_start, o, 64); c_dep += __shfl_xor(c_dep, o, 64);
        const double olt = shfl_f64(lt_d, lane ^ o);
        lt_d = olt > lt_d ? olt : lt_d;
        ovf |= __shfl_xor(ovf, o, 64);
    }
    lt = (int64_t)lt_d;
    // the one request (at most) that started in the window and is still in service: find the lane that holds it
    {
        const unsigned long long holders = __ballot(pend_new) & (K == 64 ? ~0ull : (((1ull << K) - 1ull) << (g * K)));
        if (holders) {
            const int src = (int)__builtin_ctzll(holders);
            pendD = (int64_t)shfl_f64(pendD_d, src); pendS = (int64_t)shfl_f64(pendS_d, src);
            pend_s = shfl_f64(pend_s, src);
            pend = true;
        }
        pn = holders ? 1 : 0;
    }
    c_dep += dep0;
    Candidate mine = cand_none(lp);
    unsigned ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (run && !bailed && j == 0) {
        ev[0] = c_tick; ev[1] = c_tick; ev[2] = c_notify; ev[3] = c_poll + c_dep; ev[4] = c_start; ev[5] = c_start; ev[6] = c_dep; ev[7] = c_dep;
        const int64_t acc2 = accepted + c_tick, st2 = started + c_start;
        X.generated[lp] += c_tick; X.accepted[lp] = acc2; X.started[lp] = st2; X.completed[lp] += c_dep;
        X.received[lp] += c_dep; X.sink_w[lp] = sink_w + c_dep;
        X.buf[lp] = (int64_t)c_tick - (int64_t)c_start;
        X.active[lp] = pend ? 1 : 0;
        X.D[lp] = pend ? pendD : kInfNs;
        if (pend) { X.crtD[lp] = pendS; X.svc_s[lp] = pend_s; }
        X.total_service[lp] = total_service;
        const int64_t A_next_i = run && A != kInfNs ? (int64_t)A_next : A;
        X.A[lp] = A_next_i; X.arr_time[lp] = A_next_i; X.arr_k[lp] = ak0 + (uint64_t)n_arr_total; X.svc_k[lp] = sk0 + (uint64_t)c_start;
        const int64_t crtA2 = c_tick ? (int64_t)a_last : crtA0;
        X.crtA[lp] = crtA2;
        uint32_t seq = X.seq[lp];
        uint32_t seqA = X.seqA[lp], seqD = X.seqD[lp];
        if ((c_tick | c_start) != 0u) {                  // creation stamps: only their order matters (Station::req_finish)
            const bool d_first = pend && pendS < crtA2;
            seqA = seq + (d_first ? 1u : 0u); seqD = seq + (d_first ? 0u : 1u); seq += 2u;
            X.seqA[lp] = seqA; X.seqD[lp] = seqD; X.seq[lp] = seq;
        }
        X.last_time[lp] = lt;
        // lineage of what is pending now (Station::req_finish)
        int32_t dpA = X.dpA[lp], dpD = X.dpD[lp];
        int64_t rcA = X.rcA[lp], rcD = X.rcD[lp];
        if (c_tick) { dpA = 1; rcA = acc2 >= 2 ? (c_tick >= 2 ? (int64_t)a_last2 : L.adm[(size_t)(acc2 - 2) * n + lp]) : crtA0; X.dpA[lp] = (uint8_t)dpA; X.rcA[lp] = rcA; }
        if (pend && pn) {
            const int64_t m = st2 - 1;
            const int64_t a_m = m < L.cap ? L.adm[(size_t)m * n + lp] : 0;
            if (pendS == a_m) { dpD = 6; rcD = m >= 1 ? L.adm[(size_t)(m - 1) * n + lp] : crtA0; }
            else {
                Stream st;
                st.init(((uint64_t)key1 << 32) | key0, ((uint64_t)ssid1 << 32) | ssid0, (uint64_t)(m - 1));
                const double s_prev = seconds_from_ns(ns_from_seconds(div_lambda.div(exp1_from_uniform(st.next_uniform()))));
                dpD = 4; rcD = pendS - ns_from_seconds(s_prev);
            }
            X.dpD[lp] = (uint8_t)dpD; X.rcD[lp] = rcD;
        }
        uint32_t tot_ev = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { X.ev_kind[(size_t)k * n + lp] += ev[k]; tot_ev += ev[k]; }
        X.events[lp] += tot_ev;
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
    } else if (live && !bailed && j == 0 && !frozen) {
        // (not reachable: a live LP either runs or bails)
    }
    if (live && frozen && j == 0) { /* the run is over: nothing moves */ }
    if (bailed && j == 0) {
        const unsigned pos = atomicAdd(&ctl->n_bail, 1u);
        bail[pos] = lp;
    }
    // ---- totals and the block's candidate
    const bool count = run && !bailed && j == 0;
    unsigned vals[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) vals[k] = count ? ev[k] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned s = wave_sum<unsigned>(vals[k]);
        if (lane == 0 && s) {
            atomicAdd(&tot->ev[k], (unsigned long long)s);
            if (k == 6) atomicAdd(&tot->completed, (unsigned long long)s);
            if (k == 7) atomicAdd(&tot->received, (unsigned long long)s);
        }
    }
    {
        long long mx = count ? (long long)lt : INT64_MIN;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const long long d = shfl_xor_ll(mx, o); mx = d > mx ? d : mx; }
        if (lane == 0 && mx != INT64_MIN) atomicMax(&tot->final_time, mx);
        const int any_ovf = __any(count && ovf);
        if (lane == 0 && any_ovf) atomicOr(&tot->overflow, 1);
    }
    const Candidate w = wave_min_cand(mine);
    if (lane == 0) cands[blockIdx.x] = w;
}

#ifdef HS_KERNELS_MAIN
// After hs_station_wide: the LPs that bailed run in event order (the regular Station code, one lane each), then the one event
// beyond end_ns is elected among all candidates and processed -- the last-block part of hs_station_run (SINGLE mode).
__global__ void __launch_bounds__(kBlock) hs_station_wide_finish(StationParams P, StationState X, RecordLogs L, Totals *tot,
                                                                 Candidate *cands, int n_cands, WideCtl *ctl, const int32_t *bail,
                                                                 int n, int64_t end_ns, const WavePart *parts, long long fresh_start) {
    __shared__ uint8_t qmem[kQCap][kBlock];
    __shared__ double ring_a[kRing][kBlock], ring_s[kRing][kBlock];
    __shared__ Candidate wave_c[kBlock / 64];
    const int tid = threadIdx.x;
    // fresh_start != INT64_MIN: hs_station_wave<NW, true> ran the bootstrap itself (no hs_station_reset launch before it) -- the
    // engine totals start here, as the reset kernel leaves them
    const bool fresh = fresh_start != INT64_MIN;
    if (fresh) {
        if (tid == 0) {
            for (int k = 0; k < 15; ++k) tot->ev[k] = 0;
            tot->completed = 0; tot->received = 0; tot->final_time = fresh_start; tot->cur_time = fresh_start;
            tot->overflow = 0; tot->qoverflow = 0; tot->done = 0; tot->undecided = 0;
#ifndef HS_WAVE_CYC           // (the instrumented build keeps hs_station_wave's counters)
            tot->dbg[0] = tot->dbg[1] = tot->dbg[2] = tot->dbg[3] = 0;
#endif
            tot->not_done = 0;
        }
        __syncthreads();
    }
    const long long cur = fresh ? fresh_start : tot->cur_time;
    const unsigned nb = ctl->n_bail;
    if (parts != nullptr) {                                  // hs_station_wave: the workgroups' partial totals (one per candidate)
        unsigned long long s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long mx = INT64_MIN;
        int ov = 0;
        for (int b = tid; b < n_cands; b += kBlock) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += parts[b].ev[k];
            mx = parts[b].lt > mx ? parts[b].lt : mx;
            ov |= parts[b].ovf;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long v = wave_sum<unsigned long long>(s[k]);
            if ((tid & 63) == 0 && v) {
                atomicAdd(&tot->ev[k], v);
                if (k == 6) atomicAdd(&tot->completed, v);
                if (k == 7) atomicAdd(&tot->received, v);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const long long d = shfl_xor_ll(mx, o); mx = d > mx ? d : mx; }
        if ((tid & 63) == 0 && mx != INT64_MIN) atomicMax(&tot->final_time, mx);
        if (__any(ov) && (tid & 63) == 0) atomicOr(&tot->overflow, 1);
        __syncthreads();                                     // (thread 0 reads final_time below)
    }
    Candidate best = cand_none(0);
    for (unsigned b0 = 0; b0 < nb; b0 += kBlock) {
        const unsigned b = b0 + (unsigned)tid;
        const bool live = b < nb;
        Station<1, false, false> S;
        int lp = 0;
        if (live) {
            lp = bail[b];
            load_station<1, false, false>(S, P, X, L, lp, n, qmem, ring_a, ring_s, tid);
            S.force_general = false;
        }
        for (;;) {                                       // the event-order loop of hs_station_run
            const int64_t t = live && S.qn == 0 ? S.next_time() : kInfNs;
            const bool act = live && S.qn == 0 && t <= end_ns;
            if (!__any(act)) break;
            S.top_up(act);
            S.step_c1(t, act);
        }
        if (live) {
            Candidate c = make_candidate<1, false, false>(S);
            c.rank = cand_rank(P, lp, n, c.pad);
            store_station<1, false, false>(S, X, lp, n);
            for (int k = 0; k < 8; ++k) if (S.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)S.ev[k]);
            if (S.ev[6]) atomicAdd(&tot->completed, (unsigned long long)S.ev[6]);
            if (S.ev[7]) atomicAdd(&tot->received, (unsigned long long)S.ev[7]);
            atomicMax(&tot->final_time, (long long)S.last_time);
            if (S.overflow) atomicOr(&tot->overflow, 1);
            if (S.qoverflow) atomicOr(&tot->qoverflow, 1);
            if (cand_less(c, best)) best = c;
        }
        __syncthreads();
    }
    for (int b = tid; b < n_cands; b += kBlock) {
        const Candidate c = cand_load_agent(&cands[b]);
        if (cand_less(c, best)) best = c;
    }
    best = wave_min_cand(best);
    if ((tid & 63) == 0) wave_c[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        Candidate b = wave_c[0];
        for (int w = 1; w < kBlock / 64; ++w) if (cand_less(wave_c[w], b)) b = wave_c[w];
        long long new_cur = __hip_atomic_load(&tot->final_time, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur <= end_ns && b.valid) {
            Station<1, false> W;
            load_station<1, false>(W, P, X, L, b.lp, n, qmem, ring_a, ring_s, 0);
            W.force_general = false;
            overshoot_one<1, false>(W);
            store_station<1, false>(W, X, b.lp, n);
            for (int k = 0; k < 8; ++k) if (W.ev[k]) atomicAdd(&tot->ev[k], (unsigned long long)W.ev[k]);
            if (W.ev[6]) atomicAdd(&tot->completed, (unsigned long long)W.ev[6]);
            if (W.ev[7]) atomicAdd(&tot->received, (unsigned long long)W.ev[7]);
            if (W.overflow) atomicOr(&tot->overflow, 1);
            new_cur = b.t;
            atomicMax(&tot->final_time, new_cur);
        }
        if (cur <= end_ns) tot->cur_time = new_cur;
        ctl->n_bail = 0;
    }
}
#endif  // HS_KERNELS_MAIN
