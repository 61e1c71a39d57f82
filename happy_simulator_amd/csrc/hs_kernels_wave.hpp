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
//      request, samples broadcast from LDS -- the only serial part left.
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

}  // namespace

#ifndef HS_WAVE_WPE
#define HS_WAVE_WPE 4
#endif

template <int NW>
__global__ void __launch_bounds__(NW * 64) hs_station_wave(StationParams P, StationState X, RecordLogs L, Totals *tot, Candidate *cands,
                                                           WideCtl *ctl, int32_t *bail, WavePart *parts, int n, int64_t end_ns, int flags) {
    static_assert(NW == 4 || NW == 8 || NW == 16, "a 128-byte line of a record log holds 16 LPs");
    constexpr int R = kWaveR;
    __shared__ int64_t st_a[NW][kWaveRowPad], st_d[NW][kWaveRowPad];   // the step's admission / completion records, INT64_MIN = none
    __shared__ double s_sv[NW][R];                                     // the step's service samples in completion order (0.0: not completed)
    __shared__ int64_t s_base[NW][2];                                  // index of slot 0's record in adm / sink_t
    __shared__ Candidate wave_c[NW];
    __shared__ unsigned s_ev[NW][8];
    __shared__ long long s_lt[NW];
    __shared__ int s_ovf[NW];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lp0 = blockIdx.x * NW;
    const int lp = lp0 + w;
    const bool live = lp < n;
    const long long cur = tot->cur_time;
    const int64_t T = end_ns;
    const bool frozen = cur > end_ns;
    const double NEG = -__builtin_huge_val();

    // ---- the LP's state: wave-uniform (scalar loads)
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
        elig = !frozen && X.q[lp] == 0 && X.buf[lp] == 0 && X.active[lp] <= 1 && X.arr_time[lp] == A && ak0 >= 1 &&
               A >= 0 && crtA >= 0 && last_time >= 0 && end_ns < (1ll << 51) && (A == kInfNs || A < (1ll << 51)) &&   // (exact in binary64)
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
    int64_t pendD = Dprev, pendS = Sprev;
    double pend_s = svc_s0;
    bool pend_new = false;                              // ... one that started in this window:
    double pendA_d = 0.0, pendAp_d = 0.0;               //     its arrival and the arrival before it (lineage, Station::req_finish)
    int64_t pend_i = 0;                                 //     its index among the window's requests

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

    // ---- how the window's requests map onto Philox blocks.  Slot p = 2 lane + q of step s holds arrival draw pa + 128 s + p, whose
    // increment leads from request i = 128 s + p - da to request i + 1 (da = 1: draw pa made the pending tick A, slot 0 of step 0
    // is dead).  Request i takes service draw sk0 + i = sg + 128 s + p with sg = sk0 - da: an even sg aligns the service blocks with
    // the lanes, an odd one puts a block's first half into its lane's slot 1 and its second half into the NEXT lane's slot 0.
    const uint32_t da = (uint32_t)(ak0 & 1ull);
    const uint64_t pa = ak0 - da;
    const int64_t sg = (int64_t)sk0 - (int64_t)da;       // (-1 when nothing has been served yet and draw pa is consumed)
    const bool ds = (sg & 1ll) != 0;
    uint64_t blk_a = (pa >> 1) + (uint64_t)lane;
    uint64_t blk_s = (uint64_t)((sg + (ds ? 1 : 0)) >> 1) + (uint64_t)lane;
    double carry_sv = 0.0;                               // odd sg: the second half of the block before lane 0's
    if (run && ds && da == 0) {                          // (resuming in the middle of a service block)
        const uint64_t b = sk0 >> 1;
        const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), ssid0, ssid1, key0, key1);
        carry_sv = sec_d(nsd(div_lambda.div(exp1_from_uniform(res53(o.z, o.w)))));
    }
    const bool spec = end_ns < (1ll << 39);              // speculated whole-ns steps: a + F < 2^40 (below: F < 2^39)
    const double Td = (double)T;
    double base_a = (double)A;                           // arrival time of the step's slot 0
    double carryD = busy ? (double)Dprev : NEG, carryS = busy ? (double)Sprev : NEG;   // D and S of the request before the step's first
    double carryA = NEG;                                 // arrival of the last request of the step before (NEG: none in this window)
    double lt_d = (double)lt;
    bool fin = !run || A > T;
    int64_t n_arr_total = 0;
    double A_next = (double)A, a_last = (double)crtA, a_last2 = (double)crtA;   // pending tick and the two ticks before it
    int64_t r0 = -(int64_t)da;                           // request index of the step's slot 0
    bool first = true;
    int64_t steps_left = L.cap / R + 2;                  // (more steps than the record log has room for: an overflow, never a hang)

    while (__syncthreads_or(!fin)) {
        const bool act = !fin;
        if (act && --steps_left < 0) { overflow = 1; fin = true; }
        // ---- V: the step's stream values
        const U4 oa = philox4x32_10((uint32_t)blk_a, (uint32_t)(blk_a >> 32), asid0, asid1, key0, key1);
        const U4 os = philox4x32_10((uint32_t)blk_s, (uint32_t)(blk_s >> 32), ssid0, ssid1, key0, key1);
        blk_a += 64; blk_s += 64;
        const double inc0 = div_rate.div(exp1_from_uniform(res53(oa.x, oa.y))), inc1 = div_rate.div(exp1_from_uniform(res53(oa.z, oa.w)));
        const double e0 = div_lambda.div(exp1_from_uniform(res53(os.x, os.y))), e1 = div_lambda.div(exp1_from_uniform(res53(os.z, os.w)));
        const double v0 = sec_d(nsd(e0)), v1 = sec_d(nsd(e1));                     // Duration.from_seconds(sample).to_seconds()
        double sv0 = v0, sv1 = v1;
        if (ds) {
            double up = wave_shr1(v1);
            if (lane == 0) up = carry_sv;
            carry_sv = rl64(v1, 63);
            sv0 = up; sv1 = v0;
        }
        const double du0 = nsd(sv0), du1 = nsd(sv1);
        const bool dead0 = first && da != 0 && lane == 0;                         // slot 0 of step 0: no request
        // ---- C: arrival times.  a0 / a1 = arrival of slots 2 lane / 2 lane + 1, an = the tick after slot 2 lane + 1
        double a0, a1, an;
        {
            const double F0 = __dmul_rn(inc0, 1e9), F1 = __dmul_rn(inc1, 1e9);
            const double fl0 = __builtin_floor(F0), fl1 = __builtin_floor(F1);
            const double fr0 = __dsub_rn(F0, fl0), fr1 = __dsub_rn(F1, fl1);
            constexpr double m = 1.0 / 1024.0;
            bool safe0 = spec && fr0 >= m && fr0 <= 1.0 - m && F0 < 549755813888.0;
            const bool safe1 = spec && fr1 >= m && fr1 <= 1.0 - m && F1 < 549755813888.0;
            double f0 = safe0 ? fl0 : 0.0;
            const double f1 = safe1 ? fl1 : 0.0;
            if (dead0) { f0 = 0.0; safe0 = true; }
            const double ps = f0 + f1;
            double I = ps;                                                       // inclusive scan over the lanes (whole numbers: exact)
            { const double u = dpp_shr<1>(I); if ((lane & 15) >= 1) I += u; }
            { const double u = dpp_shr<2>(I); if ((lane & 15) >= 2) I += u; }
            { const double u = dpp_shr<4>(I); if ((lane & 15) >= 4) I += u; }
            { const double u = dpp_shr<8>(I); if ((lane & 15) >= 8) I += u; }
            { const double t15 = rl64(I, 15), t47 = rl64(I, 47); if (lane & 16) I += (lane & 32) ? t47 : t15; }
            { const double t31 = rl64(I, 31); if (lane & 32) I += t31; }
            a0 = base_a + (I - ps); a1 = a0 + f0; an = base_a + I;
            unsigned long long ub0 = __ballot(!safe0), ub1 = __ballot(!safe1);
            while ((ub0 | ub1) != 0ull) {                                        // the increments too close to a whole number, in order
                const int l0 = ub0 ? (int)__builtin_ctzll(ub0) : 64, l1 = ub1 ? (int)__builtin_ctzll(ub1) : 64;
                const bool at0 = l0 <= l1;                                       // position 2 l0 before position 2 l1 + 1
                const int lu = at0 ? l0 : l1;
                const double au = at0 ? rl64(a0, lu) : rl64(a1, lu), iu = at0 ? rl64(inc0, lu) : rl64(inc1, lu);
                const double dl = nsd(__dadd_rn(sec_d(au), iu)) - au;            // the reference's step, exactly
                if (at0) { if (lane >= lu) { a1 += dl; an += dl; } if (lane > lu) a0 += dl; ub0 &= ub0 - 1; }
                else { if (lane >= lu) an += dl; if (lane > lu) { a0 += dl; a1 += dl; } ub1 &= ub1 - 1; }
            }
        }
        const double base_next = rl64(an, 63);
        // ---- L: Lindley recursion as a (max, +) scan (hs_kernels_wide.hpp MaxPlus), two requests per lane
        MaxPlus F{dead0 ? NEG : a0 + du0, dead0 ? 0.0 : du0};
        F = mp_compose(MaxPlus{a1 + du1, du1}, F);
        { MaxPlus Pm{dpp_shr<1>(F.p), dpp_shr<1>(F.q)}; if ((lane & 15) >= 1) F = mp_compose(F, Pm); }
        { MaxPlus Pm{dpp_shr<2>(F.p), dpp_shr<2>(F.q)}; if ((lane & 15) >= 2) F = mp_compose(F, Pm); }
        { MaxPlus Pm{dpp_shr<4>(F.p), dpp_shr<4>(F.q)}; if ((lane & 15) >= 4) F = mp_compose(F, Pm); }
        { MaxPlus Pm{dpp_shr<8>(F.p), dpp_shr<8>(F.q)}; if ((lane & 15) >= 8) F = mp_compose(F, Pm); }
        {   // rows 1 and 3 take the row before them, then rows 2 and 3 take rows 0-1
            const MaxPlus t15{rl64(F.p, 15), rl64(F.q, 15)}, t47{rl64(F.p, 47), rl64(F.q, 47)};
            if (lane & 16) F = mp_compose(F, (lane & 32) ? t47 : t15);
            const MaxPlus t31{rl64(F.p, 31), rl64(F.q, 31)};
            if (lane & 32) F = mp_compose(F, t31);
        }
        const MaxPlus E{wave_shr1(F.p), wave_shr1(F.q)};                         // composition of the lanes before this one
        const double xq = carryD + E.q;                                          // (-inf + q = -inf: nothing before the first request)
        const double Dp_in = lane == 0 ? carryD : (E.p > xq ? E.p : xq);         // D of the request before this lane's first
        double S0 = a0 > Dp_in ? a0 : Dp_in, D0 = S0 + du0;
        if (dead0) { S0 = carryS; D0 = Dp_in; }                                  // (no request in the slot: S and D pass through)
        const double S1 = a1 > D0 ? a1 : D0, D1 = S1 + du1;
        double Sp_in = wave_shr1(S1);                                            // S of the request before this lane's first
        if (lane == 0) Sp_in = carryS;
        double Ap_in = wave_shr1(a1);                                            // arrival of the request before this lane's first
        if (lane == 0) Ap_in = carryA;
        // which reference events happen (Station::req_step), request by request
        const bool live0 = !dead0;
        const bool arr0 = act && live0 && a0 <= Td, arr1 = act && a1 <= Td;
        const bool st0 = arr0 && S0 <= Td, st1 = arr1 && S1 <= Td;
        const bool dp0 = st0 && D0 <= Td, dp1 = st1 && D1 <= Td;
        const bool hz0 = (arr0 && (a0 == Sp_in || a0 == Dp_in || a1 <= a0)) || (st0 && du0 == 0.0);
        const bool hz1 = (arr1 && (a1 == S0 || a1 == D0 || an <= a1)) || (st1 && du1 == 0.0);
        const unsigned long long b_arr0 = __ballot(arr0), b_arr1 = __ballot(arr1);
        const unsigned long long b_st0 = __ballot(st0), b_st1 = __ballot(st1), b_dp0 = __ballot(dp0), b_dp1 = __ballot(dp1);
        const bool hz = __ballot(hz0 || hz1) != 0ull;
        if (act && hz) { bailed = true; fin = true; }
        const bool ok = act && !bailed;
        const int n_arr_l = ok ? __popcll(b_arr0) + __popcll(b_arr1) : 0;
        const int n_dp_l = ok ? __popcll(b_dp0) + __popcll(b_dp1) : 0;
        if (ok) {
            n_tick += (uint32_t)n_arr_l;
            n_notify += (uint32_t)(__popcll(__ballot(arr0 && Sp_in < a0)) + __popcll(__ballot(arr1 && S0 < a1)));
            n_poll += (uint32_t)(__popcll(__ballot(arr0 && Dp_in < a0)) + __popcll(__ballot(arr1 && D0 < a1)));
            n_start += (uint32_t)(__popcll(b_st0) + __popcll(b_st1));
            n_dep += (uint32_t)n_dp_l;
            // the latest processed event of each request (a <= S < D); a later request may have arrived before an earlier one left
            const double e0_ = dp0 ? D0 : st0 ? S0 : arr0 ? a0 : NEG, e1_ = dp1 ? D1 : st1 ? S1 : arr1 ? a1 : NEG;
            const double em = e0_ > e1_ ? e0_ : e1_;
            lt_d = em > lt_d ? em : lt_d;
            // the one request (at most) that started and is still in service at the end of the window
            const unsigned long long bp0 = __ballot(st0 && !dp0), bp1 = __ballot(st1 && !dp1);
            if ((bp0 | bp1) != 0ull) {
                const bool at1 = bp0 == 0ull;
                const int lh = (int)__builtin_ctzll(at1 ? bp1 : bp0);
                pend_new = true;
                pendD = to_i64(at1 ? rl64(D1, lh) : rl64(D0, lh)); pendS = to_i64(at1 ? rl64(S1, lh) : rl64(S0, lh));
                pend_s = at1 ? rl64(sv1, lh) : rl64(sv0, lh);
                pendA_d = at1 ? rl64(a1, lh) : rl64(a0, lh);
                pendAp_d = at1 ? rl64(a0, lh) : rl64(Ap_in, lh);
                pend_i = r0 + 2 * lh + (at1 ? 1 : 0);
                // (slot 1 of lane 0 in step 0 with a dead slot 0 is request 0: pend_i == 0, pendAp_d unused)
            }
        }
        // ---- the step's records -> LDS (coalesced transposed write below), service samples for T
        if (!(flags & (1 << 19))) {
            st_a[w][2 * lane] = (ok && arr0) ? to_i64(a0) : INT64_MIN; st_a[w][2 * lane + 1] = (ok && arr1) ? to_i64(a1) : INT64_MIN;
            st_d[w][2 * lane] = (ok && dp0) ? to_i64(D0) : INT64_MIN; st_d[w][2 * lane + 1] = (ok && dp1) ? to_i64(D1) : INT64_MIN;
            if (lane == 0) { s_base[w][0] = accepted + r0; s_base[w][1] = sink_w + (int64_t)dep0 + r0; }
        }
        // ---- T: _total_service_time in completion order (departures are a prefix of the step's requests)
        if (n_dp_l > 0 && !(flags & (1 << 18))) {
            s_sv[w][2 * lane] = dp0 ? sv0 : 0.0; s_sv[w][2 * lane + 1] = dp1 ? sv1 : 0.0;
            const int hi0 = b_dp0 ? 2 * (63 - (int)__builtin_clzll(b_dp0)) : -1, hi1 = b_dp1 ? 2 * (63 - (int)__builtin_clzll(b_dp1)) + 1 : -1;
            const int cnt = (hi0 > hi1 ? hi0 : hi1) + 1;
            for (int c = 0; c < cnt; c += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = s_sv[w][c + q];              // (past cnt: this step's 0.0 / not yet departed: 0.0; + 0.0 is exact)
#pragma unroll
                for (int q = 0; q < 8; ++q) total_service = __dadd_rn(total_service, v[q]);
            }
        }
        if (ok) {
            // the pending tick and the two ticks before it (lineage, Station::req_finish); position of the first arrival beyond T
            const int dead = (first && da != 0) ? 1 : 0;
            const int pf = n_arr_l + dead;                                       // slot of the first request that did not arrive
            auto slot_a = [&](int p) { return (p & 1) ? rl64(a1, p >> 1) : rl64(a0, p >> 1); };
            if (n_arr_l > 0) {
                const double last = slot_a(pf - 1);
                a_last2 = n_arr_l >= 2 ? slot_a(pf - 2) : a_last;
                a_last = last;
            }
            n_arr_total += n_arr_l;
            A_next = pf < R ? slot_a(pf) : base_next;
            if (pf < R) fin = true;
        }
        // carries into the next step
        carryD = rl64(D1, 63); carryS = rl64(S1, 63); carryA = rl64(a1, 63);
        base_a = base_next;
        r0 += R;
        first = false;
        // ---- the workgroup writes the step's records: 16 neighbouring LPs' k-th records are one line
        __syncthreads();
        if (!(flags & (1 << 19))) {
#pragma unroll
            for (int it = 0; it < R / 64; ++it) {
                const int idx = it * (NW * 64) + (int)threadIdx.x;
                const int slot = idx / NW, ww = idx % NW;
                const int64_t va = st_a[ww][slot], vd = st_d[ww][slot];
                const int64_t ka = s_base[ww][0] + slot, kd = s_base[ww][1] + slot;
                if (va != INT64_MIN) { if (ka < L.cap) L.adm[(size_t)ka * n + lp0 + ww] = va; else overflow = 1; }
                if (vd != INT64_MIN) { if (kd < L.cap) L.sink_t[(size_t)kd * n + lp0 + ww] = vd; else overflow = 1; }
            }
        }
    }
    lt = (int64_t)lt_d;
    {   // the lanes' latest event: maximum over the wavefront
        long long mx = (long long)lt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const long long d = shfl_xor_ll(mx, o); mx = d > mx ? d : mx; }
        lt = mx;
    }
    const int ovf_w = __any(overflow) ? 1 : 0;

    // ---- fold the window into the LP's state (Station::req_finish): wave-uniform values, lane 0 stores
    if (pend_new) pend = true;
    Candidate mine = cand_none(live ? lp : 0);
    unsigned ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool count = run && !bailed;
    if (count) {
        const uint32_t c_tick = n_tick, c_notify = n_notify, c_poll = n_poll, c_start = n_start, c_dep = n_dep;
        ev[0] = c_tick; ev[1] = c_tick; ev[2] = c_notify; ev[3] = c_poll + c_dep; ev[4] = c_start; ev[5] = c_start; ev[6] = c_dep; ev[7] = c_dep;
        const int64_t acc2 = accepted + c_tick, st2 = started + c_start;
        const int64_t A_next_i = A != kInfNs ? (int64_t)A_next : A;
        const int64_t crtA2 = c_tick ? (int64_t)a_last : crtA0;
        uint32_t seq = X.seq[lp];
        uint32_t seqA = X.seqA[lp], seqD = X.seqD[lp];
        if ((c_tick | c_start) != 0u) {                  // creation stamps: only their order matters (Station::req_finish)
            const bool d_first = pend && pendS < crtA2;
            seqA = seq + (d_first ? 1u : 0u); seqD = seq + (d_first ? 0u : 1u); seq += 2u;
        }
        // lineage of what is pending now (Station::req_finish)
        int32_t dpA = X.dpA[lp], dpD = X.dpD[lp];
        int64_t rcA = X.rcA[lp], rcD = X.rcD[lp];
        if (c_tick) { dpA = 1; rcA = acc2 >= 2 ? (c_tick >= 2 ? (int64_t)a_last2 : L.adm[(size_t)(acc2 - 2) * n + lp]) : crtA0; }
        if (pend && pend_new) {
            const int64_t m = st2 - 1;                   // the request in service: it started at pendS
            if (pendS == (int64_t)pendA_d) {             // ... on arrival: six steps from its tick, which was created at the tick before
                dpD = 6;
                rcD = m >= 1 ? (pend_i >= 1 ? (int64_t)pendAp_d : (m - 1 < L.cap ? L.adm[(size_t)(m - 1) * n + lp] : 0)) : crtA0;
            } else {                                     // ... when request m - 1 left: four steps from that continuation, created when IT started
                Stream st;
                st.init(((uint64_t)key1 << 32) | key0, ((uint64_t)ssid1 << 32) | ssid0, (uint64_t)(m - 1));
                const double s_prev = seconds_from_ns(ns_from_seconds(div_lambda.div(exp1_from_uniform(st.next_uniform()))));
                dpD = 4; rcD = pendS - ns_from_seconds(s_prev);
            }
        }
        if (lane == 0) {
            X.generated[lp] += c_tick; X.accepted[lp] = acc2; X.started[lp] = st2; X.completed[lp] += c_dep;
            X.received[lp] += c_dep; X.sink_w[lp] = sink_w + c_dep;
            X.buf[lp] = (int64_t)c_tick - (int64_t)c_start;
            X.active[lp] = pend ? 1 : 0;
            X.D[lp] = pend ? pendD : kInfNs;
            if (pend) { X.crtD[lp] = pendS; X.svc_s[lp] = pend_s; }
            X.total_service[lp] = total_service;
            X.A[lp] = A_next_i; X.arr_time[lp] = A_next_i; X.arr_k[lp] = ak0 + (uint64_t)n_arr_total; X.svc_k[lp] = sk0 + (uint64_t)c_start;
            X.crtA[lp] = crtA2;
            if ((c_tick | c_start) != 0u) { X.seqA[lp] = seqA; X.seqD[lp] = seqD; X.seq[lp] = seq; }
            X.last_time[lp] = lt;
            if (c_tick) { X.dpA[lp] = (uint8_t)dpA; X.rcA[lp] = rcA; }
            if (pend && pend_new) { X.dpD[lp] = (uint8_t)dpD; X.rcD[lp] = rcD; }
            uint32_t tot_ev = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { X.ev_kind[(size_t)k * n + lp] += ev[k]; tot_ev += ev[k]; }
            X.events[lp] += tot_ev;
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
    if (bailed && lane == 0) {
        const unsigned pos = atomicAdd(&ctl->n_bail, 1u);
        bail[pos] = lp;
    }
    // ---- the workgroup's partial totals and candidate
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_ev[w][k] = count ? ev[k] : 0u;
        s_lt[w] = count ? (long long)lt : INT64_MIN;
        s_ovf[w] = ovf_w;
        wave_c[w] = mine;
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
}
