// hs_device.hpp -- gfx950 device-side scalar semantics of the engine.
//
// Everything the event logic needs that must be bit-identical to the reference's
// arithmetic (references relative to /root/reference/happysimulator):
//   time algebra      int64 ns; from_seconds(x) = trunc(x * 1e9); to_seconds(ns) = double(ns) / 1e9
//                     (core/temporal.py:62,66,205,211,222)
//   arrivals          ns' = from_seconds(to_seconds(ns) + E / rate)       (load/arrival_time_provider.py:72-82)
//   service           s = to_seconds(from_seconds(sample)); D = S + from_seconds(s)
//                     (components/server/server.py:246-250, core/event.py:499)
//   exponential       sample = E / lambda, lambda = 1 / mean              (distributions/exponential.py:36,43)
// and the engine's counter-based streams (DESIGN.md "Random streams"):
//   u(seed, sid, k) = res53 of half of Philox4x32-10(ctr = {k>>1, sid}, key = seed);  E = -hs_log(1 - u).
//
// fp64 ops are written with the __d*_rn intrinsics so that no a*b+c is ever contracted into an FMA,
// independent of compiler flags; the file is also compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hs {

constexpr int64_t kInfNs = INT64_MAX;  // Instant.Infinity (core/temporal.py:298-368)

enum StreamKind : uint32_t { kStreamArrival = 0, kStreamService = 1, kStreamLink = 2, kStreamRoute = 3, kStreamKey = 4, kStreamLoss = 5 };

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one widening 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ double res53(uint32_t a, uint32_t b) {
    // ((a >> 5) * 2^26 + (b >> 6)) / 2^53 -- every step exact in binary64
    const double hi = __dmul_rn((double)(a >> 5), 67108864.0);
    return __dmul_rn(__dadd_rn(hi, (double)(b >> 6)), 1.0 / 9007199254740992.0);
}

// Natural log for normal positive x (engine passes x in [2^-53, 1]).  Same operation sequence as
// oracle/hs_rng_ref.h hs_log_ref and tests/golden/hs_streams_py.py hs_log.
__device__ __forceinline__ double hs_log(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                     Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                     Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    uint64_t ix = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    hx |= (i ^ 0x3ff00000u);
    k += (int32_t)(i >> 20);
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffffull);
    const double m = __longlong_as_double((long long)ix);
    const double f = __dsub_rn(m, 1.0);
    const double hfsq = __dmul_rn(__dmul_rn(0.5, f), f);
    const double s = __ddiv_rn(f, __dadd_rn(2.0, f));
    const double z = __dmul_rn(s, s);
    const double w = __dmul_rn(z, z);
    const double t1 = __dmul_rn(w, __dadd_rn(Lg2, __dmul_rn(w, __dadd_rn(Lg4, __dmul_rn(w, Lg6)))));
    const double t2 = __dmul_rn(
        z, __dadd_rn(Lg1, __dmul_rn(w, __dadd_rn(Lg3, __dmul_rn(w, __dadd_rn(Lg5, __dmul_rn(w, Lg7)))))));
    const double R = __dadd_rn(t2, t1);
    const double dk = (double)k;
    double r = __dadd_rn(__dmul_rn(s, __dadd_rn(hfsq, R)), __dmul_rn(dk, ln2_lo));
    r = __dsub_rn(r, hfsq);
    r = __dadd_rn(r, f);
    return __dadd_rn(r, __dmul_rn(dk, ln2_hi));
}

__device__ __forceinline__ double exp1_from_uniform(double u) { return -hs_log(__dsub_rn(1.0, u)); }

__device__ __forceinline__ int64_t ns_from_seconds(double x) { return __double2ll_rz(__dmul_rn(x, 1e9)); }
// double(ns) / 1e9, correctly rounded: the constant-divisor sequence of ConstDiv below with b = 1e9 and
// y = RN(1 / 1e9) = 1e-9 (bit-identical to the IEEE quotient; five multiply/FMA instead of a division)
__device__ __forceinline__ double seconds_from_ns(int64_t ns) {
    const double a = __ll2double_rn(ns);
    const double q0 = __dmul_rn(a, 1e-9);
    const double r0 = __fma_rn(-1e9, q0, a);
    const double q1 = __fma_rn(r0, 1e-9, q0);
    const double r1 = __fma_rn(-1e9, q1, a);
    return __fma_rn(r1, 1e-9, q1);
}
// Whole nanoseconds in [0, 2^52) are EXACT in binary64, so the time algebra can stay in fp64 registers where the values allow it
// (the uniform-grid kernels check it on the host): from_seconds is one v_trunc_f64 instead of the multi-instruction f64 -> i64
// conversion, to_seconds starts from the value itself instead of an i64 -> f64 conversion, and the int64 a log or a comparison
// needs is an add and a mask.  Bit-identical to ns_from_seconds / seconds_from_ns on that range.
__device__ __forceinline__ double ns_from_seconds_d(double x) { return __builtin_trunc(__dmul_rn(x, 1e9)); }
__device__ __forceinline__ double seconds_from_ns_d(double ns) {
    const double q0 = __dmul_rn(ns, 1e-9);
    const double r0 = __fma_rn(-1e9, q0, ns);
    const double q1 = __fma_rn(r0, 1e-9, q0);
    const double r1 = __fma_rn(-1e9, q1, ns);
    return __fma_rn(r1, 1e-9, q1);
}
__device__ __forceinline__ int64_t i64_from_whole_d(double d) {   // whole d in [0, 2^52)
    return (int64_t)((uint64_t)__double_as_longlong(__dadd_rn(d, 4503599627370496.0)) & 0xFFFFFFFFFFFFFull);
}
// the plain IEEE division, kept for the device-vs-division test hook
__device__ __forceinline__ double seconds_from_ns_ieee(int64_t ns) { return __ddiv_rn(__ll2double_rn(ns), 1e9); }

// a / b for a divisor that is constant per LP (1e9, a source's rate, a server's lambda): y = RN(1 / b) is
// computed once with a true division, then every quotient costs one multiply and four FMAs instead of the
// ~13-instruction v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence.  The result is the correctly
// rounded quotient, bit-identical to a / b (Markstein 1990: with y = RN(1/b) and q faithful, the residual
// r = a - b q is exact in one FMA and RN(q + r y) = RN(a / b); the first correction makes q faithful, the
// second makes it exact).  The theorem excludes divisors whose significand is all ones and needs the
// intermediates to stay normal; `fast` is false for such divisors and the hardware division is used.
// tests/test_gpu_parity.py::test_constant_divisor_quotients_are_ieee compares 10^7 quotients per divisor.
struct ConstDiv {
    double b, y;
    bool fast;
    __device__ __forceinline__ void init(double divisor) {
        b = divisor;
        y = __ddiv_rn(1.0, divisor);
        const uint64_t bits = (uint64_t)__double_as_longlong(divisor);
        const uint32_t ex = (uint32_t)(bits >> 52) & 0x7ffu;
        const bool all_ones = (bits & 0x000fffffffffffffull) == 0x000fffffffffffffull;
        fast = !all_ones && ex > 1023u - 200u && ex < 1023u + 200u && (bits >> 63) == 0;
    }
    // a must be finite and either 0 or of magnitude in [2^-300, 2^300] (every caller: event times, -log(1-u))
    __device__ __forceinline__ double div(double a) const {
        if (!fast) return __ddiv_rn(a, b);
        const double q0 = __dmul_rn(a, y);
        const double r0 = __fma_rn(-b, q0, a);
        const double q1 = __fma_rn(r0, y, q0);
        const double r1 = __fma_rn(-b, q1, a);
        return __fma_rn(r1, y, q1);
    }
};

// One entity stream.  A Philox block serves two consecutive draws, so the block is cached and only
// recomputed when the draw index crosses an even boundary.
struct Stream {
    uint32_t key0, key1, sid0, sid1;
    uint64_t k;        // next draw index
    uint32_t c2, c3;   // second half of the cached block (valid when k is odd)

    __device__ __forceinline__ void init(uint64_t seed, uint64_t sid, uint64_t k0) {
        key0 = (uint32_t)seed; key1 = (uint32_t)(seed >> 32);
        sid0 = (uint32_t)sid; sid1 = (uint32_t)(sid >> 32);
        k = k0;
        if (k0 & 1) {  // resume in the middle of a block
            const uint64_t b = k0 >> 1;
            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), sid0, sid1, key0, key1);
            c2 = o.z; c3 = o.w;
        } else {
            c2 = c3 = 0;
        }
    }
    __device__ __forceinline__ double next_uniform() {
        double u;
        if ((k & 1) == 0) {
            const uint64_t b = k >> 1;
            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), sid0, sid1, key0, key1);
            c2 = o.z; c3 = o.w;
            u = res53(o.x, o.y);
        } else {
            u = res53(c2, c3);
        }
        ++k;
        return u;
    }
    // Four consecutive draws at once: whatever the parity of k they need exactly the two blocks (k + 1) / 2 and (k + 1) / 2 + 1, so
    // both are computed in straight-line code (two independent chains the scheduler interleaves) and the parity only selects which
    // halves are which -- the same values and the same cached half as four calls of next_uniform().
    __device__ __forceinline__ void next4(double (&u)[4]) {
        const bool odd = (k & 1ull) != 0;
        const uint64_t b = (k + 1) >> 1;
        const U4 f = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), sid0, sid1, key0, key1);
        const U4 s = philox4x32_10((uint32_t)(b + 1), (uint32_t)((b + 1) >> 32), sid0, sid1, key0, key1);
        u[0] = res53(odd ? c2 : f.x, odd ? c3 : f.y);
        u[1] = res53(odd ? f.x : f.z, odd ? f.y : f.w);
        u[2] = res53(odd ? f.z : s.x, odd ? f.w : s.y);
        u[3] = res53(odd ? s.x : s.z, odd ? s.y : s.w);
        c2 = s.z; c3 = s.w;
        k += 4;
    }
    // ... and two: exactly the block (k + 1) / 2 whatever the parity
    __device__ __forceinline__ void next2(double (&u)[2]) {
        const bool odd = (k & 1ull) != 0;
        const uint64_t b = (k + 1) >> 1;
        const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), sid0, sid1, key0, key1);
        u[0] = res53(odd ? c2 : o.x, odd ? c3 : o.y);
        u[1] = res53(odd ? o.x : o.z, odd ? o.y : o.w);
        c2 = o.z; c3 = o.w;
        k += 2;
    }
};

__host__ __device__ __forceinline__ uint64_t stream_id(uint64_t base, uint32_t kind) { return (base << 3) | kind; }

// ---- wave-wide data movement on the VALU (DPP) -----------------------------------------------------------------------
#ifdef __HIPCC__   // (tools/simpson_split_check.py compiles this header for the host, where the builtin does not exist)
// v_mov_b32_dpp with an identity for the lanes that have no source (and for the rows outside the row mask RM): the scans below
// combine unconditionally instead of selecting per lane.  CTRL: 0x110 + N row_shr:N, 0x138 wave_shr:1, 0x142 row_bcast:15 (lane
// 15 of every row to the next row), 0x143 row_bcast:31 (lane 31 to rows 2 and 3).
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_or0(double v) {                 // identity 0.0 (bound_ctrl: no source = zero)
    const long long b = __double_as_longlong(v);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, RM, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_orneg(double v) {               // identity -infinity
    const long long b = __double_as_longlong(v);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, true);
    hi = __builtin_amdgcn_update_dpp((int)0xfff00000u, hi, CTRL, RM, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_orv(double v, double other) {   // identity: a given value
    const long long b = __double_as_longlong(v), o = __double_as_longlong(other);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp((int)(unsigned)(o & 0xffffffffll), lo, CTRL, RM, 0xf, false);
    hi = __builtin_amdgcn_update_dpp((int)(o >> 32), hi, CTRL, RM, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_orpos(double v) {               // identity +infinity
    const long long b = __double_as_longlong(v);
    int lo = (int)(unsigned)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, true);
    hi = __builtin_amdgcn_update_dpp((int)0x7ff00000u, hi, CTRL, RM, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
#endif

}  // namespace hs
