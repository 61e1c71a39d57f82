/* hs_rng_ref.h -- ORACLE-SIDE scalar semantics (TEST INFRASTRUCTURE, not product).
 *
 * Plain-C restatement of the numeric rules the reference applies on the hot
 * path, plus the counter-based random streams ("Oracle-B").
 * The product (happy_simulator_amd/csrc) has its OWN implementation of the
 * same definitions for gfx950; the two are compared bit-for-bit by tests/.
 *
 * Reference rules restated here (all paths relative to /root/reference):
 *   - Instant/Duration.from_seconds(float x) = int(x * 1e9), truncation toward
 *     zero                       happysimulator/core/temporal.py:62, :205
 *   - Instant + float d = ns + int(d * 1e9)   core/temporal.py:222
 *   - to_seconds() = float(ns) / 1e9          core/temporal.py:66, :211
 *   - exponential sample = -log(1 - u) / lambda, lambda = 1 / mean
 *                                 distributions/exponential.py:36,43
 *   - Poisson target integral = -log(1 - u)   load/providers/poisson_arrival.py:31
 *   - constant-rate next arrival: from_seconds(to_seconds(t) + E / rate)
 *                                 load/arrival_time_provider.py:72-82
 *   - stock uniforms are MT19937 genrand_res53 (CPython `random`, numpy legacy
 *     `np.random`), SURVEY.md Appendix A5.
 *
 * Stream definition of the Philox-plugged mode (ours; shared by oracle, golden
 * generator and the HIP engine -- see DESIGN.md "Random streams"):
 *   block(seed, sid, b) = Philox4x32-10(ctr = {lo32 b, hi32 b, lo32 sid, hi32 sid},
 *                                       key = {lo32 seed, hi32 seed})
 *   u(seed, sid, k)     = res53(block[2*(k&1)], block[2*(k&1)+1]),  b = k >> 1
 *   res53(a, b)         = ((a >> 5) * 2^26 + (b >> 6)) / 2^53
 *   sid                 = (entity_stream_base << 3) | kind
 *   E                   = -hs_log(1.0 - u)
 * hs_log is a fixed sequence of IEEE-754 binary64 operations (no FMA
 * contraction); compile with -ffp-contract=off.
 */
#ifndef HS_RNG_REF_H
#define HS_RNG_REF_H

#include <stdint.h>
#include <string.h>

enum { HS_STREAM_ARRIVAL = 0, HS_STREAM_SERVICE = 1, HS_STREAM_LINK = 2, HS_STREAM_ROUTE = 3, HS_STREAM_KEY = 4, HS_STREAM_LOSS = 5 };

static inline void hsr_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double hsr_res53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* k-th uniform of stream `sid` under run seed `seed`. */
static inline double hsr_uniform(uint64_t seed, uint64_t sid, uint64_t k) {
    uint64_t b = k >> 1;
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)(b >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    hsr_philox4x32_10(ctr, key, o);
    return (k & 1) ? hsr_res53(o[2], o[3]) : hsr_res53(o[0], o[1]);
}

/* Natural log for normal positive x (the engine only ever passes x in
 * [2^-53, 1]).  Argument reduction x = 2^k * (1+f), sqrt(2)/2 < 1+f <= sqrt(2),
 * then log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)), s = f/(2+f), R = degree-14
 * even minimax polynomial (the classic 7-term Remez fit).  Every operation
 * below is one IEEE binary64 op, evaluated in the order written. */
static inline double hs_log_ref(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t ix;
    memcpy(&ix, &x, 8);
    uint32_t hx = (uint32_t)(ix >> 32);
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u; /* 1+f > sqrt(2): use x/2 */
    hx |= (i ^ 0x3ff00000u);
    k += (int32_t)(i >> 20);
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffffu);
    double m;
    memcpy(&m, &ix, 8);
    double f = m - 1.0;
    double hfsq = (0.5 * f) * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return (((s * (hfsq + R) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}

/* Unit-mean exponential variate from a uniform in [0,1). */
static inline double hsr_exp1(double u) { return -hs_log_ref(1.0 - u); }

/* --- time algebra (core/temporal.py) ------------------------------------ */
static inline int64_t hsr_ns_from_seconds(double x) { return (int64_t)(x * 1e9); } /* :205 */
static inline double hsr_seconds_from_ns(int64_t ns) { return (double)ns / 1e9; }  /* :211 */

/* ------------------------------------------------------------------------
 * MT19937 (stock reference streams, "Oracle-A").  Standard Matsumoto-Nishimura
 * generator; `init_genrand` is numpy's legacy `np.random.seed(int)`,
 * `init_by_array` is CPython's `random.seed(int)` (key = 32-bit limbs of the
 * absolute value).  Checked against both libraries in tests/test_oracle_rng.py.
 * ---------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } hsr_mt19937;

static inline void hsr_mt_init_genrand(hsr_mt19937 *g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static inline void hsr_mt_init_by_array(hsr_mt19937 *g, const uint32_t *key, int len) {
    hsr_mt_init_genrand(g, 19650218u);
    int i = 1, j = 0;
    int k = 624 > len ? 624 : len;
    for (; k; --k) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = 623; k; --k) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
    }
    g->mt[0] = 0x80000000u;
    g->idx = 624;
}

static inline uint32_t hsr_mt_next(hsr_mt19937 *g) {
    if (g->idx >= 624) {
        for (int kk = 0; kk < 624; ++kk) {
            uint32_t y = (g->mt[kk] & 0x80000000u) | (g->mt[(kk + 1) % 624] & 0x7fffffffu);
            g->mt[kk] = g->mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static inline double hsr_mt_res53(hsr_mt19937 *g) {
    uint32_t a = hsr_mt_next(g), b = hsr_mt_next(g);
    return hsr_res53(a, b);
}

#endif /* HS_RNG_REF_H */
