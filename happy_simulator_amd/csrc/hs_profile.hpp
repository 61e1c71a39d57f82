// hs_profile.hpp -- time-varying arrival rates on the device: the general path of
// `ArrivalTimeProvider.next_arrival_time` (happysimulator/load/arrival_time_provider.py:84-144).
//
// The next arrival after t0 is the time t with  integral_{t0}^{t} rate(x) dx = target  (target = 1.0 for deterministic
// arrivals, -log(1-u) for Poisson).  The reference finds it numerically -- adaptive Simpson
// (numerics/integration.py:11-90) inside a bracket search and Brent's method (numerics/root_finding.py:27-152) -- so the
// arrival times are DEFINED by those procedures; this file restates them operation for operation (binary64, same order,
// no contraction: the translation units are built with -ffp-contract=off) for the reference's parametric profiles
// (load/profile.py:38-113).  Recursion becomes an explicit per-lane stack (one frame per depth, <= 51 frames).
#pragma once

#include "hs_device.hpp"

namespace hs {

enum : uint32_t { kProfConstant = 0, kProfLinearRamp = 1, kProfSpike = 2,
                  kProfGeneralConstant = 3 /* constant rate p0 through the general path (instrumentation/probe.py:24-35) */ };

struct Profile {
    uint32_t kind;
    double p0, p1, p2, p3;   // ramp: duration_s, start_rate, end_rate;  spike: baseline, spike_rate, warmup_s, spike_duration_s
    int32_t owner = -1;      // the LP the inversion runs for (named when the evaluation budget is exceeded)
};

// Evaluation budget of ONE next_arrival_time call.  The reference's adaptive Simpson rule works on a rate that is quantised to
// whole nanoseconds; where that quantisation noise exceeds the (halving) tolerance -- a wide bracket on a ramp: a first arrival
// drawn from a near-zero rate, or an arrival that spans the end of a ramp towards a low rate -- its error test is only met once
// the intervals are narrower than 1 ns: 10^6 .. 10^8 rate evaluations for a single arrival (seconds to minutes in the reference's
// Python).  Typical arrivals need 10-100 intervals.  The inversion runs in the tick-table kernel
// (hs_tables.hpp: 64 lanes per integral); the budget is a RUN-TIME argument of that kernel (hs_engine_set_profile_budget) and
// only a time guard: an arrival that exceeds it is reported (HS_E_UNSUPPORTED naming the LP), never guessed.

// rate_fn(t) = profile.get_rate(Instant.from_seconds(t))
__device__ __forceinline__ double prof_rate(const Profile &pf, double t_seconds) {
    const double t = seconds_from_ns_ieee(ns_from_seconds(t_seconds));
    if (pf.kind == kProfLinearRamp) {
        if (t <= 0.0) return pf.p1;
        if (t >= pf.p0) return pf.p2;
        const double fraction = t / pf.p0;
        return pf.p1 + fraction * (pf.p2 - pf.p1);
    }
    if (pf.kind == kProfGeneralConstant) return pf.p0;
    if (t < pf.p2) return pf.p0;                       // SpikeProfile
    if (t < pf.p2 + pf.p3) return pf.p1;
    return pf.p0;
}

__device__ __forceinline__ double prof_simpson3(double fa, double fm, double fb, double h) {
    return h / 3.0 * (fa + 4.0 * fm + fb);
}

constexpr int kSimpsonMaxDepth = 50;

// One visit of the recursion (the body of _adaptive_simpson_recursive, numerics/integration.py:46-90) for the node
// [a, b] with f(a), f(b), S_whole, tolerance and depth given.  True: a leaf, `v` is its value (Richardson extrapolation);
// false: the node splits at `m` (f(m) = fm) into halves whose S_whole are s_left / s_right and whose tolerance is tol / 2.
__device__ __forceinline__ bool prof_visit(const Profile &pf, double a, double b, double fa, double fb, double sw, double tol,
                                           int depth, double &v, double &m, double &fm, double &s_left, double &s_right) {
    m = (a + b) / 2.0;
    const double h = (b - a) / 2.0;
    fm = prof_rate(pf, m);
    const double lm = (a + m) / 2.0;
    const double rm = (m + b) / 2.0;
    const double flm = prof_rate(pf, lm);
    const double frm = prof_rate(pf, rm);
    s_left = prof_simpson3(fa, flm, fm, h / 2.0);
    s_right = prof_simpson3(fm, frm, fb, h / 2.0);
    const double s_combined = s_left + s_right;
    const double error_estimate = (s_combined - sw) / 15.0;
    if (depth >= kSimpsonMaxDepth || fabs(error_estimate) < tol) { v = s_combined + error_estimate; return true; }
    return false;
}

// The recursion below one node as an explicit stack (one frame per depth, left half first), advanced one visit or one return
// at a time so that many lanes walking different sub-trees stay converged on the visit (hs_tables.hpp).
struct ProfWalk {
    double A[kSimpsonMaxDepth + 1], B[kSimpsonMaxDepth + 1], FA[kSimpsonMaxDepth + 1], FB[kSimpsonMaxDepth + 1];
    double SW[kSimpsonMaxDepth + 1], TOL[kSimpsonMaxDepth + 1], M[kSimpsonMaxDepth + 1], FM[kSimpsonMaxDepth + 1];
    double SR[kSimpsonMaxDepth + 1], LEFT[kSimpsonMaxDepth + 1];
    uint8_t STAGE[kSimpsonMaxDepth + 1];
    int d, depth0;
    bool have;
    double ret;
    __device__ __forceinline__ void begin(double a, double b, double fa, double fb, double sw, double tol, int depth) {
        A[0] = a; B[0] = b; FA[0] = fa; FB[0] = fb; SW[0] = sw; TOL[0] = tol; STAGE[0] = 0;
        d = 0; depth0 = depth; have = false; ret = 0.0;
    }
    // true: the walk is complete, `ret` is the node's value
    __device__ __forceinline__ bool step(const Profile &pf, long long &visits) {
        if (!have) {
            ++visits;
            double v, m, fm, s_left, s_right;
            if (prof_visit(pf, A[d], B[d], FA[d], FB[d], SW[d], TOL[d], depth0 + d, v, m, fm, s_left, s_right)) {
                ret = v;
                have = true;
            } else {                                         // recurse on the left half first
                M[d] = m; FM[d] = fm; SR[d] = s_right; STAGE[d] = 1;
                A[d + 1] = A[d]; B[d + 1] = m; FA[d + 1] = FA[d]; FB[d + 1] = fm; SW[d + 1] = s_left; TOL[d + 1] = TOL[d] / 2.0;
                STAGE[d + 1] = 0;
                ++d;
                return false;
            }
        }
        if (d == 0) return true;
        --d;
        if (STAGE[d] == 1) {                                 // left result in: now the right half
            LEFT[d] = ret; STAGE[d] = 2;
            A[d + 1] = M[d]; B[d + 1] = B[d]; FA[d + 1] = FM[d]; FB[d + 1] = FB[d]; SW[d + 1] = SR[d]; TOL[d + 1] = TOL[d] / 2.0;
            STAGE[d + 1] = 0;
            ++d;
            have = false;
        } else ret = LEFT[d] + ret;                          // left_result + right_result
        return false;
    }
};

// integrate_adaptive_simpson(rate_fn, a, b, tol) for a <= b on ONE lane; `budget` = intervals it may still visit (< 0 afterwards:
// it gave up).  Used by the host-side tools and the device's reference chain (hs_debug_tick_table); the engine's tables come
// from the cooperative integrator of hs_tables.hpp.
__device__ inline double prof_integrate(const Profile &pf, double a0, double b0, double tol0, long long &budget) {
    if (a0 == b0) return 0.0;
    ProfWalk W;
    {
        const double fa = prof_rate(pf, a0), fb = prof_rate(pf, b0);
        const double m = (a0 + b0) / 2.0;
        const double fm = prof_rate(pf, m);
        const double h = (b0 - a0) / 2.0;
        W.begin(a0, b0, fa, fb, prof_simpson3(fa, fm, fb, h), tol0, 0);
    }
    long long visits = 0;
    for (;;) {
        const bool fin = W.step(pf, visits);
        if (visits > budget) { budget = -1; return 0.0; }    // over budget: the caller gives up on this arrival
        if (fin) break;
    }
    budget -= visits;
    return W.ret;
}
struct LoneIntegrator {
    long long *budget;
    __device__ __forceinline__ double operator()(const Profile &pf, double a, double b, double tol) const {
        return prof_integrate(pf, a, b, tol, *budget);
    }
    __device__ __forceinline__ bool over() const { return *budget < 0; }
};

template <class Integ>
struct ProfObjective {
    const Profile *pf; double t_start, target;
    const Integ *integ;
    __device__ __forceinline__ double operator()(double t) const {
        // (Brent and the bracket search stay at or above t_start; the a > b branch of the integrator is not reachable)
        return (*integ)(*pf, t_start, t, 1e-10) - target;
    }
};

__device__ __forceinline__ double py_min(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double py_max(double a, double b) { return b > a ? b : a; }

// brentq(f, a, b); false: not converged / no sign change (the reference raises)
template <class Integ>
__device__ inline bool prof_brentq(const ProfObjective<Integ> &f, double a, double b, double &root) {
    const double xtol = 1e-12, rtol = 4 * 2.220446049250313e-16;
    double fa = f(a), fb = f(b);
    if (fa * fb > 0) return false;
    if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
    double c = a, fc = fa, d = b - a, e = d;
    for (int iteration = 0; iteration < 100; ++iteration) {
        const double tol = 2.0 * rtol * fabs(b) + xtol;
        const double m = (c - b) / 2.0;
        if (fabs(m) <= tol || fb == 0) { root = b; return true; }
        if (fabs(e) >= tol && fabs(fa) > fabs(fb)) {
            const double s = fb / fa;
            double p, q;
            if (a == c) { p = 2.0 * m * s; q = 1.0 - s; }                    // secant
            else {                                                           // inverse quadratic interpolation
                q = fa / fc;
                const double r = fb / fc;
                p = s * (2.0 * m * q * (q - r) - (b - a) * (r - 1.0));
                q = (q - 1.0) * (r - 1.0) * (s - 1.0);
            }
            if (p > 0) q = -q; else p = -p;
            if (2.0 * p < py_min(3.0 * m * q - fabs(tol * q), fabs(e * q))) { e = d; d = p / q; }
            else { d = m; e = m; }
        } else { d = m; e = m; }
        a = b; fa = fb;
        if (fabs(d) > tol) b = b + d;
        else if (m > 0) b = b + tol;
        else b = b - tol;
        fb = f(b);
        if (fb * fc > 0) { c = a; fc = fa; d = b - a; e = d; }
        else if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    }
    root = b;
    return false;
}

// next_arrival_time from t_start_ns; kInfNs when the reference would raise (rate zero for ever, no convergence) or -- `over` --
// when the integrator ran out of its evaluation budget
template <class Integ>
__device__ inline int64_t prof_next_arrival_with(const Profile &pf, int64_t t_start_ns, double target_area, const Integ &integ,
                                                 bool &over) {
    const double t_start_sec = seconds_from_ns_ieee(t_start_ns);
    ProfObjective<Integ> f{&pf, t_start_sec, target_area, &integ};
    const double current_rate = prof_rate(pf, t_start_sec);
    double t_high;
    if (current_rate > 0) {
        double estimated_delay = (target_area / current_rate) * 2.0;         // optimistic linear prediction
        estimated_delay = py_max(1e-9, py_min(estimated_delay, 3600.0));
        t_high = t_start_sec + estimated_delay;
    } else t_high = t_start_sec + 0.1;
    const double t_low = t_start_sec;
    bool found = false;
    for (int i = 0; i < 50; ++i) {                                           // bracket search, geometric expansion
        if (f(t_high) > 0) { found = true; break; }
        if (integ.over()) break;
        const double step = py_max(1e-6, t_high - t_low);
        t_high += step * 2.0;
    }
    if (integ.over()) { over = true; return kInfNs; }
    if (!found) return kInfNs;
    double root;
    const bool ok = prof_brentq(f, t_low, t_high, root);
    if (integ.over()) { over = true; return kInfNs; }
    if (!ok) return kInfNs;
    return ns_from_seconds(root);                                            // Instant.from_seconds(result.root)
}
// ... on one lane (tools, the debug reference chain)
__device__ inline int64_t prof_next_arrival(const Profile &pf, int64_t t_start_ns, double target_area, long long budget, bool &over) {
    LoneIntegrator li{&budget};
    return prof_next_arrival_with(pf, t_start_ns, target_area, li, over);
}

}  // namespace hs
