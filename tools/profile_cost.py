#!/usr/bin/env python3
"""How expensive are a Source's first arrivals under a time-varying profile?  (diagnostic, CPU only)

The arrival times of `Source.with_profile(...)` are DEFINED by the reference's numerical procedure (adaptive Simpson
inside a bracket search and Brent's method, load/arrival_time_provider.py:84-144); csrc/hs_profile.hpp restates it for
the device.  For a few inputs the procedure itself needs ~10^8 rate evaluations for ONE arrival (DESIGN.md section 1.2:
minutes in the reference; the tick-table kernel, csrc/hs_tables.hpp, shares such an integral among 64 lanes).  This tool compiles the very same header for the host
(a 20-line stand-in for <hip/hip_runtime.h>, g++ -ffp-contract=off) and times the arrivals of one stream, so that a
profile / seed / station combination can be checked before a long run:

    python tools/profile_cost.py --ramp 3 1 9 --seed 77 --station 97
    python tools/profile_cost.py --spike 3 40 4 2 --seed 71 --station 3 --constant-arrivals
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "happy_simulator_amd", "csrc")

HIP_STANDIN = r"""
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(x)
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
static inline long long __double2ll_rz(double a) { return (long long)a; }
static inline double __ll2double_rn(long long a) { return (double)a; }
static inline long long __double_as_longlong(double a) { long long r; memcpy(&r, &a, 8); return r; }
static inline double __longlong_as_double(long long a) { double r; memcpy(&r, &a, 8); return r; }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
"""

MAIN = r"""
#include "hs_profile.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace hs;
#ifndef HS_TOOL_BUDGET_LOG2
#define HS_TOOL_BUDGET_LOG2 30       /* 64 lanes x the device's default 2^24 */
#endif
static const long long BUDGET = 1ll << HS_TOOL_BUDGET_LOG2;
int main(int argc, char **argv) {
    Profile pf; pf.kind = (uint32_t)atoi(argv[1]);
    pf.p0 = atof(argv[2]); pf.p1 = atof(argv[3]); pf.p2 = atof(argv[4]); pf.p3 = atof(argv[5]);
    const uint64_t seed = strtoull(argv[6], nullptr, 10), base = strtoull(argv[7], nullptr, 10);
    const int poisson = atoi(argv[8]), n_arr = atoi(argv[9]);
    const double limit = atof(argv[10]);
    Stream s; s.init(seed, stream_id(base, kStreamArrival), 0);
    int64_t t = 0;
    for (int k = 0; k < n_arr; ++k) {
        const double area = poisson ? exp1_from_uniform(s.next_uniform()) : 1.0;
        const double rate = prof_rate(pf, seconds_from_ns_ieee(t));
        const auto t0 = std::chrono::steady_clock::now();
        bool over = false;
        const int64_t t2 = prof_next_arrival(pf, t, area, BUDGET, over);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("arrival %d: from %.9f s, target area %.6g, rate there %.6g/s, first bracket %.4g s -> %.9f s   host %.4f s%s\n",
               k, t / 1e9, area, rate, rate > 0 ? 2.0 * area / rate : 0.1, t2 == kInfNs ? INFINITY : t2 / 1e9, dt,
               over ? "   <-- over 64 x the device's default evaluation budget per lane (hs_engine_set_profile_budget raises it)" :
               dt > limit ? "   <-- slow" : "");
        if (over) break;
        if (t2 == kInfNs || t2 <= t) break;
        t = t2;
    }
}
"""


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--ramp", nargs=3, type=float, metavar=("DURATION_S", "START_RATE", "END_RATE"))
    g.add_argument("--spike", nargs=4, type=float, metavar=("BASELINE", "SPIKE_RATE", "WARMUP_S", "SPIKE_DURATION_S"))
    ap.add_argument("--seed", type=int, default=42, help="Simulation seed (the Philox key)")
    ap.add_argument("--station", type=int, default=0, help="stream base of the Source = index of its station / chain")
    ap.add_argument("--constant-arrivals", action="store_true", help="Source.with_profile(poisson=False): target area 1.0")
    ap.add_argument("--arrivals", type=int, default=8)
    ap.add_argument("--slow-s", type=float, default=0.01, help="host seconds per arrival above which a line is flagged")
    ap.add_argument("--budget-log2", type=int, default=0, help="compile with another evaluation budget (2^N Simpson intervals per arrival)")
    a = ap.parse_args()
    kind, p = (1, list(a.ramp) + [0.0]) if a.ramp else (2, list(a.spike))
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "hip"))
        open(os.path.join(d, "hip", "hip_runtime.h"), "w").write(HIP_STANDIN)
        open(os.path.join(d, "main.cpp"), "w").write(MAIN)
        exe = os.path.join(d, "profile_cost")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", d, "-I", CSRC,
                               *([f"-DHS_TOOL_BUDGET_LOG2={a.budget_log2}"] if a.budget_log2 else []),
                               os.path.join(d, "main.cpp"), "-o", exe])
        return subprocess.call([exe, str(kind), *[repr(x) for x in p], str(a.seed), str(a.station),
                                "0" if a.constant_arrivals else "1", str(a.arrivals), repr(a.slow_s)])


if __name__ == "__main__":
    sys.exit(main())
