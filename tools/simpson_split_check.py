#!/usr/bin/env python3
"""Design check for DESIGN.md section 9 "arrival tables from a cooperative kernel" (CPU only, no GPU code): can ONE adaptive-
Simpson integral of csrc/hs_profile.hpp be evaluated by many lanes without changing a bit, and how well does it balance?

The value of a node of the recursion is a pure function of (a, b, f(a), f(b), S_whole, tolerance, depth):
    leaf:      S_left + S_right + (S_left + S_right - S_whole) / 15
    otherwise: value(left half) + value(right half)
so whoever computes the halves, the root's bits are those of the sequential walk as long as every `left + right` is formed
from the same two doubles.  This tool compiles the very header the engine uses for the host (tools/profile_cost.py's
stand-in for <hip/hip_runtime.h>) and, for a set of integrals including the pathological ones of DESIGN.md section 1.2,
  1. expands the tree breadth-first until the frontier holds >= TASKS sub-trees (what the lanes would do together),
  2. evaluates every frontier sub-tree on its own with a sequential walk, assigning them to LANES lanes either statically
     (round robin) or dynamically (next free lane takes the next task -- an LDS counter on the device),
  3. adds the values bottom-up in the tree's own order,
and compares the result bitwise with prof_integrate() of the header.  It prints the intervals visited in total, by the busiest
lane under both assignments, and the speed-up over one lane that this bounds.

    python tools/simpson_split_check.py [--lanes 64] [--tasks 1024]
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "happy_simulator_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from profile_cost import HIP_STANDIN  # noqa: E402

MAIN = r"""
#include "hs_profile.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <vector>
using namespace hs;

struct Node { double a, b, fa, fb, sw, tol; int depth; int left = -1, right = -1; bool leaf = false; double value = 0.0; };

// one visit of the recursion (the body of prof_integrate's loop): leaf value, or the two children
static bool visit(const Profile &pf, const Node &n, double &leaf_value, Node &l, Node &r) {
    const double m = (n.a + n.b) / 2.0, h = (n.b - n.a) / 2.0;
    const double fm = prof_rate(pf, m);
    const double lm = (n.a + m) / 2.0, rm = (m + n.b) / 2.0;
    const double flm = prof_rate(pf, lm), frm = prof_rate(pf, rm);
    const double s_left = prof_simpson3(n.fa, flm, fm, h / 2.0), s_right = prof_simpson3(fm, frm, n.fb, h / 2.0);
    const double s_combined = s_left + s_right;
    const double error_estimate = (s_combined - n.sw) / 15.0;
    if (n.depth >= kSimpsonMaxDepth || fabs(error_estimate) < n.tol) { leaf_value = s_combined + error_estimate; return true; }
    l = Node{n.a, m, n.fa, fm, s_left, n.tol / 2.0, n.depth + 1};
    r = Node{m, n.b, fm, n.fb, s_right, n.tol / 2.0, n.depth + 1};
    return false;
}

// a sub-tree on its own, sequentially (what one lane does with one task); counts the intervals it visits
static double walk(const Profile &pf, const Node &n, long long &visits) {
    ++visits;
    double v; Node l, r;
    if (visit(pf, n, v, l, r)) return v;
    const double left = walk(pf, l, visits);
    const double right = walk(pf, r, visits);
    return left + right;
}

int main(int argc, char **argv) {
    const int lanes = atoi(argv[1]), tasks = atoi(argv[2]);
    struct Case { const char *name; Profile pf; double a, b; };
    std::vector<Case> cases;
    auto ramp = [](double dur, double r0, double r1) { Profile p; p.kind = kProfLinearRamp; p.p0 = dur; p.p1 = r0; p.p2 = r1; p.p3 = 0; return p; };
    auto spike = [](double base, double peak, double warm, double dur) { Profile p; p.kind = kProfSpike; p.p0 = base; p.p1 = peak; p.p2 = warm; p.p3 = dur; return p; };
    cases.push_back({"ramp 3 s 1->9, [0, 9.12] (DESIGN 1.2: station 97)", ramp(3, 1, 9), 0.0, 9.12});
    cases.push_back({"ramp 2.21 s 25->2.25, [1.852, 3.223] (lb_profile_spec(1351))", ramp(2.21, 25.0, 2.25), 1.852134, 3.223});
    cases.push_back({"ramp 2.78 s 25->2.01, [2.2, 4.9]", ramp(2.78, 25.0, 2.01), 2.2, 4.9});
    cases.push_back({"ramp 5 s 3->20, [0.3, 0.45] (ordinary)", ramp(5, 3, 20), 0.3, 0.45});
    cases.push_back({"spike 3/40 at 4 s for 2 s, [3.5, 6.5]", spike(3, 40, 4, 2), 3.5, 6.5});
    cases.push_back({"ramp 10 s 0.5->30, [0, 4]", ramp(10, 0.5, 30), 0.0, 4.0});
    int bad = 0;
    for (const Case &c : cases) {
        long long budget = 1ll << 40;
        const double want = prof_integrate(c.pf, c.a, c.b, 1e-10, budget);
        const long long seq_visits = (1ll << 40) - budget;
        // 1. breadth-first expansion
        std::vector<Node> T;
        {
            const double fa = prof_rate(c.pf, c.a), fb = prof_rate(c.pf, c.b), m = (c.a + c.b) / 2.0, h = (c.b - c.a) / 2.0;
            T.push_back(Node{c.a, c.b, fa, fb, prof_simpson3(fa, prof_rate(c.pf, m), fb, h), 1e-10, 0});
        }
        std::vector<int> frontier{0};
        long long bfs_visits = 0, bfs_rounds = 0;
        while (!frontier.empty() && (int)frontier.size() < tasks) {
            std::vector<int> next;
            ++bfs_rounds;
            for (int i : frontier) {                       // (all of one round at once on the device: a lane per node)
                ++bfs_visits;
                double v; Node l, r;
                if (visit(c.pf, T[i], v, l, r)) { T[i].leaf = true; T[i].value = v; continue; }
                T[i].left = (int)T.size(); T.push_back(l);
                T[i].right = (int)T.size(); T.push_back(r);
                next.push_back(T[i].left); next.push_back(T[i].right);
            }
            frontier.swap(next);
        }
        // 2. the frontier's sub-trees, each on its own
        std::vector<long long> cost(frontier.size(), 0);
        for (size_t q = 0; q < frontier.size(); ++q) { Node &n = T[frontier[q]]; n.value = walk(c.pf, n, cost[q]); n.leaf = true; }
        long long total = bfs_visits, stat_max = 0, dyn_max = 0;
        {
            std::vector<long long> lane(lanes, 0);
            for (size_t q = 0; q < cost.size(); ++q) lane[q % lanes] += cost[q];
            stat_max = *std::max_element(lane.begin(), lane.end());
            std::priority_queue<long long, std::vector<long long>, std::greater<long long>> free_at;   // next free lane takes the next task
            for (int i = 0; i < lanes; ++i) free_at.push(0);
            for (long long k : cost) { const long long t = free_at.top(); free_at.pop(); free_at.push(t + k); total += k; }
            while (!free_at.empty()) { dyn_max = free_at.top(); free_at.pop(); }
        }
        // 3. bottom-up in the tree's own order (children were appended after their parents: walk the array backwards)
        for (int i = (int)T.size() - 1; i >= 0; --i) if (!T[i].leaf) T[i].value = T[T[i].left].value + T[T[i].right].value;
        const bool same = memcmp(&want, &T[0].value, 8) == 0;
        bad += same ? 0 : 1;
        const double crit_dyn = (double)(bfs_rounds + dyn_max), crit_stat = (double)(bfs_rounds + stat_max);
        printf("%-62s %s  intervals %10lld (sequential walk %10lld)  tasks %5zu after %2lld rounds | busiest lane: static %9lld, dynamic %9lld | "
               "speed-up over one lane: static %5.1fx, dynamic %5.1fx\n",
               c.name, same ? "bit-identical" : "DIFFERENT", total, seq_visits, frontier.size(), bfs_rounds, stat_max, dyn_max,
               seq_visits / crit_stat, seq_visits / crit_dyn);
    }
    return bad ? 1 : 0;
}
"""


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--lanes", type=int, default=64)
    ap.add_argument("--tasks", type=int, default=1024, help="expand breadth-first until the frontier holds this many sub-trees")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "hip"))
        open(os.path.join(d, "hip", "hip_runtime.h"), "w").write(HIP_STANDIN)
        open(os.path.join(d, "main.cpp"), "w").write(MAIN)
        exe = os.path.join(d, "split_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", d, "-I", CSRC,
                               os.path.join(d, "main.cpp"), "-o", exe])
        return subprocess.call([exe, str(a.lanes), str(a.tasks)])


if __name__ == "__main__":
    sys.exit(main())
