// hs_tables.hpp -- tick tables: the arrival times of every tick stream whose next tick is DEFINED by the reference's numerical
// procedure (`ArrivalTimeProvider.next_arrival_time`'s general path, load/arrival_time_provider.py:84-144: adaptive Simpson
// inside a bracket search and Brent's method) -- Sources with a time-varying profile (load/profile.py:52-113) and Probes
// (`_ProbeProfile` is not a ConstantRateProfile, instrumentation/probe.py:24-35) -- produced BEFORE the run by a kernel of
// their own.  Nothing in a simulation feeds back into a Source's tick times (tick k+1 is a pure function of tick k and of
// draw k of the Source's arrival stream), so the run kernels only read `times[row][k]`; they carry none of the inversion's
// code, registers or scratch.
//
// One wavefront per tick stream.  Brent's iteration and the bracket search are inherently sequential and are executed by all 64
// lanes redundantly (uniform values); each INTEGRAL is evaluated by the 64 lanes together:
//   * the recursion tree of integrate_adaptive_simpson (numerics/integration.py:11-90) is expanded breadth-first in LDS, one
//     lane per node of the current level -- a typical integral (10 - 100 intervals, 5 - 8 levels) is finished here;
//   * when a level holds >= kTasks nodes (or the node store is full) every node of that level becomes a TASK: lanes pull tasks
//     from an LDS counter and walk their sub-trees depth-first with the explicit stack of hs_profile.hpp (prof_walk), all lanes
//     converged on the one expensive piece of code (the three rate evaluations of a visit);
//   * values are added bottom-up level by level: value(node) = value(left) + value(right), the same two doubles the sequential
//     recursion adds, so not one bit changes (tools/simpson_split_check.py checks exactly this split on the host).
// The inputs the reference needs 10^7 - 10^8 rate evaluations for therefore cost ~1/60 of a lone lane,
// and ordinary arrivals stop paying 2.9 us per interval.
#pragma once

#include "hs_profile.hpp"

namespace hs {

struct TickRow {               // one tick stream (device-resident array of these)
    uint32_t kind;             // kProfLinearRamp / kProfSpike / kProfGeneralConstant
    uint32_t poisson;          // 1: the target area of tick k is E_k = -log(1 - u_k) of stream (seed, sid); 0: 1.0
    double p0, p1, p2, p3;
    uint64_t seed, sid;
    int32_t owner, pad;        // the LP / Source / probe the row belongs to (named in errors)
};

// what the run kernels see (device-resident; StationParams::tabs points at one)
struct TickTables {
    const int64_t *times;      // [n_rows][cap] tick k of row r at times[r * cap + k]; kInfNs: the stream has ended
    int64_t cap;
    const int32_t *src_row;    // [n_lp] row of the LP's time-varying Source, -1: none
    const int32_t *probe_row;  // [kMaxProbes][n_lp] row of the Probe in that slot, -1: none
    // ---- tandem queues (Server(downstream=<Server>), hs_station.hpp "tandem"): further per-engine arrays that only the PF
    // instantiations read live behind this pointer as well, so that the headline kernels' argument list stays what it was.
    // null = the engine has no Server that forwards to a Server.
    const int32_t *tandem;     // [kMaxUp + 2][n_lp]: rows 0 .. kMaxUp-1 the LPs whose forwarded Requests arrive here (-1: none,
                               // filled from 0); row kMaxUp the LP's pass; row kMaxUp + 1 the LP this LP's Server forwards to (-1: none)
    int64_t *inj_i;            // [kMaxUp][n_lp] forwards of each upstream LP consumed so far
    // lineage of forward record m of an LP (its time / created_at are record m of the LP's sink_t / sink_created logs): when the
    // root of the nanosecond group the forward was created in was created, that root's own lineage (its root's creation time;
    // steps | construction rank << 8), and how many steps below the root the forwarded Request is.  [cap][n_lp]
    int64_t *fw_rc, *fw_rrc, *fw_rdr, *fw_dep;
    // ... and the same root key for every entry of the in-group FIFO, plus the created_at an arriving forward carries.  [kQCap][n_lp]
    int64_t *q_rrc, *q_rdr, *q_pay;
    // every LP's first event beyond end_ns as the electing launch saw it: the election's check for ties that the lineage key does
    // not decide (hs_kernels.hpp hs_station_run, Totals::undecided).  [n_lp] {t, t_created, rcrt, depth | valid << 32}
    int64_t *cand_key;
    // several Sources per Server (no tandem queues): which of the LP's Sources the lineage of a pending departure / an in-group FIFO
    // entry goes back to -- the slot of the Source whose tick is the most recent one in its ancestry (255: none: an injected Request,
    // a Probe) -- so that the election ranks a departure by THAT Source's construction position (cand_rank pad 2 + slot) instead of
    // the LP's first-listed Source.  tools/election_rules.py (`rsrc`): right on 9 100 several-Source cases incl. the three the
    // stand-in gets wrong.  [C][n_lp] and [kQCap][n_lp]; null = the engine has no LP with several Sources.
    uint8_t *rs_dep, *rs_q;
    // the engine has Requests injected with Simulation.schedule(): a departure's lineage may go back to one of them, whose construction
    // rank (the order of the schedule() calls, across LPs) the engine does not carry -- every departure / injected Request candidate
    // ranks with a stand-in in the election's tie check (round 6: tools/gpu_random_sweep.py, tie case 120013 -- two Requests
    // scheduled for the end instant on two lock-step Servers; until then only LPs with several Sources were checked)
    int32_t standin_sched;
    // start of the run: the creation stamp of the events constructed before it (the Sources' first ticks).  The reference numbers
    // those first, then restarts the count for the run's own events -- so until the run has created as many events as there are
    // Sources, a new event can sort BEFORE a first tick of its nanosecond, and the breadth-first order the lineage key stands for
    // does not hold around first ticks (Station::pre_run_tie).
    int64_t t_start;
};
__device__ __forceinline__ int64_t tick_lookup(const int64_t *row, int64_t cap, int64_t k, int &overflow) {
    if (k < cap) return row[k];
    overflow |= 1;             // reported as HS_E_OVERFLOW (the table was sized from the rates; raise log_capacity)
    return kInfNs;
}

constexpr int kCoopNodes = 1024;   // nodes of the breadth-first part (44 B each in LDS)
constexpr int kCoopTasks = 128;    // a level this wide is handed to the depth-first walkers (two tasks per lane to balance)

struct CoopTree {                  // LDS, one per wavefront
    double a[kCoopNodes], b[kCoopNodes], fa[kCoopNodes], fb[kCoopNodes];
    double sw[kCoopNodes];         // S_whole of the node; after its visit: the node's VALUE
    int left[kCoopNodes];          // index of the left child (right = left + 1); -1: the value is final (leaf / walked sub-tree)
    int lvl_start[kSimpsonMaxDepth + 3];
    int counter;
};

// integrate_adaptive_simpson(rate_fn, a0, b0, tol0), all 64 lanes of the (single-wavefront) workgroup together; uniform
// arguments, uniform result.  `visits` (per lane) counts the intervals this lane looked at.
__device__ inline double coop_integrate(const Profile &pf, double a0, double b0, double tol0, CoopTree &T, long long &visits,
                                        long long lane_budget) {
    if (a0 == b0) return 0.0;
    const int lane = threadIdx.x & 63;
    {
        const double fa = prof_rate(pf, a0), fb = prof_rate(pf, b0);
        const double m = (a0 + b0) / 2.0;
        const double fm = prof_rate(pf, m);
        const double h = (b0 - a0) / 2.0;
        if (lane == 0) { T.a[0] = a0; T.b[0] = b0; T.fa[0] = fa; T.fb[0] = fb; T.sw[0] = prof_simpson3(fa, fm, fb, h); T.left[0] = -1; }
    }
    int lvl_begin = 0, lvl_end = 1, n_levels = 0;
    double tol = tol0;
    __syncthreads();
    for (;;) {
        const int F = lvl_end - lvl_begin;
        if (F == 0) break;
        if (lane == 0) T.lvl_start[n_levels] = lvl_begin;
        const int level = n_levels++;
        if (F >= kCoopTasks || lvl_end + 2 * F > kCoopNodes) {
            // ---- the level's nodes become tasks: depth-first walks, one visit per loop trip, lanes converged on the visit
            if (lane == 0) T.counter = 0;
            __syncthreads();
            ProfWalk W;
            bool have_task = false, done = false;
            int node = 0;
            for (;;) {
                if (!have_task && !done) {
                    const int i = atomicAdd(&T.counter, 1);
                    if (i >= F) done = true;
                    else {
                        node = lvl_begin + i;
                        W.begin(T.a[node], T.b[node], T.fa[node], T.fb[node], T.sw[node], tol, level);
                        have_task = true;
                    }
                }
                if (!__any(have_task)) break;
                if (have_task) {
                    if (W.step(pf, visits)) { T.sw[node] = W.ret; T.left[node] = -1; have_task = false; }
                    if (visits > lane_budget) { T.sw[node] = 0.0; T.left[node] = -1; have_task = false; done = true; }   // over budget: the caller gives up
                }
            }
            __syncthreads();
            if (lane == 0) T.lvl_start[n_levels] = lvl_end;
            break;
        }
        // ---- one breadth-first round: a lane per node of the level
        int new_end = lvl_end;
        for (int base = lvl_begin; base < lvl_end; base += 64) {
            const int i = base + lane;
            const bool act = i < lvl_end;
            bool split = false;
            double v = 0.0, m = 0.0, fm = 0.0, s_left = 0.0, s_right = 0.0, na = 0.0, nb = 0.0, nfa = 0.0, nfb = 0.0;
            if (act) {
                na = T.a[i]; nb = T.b[i]; nfa = T.fa[i]; nfb = T.fb[i];
                ++visits;
                split = !prof_visit(pf, na, nb, nfa, nfb, T.sw[i], tol, level, v, m, fm, s_left, s_right);
            }
            const unsigned long long mask = __ballot(split);
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            if (split) {
                const int c = new_end + 2 * pos;
                T.a[c] = na; T.b[c] = m; T.fa[c] = nfa; T.fb[c] = fm; T.sw[c] = s_left; T.left[c] = -1;
                T.a[c + 1] = m; T.b[c + 1] = nb; T.fa[c + 1] = fm; T.fb[c + 1] = nfb; T.sw[c + 1] = s_right; T.left[c + 1] = -1;
                T.left[i] = c;
            } else if (act) { T.sw[i] = v; T.left[i] = -1; }
            new_end += 2 * __popcll(mask);
        }
        lvl_begin = lvl_end; lvl_end = new_end;
        tol = tol / 2.0;
        if (lane == 0) T.lvl_start[n_levels] = lvl_begin;
        __syncthreads();
    }
    // ---- bottom-up, in the tree's own order: value(node) = value(left) + value(right)
    __syncthreads();
    for (int L = n_levels - 2; L >= 0; --L) {
        const int s = T.lvl_start[L], e = T.lvl_start[L + 1];
        for (int i = s + lane; i < e; i += 64) {
            const int c = T.left[i];
            if (c >= 0) T.sw[i] = T.sw[c] + T.sw[c + 1];
        }
        __syncthreads();
    }
    const double r = T.sw[0];
    __syncthreads();               // (the next integral overwrites node 0)
    return r;
}

struct CoopIntegrator {
    CoopTree *T;
    long long *visits;
    long long lane_budget;
    __device__ __forceinline__ double operator()(const Profile &pf, double a, double b, double tol) const {
        return coop_integrate(pf, a, b, tol, *T, *visits, lane_budget);
    }
    __device__ __forceinline__ bool over() const { return __any(*visits > lane_budget) != 0; }
};

}  // namespace hs
