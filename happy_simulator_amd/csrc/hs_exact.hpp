// hs_exact.hpp -- the PROLOGUE of a run: the reference's heap loop, event by event, on one lane.
//
// Why it exists.  The reference orders same-nanosecond events by `_sort_index`, and that index comes from TWO counters
// (SURVEY.md A1): events constructed before `run()` -- the first SourceEvent of every Source, the first tick of every
// Probe (`Simulation.__init__`, core/simulation.py:145-160), Events handed to `Simulation.schedule()`
// (core/simulation.py:195-206) -- take consecutive values of the process-wide counter that `Simulation.__init__` reset
// (core/event.py:53-77), while `run()` switches Event construction to the heap's own counter, which starts again at 0
// (core/event_heap.py:48, core/sim_future.py:64-73).  So while the run has constructed fewer than N_init events
// (N_init = sources + probes + scheduled Events), a run-time event can carry a SMALLER index than a pre-run event that is
// still pending, and on a shared nanosecond it overtakes it: the first injected Request's Notify / Poll / Deliver run
// before a second Request injected for the same instant, a probe's first tick is overtaken by the Requests of the first
// source ticks, ... -- and when the two indices are EQUAL, `heapq`'s array layout decides.  None of that is LP-local: the
// run-time index is a global creation count.
//
// The parallel engines (hs_station.hpp, hs_netstation.hpp) order an LP's same-nanosecond events by LP-local creation
// stamps, which is exact once every run-time index exceeds every pre-run index, i.e. after the first N_init
// constructions.  This file covers the stretch before that: ONE lane executes the reference's loop literally --
//   * every reference event is materialised {time, sort index, LP, kind} in a binary heap whose push / pop follow CPython's
//     `heapq` sift procedures step for step (Lib/heapq.py `_siftdown` / `_siftup`), so equal (time, index) keys pop in
//     heapq's order;
//   * handlers restate the reference handler by handler (citations below) and construct events in the reference's order,
//     taking indices from one global counter G;
//   * all per-LP state lives in the SAME struct-of-arrays the parallel kernels use (StationState / NetState / RecordLogs),
//     and pending ticks / departures / probe ticks carry their TRUE sort index as creation stamp --
// until G >= N_init and the current timestamp group is complete.  Then every LP's creation counter continues from G and the
// parallel kernel takes over in the same hs_engine_run_until call (messages in flight on links are handed to the
// destination's bag / link queue).  Runs whose every pre-run event is a Source's first tick (the headline grid, the ring,
// the load balancer) have no LP on which the two counters can meet and skip the prologue entirely.
//
// Cost: O(N_init) events at a few microseconds each (dependent global loads on one lane) -- tens of microseconds for the tie
// storms this exists for, ~0.3 s for 65 536 stations that all carry a probe.
#pragma once

#include "hs_netstation.hpp"

namespace hs {

enum : uint16_t { XE_TICK = 0, XE_ENQ = 1, XE_NOTIFY = 2, XE_POLL = 3, XE_DELIVER = 4, XE_WORK = 5, XE_CONT = 6, XE_SINK = 7,
                  XE_LINK = 8, XE_LINKCONT = 9, XE_ROUTE = 10, XE_PTICK = 13, XE_PSAMPLE = 14, XE_SCHED = 15 };

struct XEvent {
    int64_t t;        // Event.time
    uint64_t idx;     // Event._sort_index
    int64_t cr;       // context["created_at"] of the Request it carries; DELIVER / WORK: the request's admission ordinal
    int64_t aux;      // LINK / LINKCONT: link id;  DELIVER: pool entry of the payload;  LINKCONT: see ts
    int64_t ts;       // LINKCONT: send time (the message's creation stamp in the parallel engines)
    // lineage (hs_station.hpp StationState::dpA ...): when it was created (INT64_MIN: before run()), how many steps after the
    // root of the group it was created in, and when that root was created -- filled by xpush from XState's context
    int64_t crt, rcrt;
    int32_t dep;
    int32_t lp;
    uint16_t code, slot;
};

struct XState {       // device memory, one per engine
    XEvent *heap;
    int64_t heap_len, heap_cap;
    unsigned long long G;        // the heap's own counter (run-time sort indices)
    unsigned long long n_init;   // pre-run events: indices 0 .. n_init-1 of the process-wide counter
    int32_t phase;               // 0 not started, 1 running, 2 handed over to the parallel engine
    int32_t err;                 // 1 heap full, 2 payload pool full
    // FIFO of the payload Events' sort indices (the queued payload is re-pushed with its ORIGINAL index at delivery,
    // components/queue_driver.py:86-90): a pool of list cells, one list per LP
    int32_t *qhead, *qtail;      // [n_lp]
    int32_t *pnext;              // [pool_cap]
    unsigned long long *pidx;    // [pool_cap]
    int64_t pool_n, pool_cap;
    int64_t processed;           // events processed by the prologue (telemetry)
    // Equal (time, index) keys between a pending pre-run event and a pending run-time event pop in heapq's layout order, which
    // only this loop reproduces: init_t[i] = time of the pending pre-run event with index i (-1: processed); tc = the latest
    // time at which such a pair is still pending when G reaches n_init (INT64_MIN: none; INT64_MAX: not computed yet) -- the
    // hand-over waits until the clock has passed it
    int64_t *init_t;             // [n_init]
    int64_t tc;
    // lineage context: the event being processed (ctx_on = 0: none, events are being constructed before run())
    int64_t ctx_t, ctx_rc;
    int32_t ctx_dp, ctx_on;
};
constexpr uint16_t kXInitFlag = 0x8000;   // XEvent::slot: constructed before run()

struct XInit {        // pre-run events in the order the reference constructs them
    const int32_t *src_lp;       // [n_src]   LPs of the Sources in `sources=[...]` order
    const uint8_t *src_slot;     // [n_src]   ... and their slot on that LP (0 = the LP's first Source, 1.. = further ones)
    const uint8_t *lp_src_slots; // [kMaxXSrc + 1][n_lp] per_lp: the slots of every LP's Sources in ITS construction order (255 = end)
    const int32_t *probe_lp;     // [n_probe] LPs of the Probes in `probes=[...]` order
    const uint8_t *probe_slot;   // [n_probe] ... and their slot on that LP
    const int32_t *sched_lp;     // [n_sched] LP of the j-th Event handed to schedule(), in construction order
    const int64_t *sched_entry;  // [n_sched] its position in StationParams::sched_t
    const int64_t *sched_rank;   // [n_sched] its position among all Events the caller constructed (cancelled ones leave gaps)
    int32_t n_src, n_probe;
    int64_t n_sched;
    uint32_t *sched_idx;         // [n_sched] OUT, indexed like sched_t: the Event's sort index
    // HS_MODE_REPLICAS: one prologue per LP; sched_entry / sched_rank are sorted by (LP, rank) so that LP i's Events are the
    // positions [sched_off[i], sched_off[i + 1]); the buffers of XState[0] are cut into per-LP slices of these sizes
    int32_t per_lp;
    int64_t heap_cap_lp, pool_cap_lp, init_cap_lp;
    // 1: never hand over -- the whole run on this loop.  Tandem queues (Server -> Server) whose nanosecond ties the parallel
    // engine's lineage key does not decide (Totals::undecided), and tandem queues next to Probes / scheduled Requests
    int32_t no_handover;
};

namespace xdetail {

__device__ __forceinline__ bool xlt(const XEvent &a, const XEvent &b) {      // Event.__lt__, core/event.py:337-344
    if (a.t != b.t) return a.t < b.t;
    return a.idx < b.idx;
}
// heapq.heappush: append, then _siftdown(heap, 0, len - 1)
__device__ inline void xpush(XState &S, const XEvent &e0) {
    if (S.heap_len >= S.heap_cap) { S.err |= 1; return; }
    XEvent e = e0;
    e.crt = S.ctx_on ? S.ctx_t : INT64_MIN;
    e.dep = S.ctx_on ? (S.ctx_dp >= 254 ? 255 : S.ctx_dp + 1) : 0;
    e.rcrt = S.ctx_on ? S.ctx_rc : INT64_MIN;
    int64_t pos = S.heap_len++;
    while (pos > 0) {
        const int64_t parent = (pos - 1) >> 1;
        if (!xlt(e, S.heap[parent])) break;
        S.heap[pos] = S.heap[parent];
        pos = parent;
    }
    S.heap[pos] = e;
}
// heapq.heappop: the last leaf replaces the root; _siftup walks the hole down to a leaf along the smaller child (the
// right one unless left < right), then _siftdown bubbles the moved item back up
__device__ inline XEvent xpop(XState &S) {
    const XEvent top = S.heap[0];
    const XEvent last = S.heap[--S.heap_len];
    const int64_t n = S.heap_len;
    if (n == 0) return top;
    int64_t pos = 0, child = 1;
    while (child < n) {
        const int64_t right = child + 1;
        if (right < n && !xlt(S.heap[child], S.heap[right])) child = right;
        S.heap[pos] = S.heap[child];
        pos = child;
        child = 2 * pos + 1;
    }
    while (pos > 0) {
        const int64_t parent = (pos - 1) >> 1;
        if (!xlt(last, S.heap[parent])) break;
        S.heap[pos] = S.heap[parent];
        pos = parent;
    }
    S.heap[pos] = last;
    return top;
}

__device__ __forceinline__ XEvent xev(int64_t t, uint64_t idx, uint16_t code, int32_t lp, int64_t cr = 0, int64_t aux = 0,
                                      uint16_t slot = 0, int64_t ts = 0) {
    XEvent e;
    e.t = t; e.idx = idx; e.cr = cr; e.aux = aux; e.ts = ts; e.lp = lp; e.code = code; e.slot = slot;
    e.crt = INT64_MIN; e.rcrt = INT64_MIN; e.dep = 0;
    return e;
}

__device__ inline double xuniform(uint64_t seed, uint64_t sid, uint64_t k) {
    Stream s;
    s.init(seed, sid, k);
    return s.next_uniform();
}

}  // namespace xdetail

// The sequential loop.  Called by lane 0 of hs_exact_run; returns true when the parallel engine takes over.
// `only` >= 0 (HS_MODE_REPLICAS): LP `only` is a Simulation of its own -- its own heap, its own two counters, its own clock
// (X.last_time) -- and S / I.sched_* are that LP's slices.
__device__ inline bool exact_loop(const StationParams &P, const NetParams &NP, const StationState &X, const NetState &NX,
                                  const RecordLogs &L, Totals *tot, XState &S, const XInit &I, int n, int C, bool net,
                                  int64_t start_ns, int64_t end_ns, int only = -1) {
    using namespace xdetail;
    if (S.phase == 0) { S.ctx_on = 0; S.ctx_t = INT64_MIN; S.ctx_rc = INT64_MIN; S.ctx_dp = 0; }
    if (S.phase == 0 && only >= 0) {
        unsigned long long g = 0;
        const int lp = only;
        for (int q = 0; q <= kMaxXSrc; ++q) {                // this Simulation's Sources in `sources=[...]` order
            const int slot = I.lp_src_slots != nullptr ? I.lp_src_slots[(size_t)q * (size_t)n + lp] : (q == 0 && P.src_kind[lp] != 0 ? 0 : 255);
            if (slot == 255) break;
            if (slot == 0) {
                if (X.A[lp] == kInfNs) continue;
                X.seqA[lp] = (uint32_t)g; S.init_t[g] = X.A[lp];
                xpush(S, xev(X.A[lp], g++, XE_TICK, lp, 0, 0, kXInitFlag));
            } else {
                const size_t o = (size_t)(slot - 1) * (size_t)n + lp;
                if (X.XA[o] == kInfNs) continue;
                X.seqX[o] = (uint32_t)g; S.init_t[g] = X.XA[o];
                xpush(S, xev(X.XA[o], g++, XE_TICK, lp, 0, 0, (uint16_t)(kXInitFlag | slot)));
            }
        }
        for (int j = 0; j < kMaxProbes; ++j) {
            const size_t o = (size_t)j * (size_t)n + lp;
            if (P.probe_metric[o] == kProbeNone || X.PA[o] == kInfNs) continue;
            X.seqP[o] = (uint32_t)g; S.init_t[g] = X.PA[o];
            xpush(S, xev(X.PA[o], g++, XE_PTICK, lp, 0, 0, (uint16_t)(kXInitFlag | j)));
        }
        const unsigned long long g0 = g;
        if (P.sched_off != nullptr)
            for (int64_t j = P.sched_off[lp]; j < P.sched_off[lp + 1]; ++j) {   // (sorted by construction rank inside the LP)
                const int64_t e = I.sched_entry[j];
                const int64_t t = P.sched_t[e];
                g = g0 + (unsigned long long)I.sched_rank[j];
                I.sched_idx[e] = (uint32_t)g;
                S.init_t[g] = t;
                xpush(S, xev(t, g++, XE_SCHED, lp, t, 0, kXInitFlag));
            }
        S.n_init = g; S.G = 0; S.tc = INT64_MAX; S.phase = 1;
    }
    if (S.phase == 0) {
        // Simulation.__init__: sources in list order, then probes (core/simulation.py:145-160); then the Events the caller
        // built for schedule(), in construction order -- all numbered by the process-wide counter
        unsigned long long g = 0;
        for (int i = 0; i < I.n_src; ++i) {
            const int lp = I.src_lp[i];
            const int slot = I.src_slot != nullptr ? I.src_slot[i] : 0;
            if (slot == 0) {
                const int64_t a = X.A[lp];
                if (a == kInfNs) continue;                  // "Rate is zero indefinitely. Source will not start." (source.py:137-139)
                X.seqA[lp] = (uint32_t)g;
                S.init_t[g] = a;
                xpush(S, xev(a, g++, XE_TICK, lp, 0, 0, kXInitFlag));
            } else {                                        // one of the further Sources of the LP's Server
                const size_t o = (size_t)(slot - 1) * (size_t)n + lp;
                const int64_t a = X.XA[o];
                if (a == kInfNs) continue;
                X.seqX[o] = (uint32_t)g;
                S.init_t[g] = a;
                xpush(S, xev(a, g++, XE_TICK, lp, 0, 0, (uint16_t)(kXInitFlag | slot)));
            }
        }
        for (int i = 0; i < I.n_probe; ++i) {
            const int lp = I.probe_lp[i];
            const size_t o = (size_t)I.probe_slot[i] * (size_t)n + lp;
            const int64_t a = X.PA[o];
            if (a == kInfNs) continue;
            X.seqP[o] = (uint32_t)g;
            S.init_t[g] = a;
            xpush(S, xev(a, g++, XE_PTICK, lp, 0, 0, (uint16_t)(kXInitFlag | I.probe_slot[i])));
        }
        const unsigned long long g0 = g;
        for (int64_t j = 0; j < I.n_sched; ++j) {
            const int64_t e = I.sched_entry[j];
            const int64_t t = P.sched_t[e];
            g = g0 + (unsigned long long)I.sched_rank[j];
            I.sched_idx[e] = (uint32_t)g;
            S.init_t[g] = t;
            xpush(S, xev(t, g++, XE_SCHED, I.sched_lp[j], t, 0, kXInitFlag));   // context["created_at"] = its own time (event.py:176)
        }
        S.n_init = g;
        S.G = 0;
        S.tc = INT64_MAX;
        S.phase = 1;
    }
    if (S.phase != 1) return false;

    unsigned long long evk[15];
    for (int k = 0; k < 15; ++k) evk[k] = 0;
    unsigned long long n_completed = 0, n_received = 0;
    int overflow = 0;
    int64_t cur = only >= 0 ? X.last_time[only] : tot->cur_time;
    bool handover = false;
    const size_t N = (size_t)n;

    while (S.heap_len > 0 && cur <= end_ns && S.err == 0) {          // core/simulation.py:472 tests the PREVIOUS event's time
        const XEvent e = xpop(S);
        if (e.slot & kXInitFlag) S.init_t[e.idx] = -1;
        if (e.t < cur) continue;                                     // time-travel drop, core/simulation.py:480-489
        cur = e.t;
        const int lp = e.lp;
        const int64_t t = e.t;
        // lineage context of whatever this event constructs: created earlier -> it is the root of a chain of this nanosecond's
        // group; created in this very nanosecond -> it carries its group's context
        S.ctx_on = 1; S.ctx_t = t;
        if (e.crt < t) { S.ctx_dp = 0; S.ctx_rc = e.crt; } else { S.ctx_dp = e.dep; S.ctx_rc = e.rcrt; }
        const uint8_t lin_dp = (uint8_t)(S.ctx_dp >= 254 ? 255 : S.ctx_dp + 1);   // ... of a pending event it creates
        int kind = e.code == XE_SCHED ? 1 : (int)e.code;
        evk[kind]++;
        S.processed++;
        X.events[lp] += 1;
        X.last_time[lp] = t;
        if (kind <= 10) X.ev_kind[(size_t)kind * N + lp] += 1;
        switch (e.code) {
        case XE_TICK: {
            // Source.handle_event (load/source.py:142-180): payload first, then the next SourceEvent
            if ((e.slot & 0xff) != 0) {                              // one of the further Sources of the LP's Server
                const int j = (int)(e.slot & 0xff) - 1;
                const size_t o = (size_t)j * N + lp;
                X.x_n[o] += 1;
                const int64_t stop = P.xsrc_stop[o];
                const bool payload = !(stop >= 0 && t > stop);
                unsigned long long idx_p = 0;
                if (payload) idx_p = S.G++;
                double area = 1.0;
                if (P.xsrc_kind[o] == 1) {
                    const uint64_t k = X.x_k[o];
                    area = exp1_from_uniform(xuniform(P.seed[lp], xsrc_stream_id(P.stream_base[lp], j), k));
                    X.x_k[o] = k + 1;
                }
                const int64_t a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(X.x_arr[o]), __ddiv_rn(area, P.xsrc_rate[o])));
                X.x_arr[o] = a2;
                if (payload) xpush(S, xev(t, idx_p, XE_ENQ, lp, t));
                const unsigned long long idx_t = S.G++;
                xpush(S, xev(a2, idx_t, XE_TICK, lp, 0, 0, (uint16_t)(j + 1)));
                X.XA[o] = a2 < t ? kInfNs : a2;
                X.seqX[o] = (uint32_t)idx_t; X.crtX[o] = t; X.dpX[o] = lin_dp; X.rcX[o] = S.ctx_rc;
                break;
            }
            X.generated[lp] += 1;
            const int64_t stop = P.src_stop[lp];
            const bool payload = !(stop >= 0 && t > stop);           // SimpleEventProvider.get_events :68
            const uint32_t vk = P.svc_kind[lp];
            const bool direct_sink = vk == 2;                        // the Source feeds a Sink / Counter directly
            const bool has_target = !direct_sink || P.egress[lp] == 1;
            unsigned long long idx_p = 0;
            if (payload && has_target) idx_p = S.G++;
            double area = 1.0;                                       // constant_arrival.py:23
            if (P.src_kind[lp] == 1) {                               // poisson_arrival.py:31
                const uint64_t k = X.arr_k[lp];
                area = exp1_from_uniform(xuniform(P.seed[lp], stream_id(P.stream_base[lp], kStreamArrival), k));
                X.arr_k[lp] = k + 1;
            }
            int64_t a2;
            if (P.prof_kind[lp] != kProfConstant)                    // tick number `generated` of the Source's table (hs_tables.hpp)
                a2 = tick_lookup(P.tabs->times + (size_t)P.tabs->src_row[lp] * (size_t)P.tabs->cap, P.tabs->cap, X.generated[lp], overflow);
            else
                a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(X.arr_time[lp]), __ddiv_rn(area, P.src_rate[lp])));
            X.arr_time[lp] = a2;
            if (payload && has_target) xpush(S, xev(t, idx_p, direct_sink ? XE_SINK : XE_ENQ, lp, t));
            if (a2 != kInfNs) {                                      // RuntimeError: the source is exhausted (:176-180)
                const unsigned long long idx_t = S.G++;
                xpush(S, xev(a2, idx_t, XE_TICK, lp));
                X.A[lp] = a2 < t ? kInfNs : a2;                      // a tick in the past is popped and dropped
                X.seqA[lp] = (uint32_t)idx_t; X.crtA[lp] = t; X.dpA[lp] = lin_dp; X.rcA[lp] = S.ctx_rc;
            } else X.A[lp] = kInfNs;
        } break;
        case XE_SCHED:
            X.sched_i[lp] += 1;
            [[fallthrough]];
        case XE_ENQ: {
            // QueuedResource.handle_event -> Queue._handle_enqueue (components/queue.py:122-147)
            const int64_t qcap = P.qcap[lp], buf = X.buf[lp];
            if (qcap >= 0 && buf >= qcap) { X.dropped[lp] += 1; break; }     // FIFOQueue.push refuses (queue_policy.py:94-98)
            const int64_t acc = X.accepted[lp];
            if (acc < L.cap) L.adm[log_at(L, acc, lp, N)] = e.cr; else overflow |= 1;
            if (S.pool_n >= S.pool_cap) { S.err |= 2; break; }
            const int32_t pe = (int32_t)S.pool_n++;
            S.pidx[pe] = e.idx; S.pnext[pe] = -1;
            if (S.qtail[lp] >= 0) S.pnext[S.qtail[lp]] = pe; else S.qhead[lp] = pe;
            S.qtail[lp] = pe;
            X.accepted[lp] = acc + 1; X.buf[lp] = buf + 1;
            if (buf == 0) xpush(S, xev(t, S.G++, XE_NOTIFY, lp));            // queue.py:144-146
        } break;
        case XE_NOTIFY:                                                      // QueueDriver._handle_notify (queue_driver.py:92-99)
            if (X.active[lp] < P.conc[lp]) xpush(S, xev(t, S.G++, XE_POLL, lp));
            break;
        case XE_POLL: {                                                      // Queue._handle_poll (queue.py:149-166)
            const int64_t buf = X.buf[lp];
            if (buf == 0) break;
            const int64_t k = X.accepted[lp] - buf;
            X.buf[lp] = buf - 1;
            const int32_t pe = S.qhead[lp];
            S.qhead[lp] = S.pnext[pe];
            if (S.qhead[lp] < 0) S.qtail[lp] = -1;
            xpush(S, xev(t, S.G++, XE_DELIVER, lp, k, pe));
        } break;
        case XE_DELIVER:                                                     // queue_driver.py:66-90: same payload, ORIGINAL index
            xpush(S, xev(t, S.pidx[e.aux], XE_WORK, lp, e.cr));
            break;
        case XE_WORK: {
            // Server.handle_queued_event up to its yield (server/server.py:202-250) via Event._start_process
            // (core/event.py:313-325): one continuation is built and invoked at once, the pushed one is the second
            X.started[lp] += 1;
            (void)S.G++;
            const int32_t active = X.active[lp];
            if (active >= P.conc[lp]) { X.rejected[lp] += 1; break; }        // acquire() failed (server.py:223-234)
            X.active[lp] = active + 1;
            double s; int64_t dur;
            const double mean = P.svc_mean[lp];
            if (P.svc_kind[lp] == 0) {
                const uint64_t k = X.svc_k[lp];
                const double u = xuniform(P.seed[lp], stream_id(P.stream_base[lp], kStreamService), k);
                X.svc_k[lp] = k + 1;
                const double sample = __ddiv_rn(exp1_from_uniform(u), __ddiv_rn(1.0, mean));   // expovariate(1 / mean)
                s = seconds_from_ns(ns_from_seconds(sample));
            } else s = seconds_from_ns(ns_from_seconds(mean));
            dur = ns_from_seconds(s);                                        // Instant + float seconds (temporal.py:222)
            int j = 0;
            for (int i = C - 1; i >= 0; --i) if (X.D[(size_t)i * N + lp] == kInfNs) j = i;
            const size_t sj = (size_t)j * N + lp;
            const int64_t k = e.cr;
            X.svc_s[sj] = s;
            X.crt[sj] = k < L.cap ? L.adm[log_at(L, k, lp, N)] : 0;
            const unsigned long long idx_c = S.G++;
            X.D[sj] = t + dur; X.seqD[sj] = (uint32_t)idx_c; X.crtD[sj] = t; X.dpD[sj] = lin_dp; X.rcD[sj] = S.ctx_rc;
            xpush(S, xev(t + dur, idx_c, XE_CONT, lp, 0, 0, (uint16_t)j));
        } break;
        case XE_CONT: {
            // the generator resumes (server/server.py:252-273): statistics, forward(event, downstream), then the
            // schedule_poll completion hook (queue_driver.py:79-84)
            const size_t sj = (size_t)(e.slot & 0xff) * N + lp;
            const double s = X.svc_s[sj];
            const int64_t cr = X.crt[sj];
            X.D[sj] = kInfNs;
            const int32_t active = X.active[lp] > 0 ? X.active[lp] - 1 : 0;
            X.active[lp] = active;
            X.completed[lp] += 1; n_completed++;
            X.total_service[lp] = __dadd_rn(X.total_service[lp], s);
            const uint32_t eg = net ? NP.egress[lp] : P.egress[lp];
            if (eg == EG_SINK) xpush(S, xev(t, S.G++, XE_SINK, lp, cr));
            else if (eg == EG_LINK) xpush(S, xev(t, S.G++, XE_LINK, lp, cr, NP.link_of[lp]));
            else if (eg == EG_ROUTER) xpush(S, xev(t, S.G++, XE_ROUTE, lp, cr));
            else if (eg == kEgressServer)                                    // forward(event, downstream) to another Server: its Request,
                xpush(S, xev(t, S.G++, XE_ENQ, P.tabs->tandem[(size_t)(kMaxUp + 1) * N + lp], cr));   // context preserved (core/entity.py:83-105)
            if (active < P.conc[lp]) xpush(S, xev(t, S.G++, XE_POLL, lp));
        } break;
        case XE_SINK: {                                                      // Sink.handle_event (components/common.py:36-44)
            const int64_t r = X.received[lp];
            if (r < L.cap) { L.sink_t[log_at(L, r, lp, N)] = t; L.sink_created[log_at(L, r, lp, N)] = e.cr; } else overflow |= 1;
            X.received[lp] = r + 1; X.sink_w[lp] = r + 1; n_received++;
        } break;
        case XE_ROUTE: {                                                     // RandomRouter.handle_event (random_router.py:32-45)
            NX.routed[lp] += 1;
            const uint64_t k = NX.route_k[lp];
            const double u = xuniform(P.seed[lp], stream_id(NP.route_base[lp], kStreamRoute), k);
            NX.route_k[lp] = k + 1;
            const int ri = (int)__dmul_rn(u, (double)NP.rt_cnt[lp]);         // targets[int(u * len(targets))]
            const int32_t target = ri == 0 ? NP.rt0[lp] : ri == 1 ? NP.rt1[lp] : ri == 2 ? NP.rt2[lp] : NP.rt3[lp];
            if (target < 0) xpush(S, xev(t, S.G++, XE_SINK, lp, e.cr));
            else xpush(S, xev(t, S.G++, XE_LINK, lp, e.cr, target));
        } break;
        case XE_LINK: {
            // NetworkLink.handle_event up to its yield (components/network/link.py:114-154, _calculate_delay :190-216)
            const int32_t l = (int32_t)e.aux;
            (void)S.G++;                                                     // the continuation built by _start_process
            const int64_t entered = NX.link_in[l];
            NX.link_in[l] = entered + 1;
            const double loss = NP.link_loss[l];
            if (loss > 1.0) { if (link_loses(NP, P.seed[lp], l, entered, t)) break; }    // (a table decides: PartitionLink.packet_loss)
            else if (loss > 0.0 &&
                xuniform(P.seed[lp], stream_id(NP.link_base[l], kStreamLoss), (uint64_t)entered) < loss) break;   // link.py:131-138
            NX.link_sent[l] += 1;
            double delay = seconds_from_ns(ns_from_seconds(NP.link_lat_min[l]));
            if (NP.link_jit_kind[l] == 0) {
                const uint64_t k = NX.link_k[l];
                const double u = xuniform(P.seed[lp], stream_id(NP.link_base[l], kStreamLink), k);
                NX.link_k[l] = k + 1;
                const double sample = __ddiv_rn(exp1_from_uniform(u), __ddiv_rn(1.0, NP.link_jit_mean[l]));
                delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(sample)));
            } else delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(NP.link_jit_mean[l])));   // ConstantLatency jitter (0: none)
            if (!(delay > 0.0)) delay = 0.0;
            xpush(S, xev(t + ns_from_seconds(delay), S.G++, XE_LINKCONT, NP.link_dst[l], e.cr, l, 0, t));
        } break;
        case XE_LINKCONT:                                                    // transit over (link.py:156-189): a NEW Event for the egress
            NX.link_packets[e.aux] += 1;
            xpush(S, xev(t, S.G++, XE_ENQ, lp, e.cr));
            break;
        case XE_PTICK: {
            // Source.handle_event with _ProbeEventProvider (instrumentation/probe.py:69-78): the daemon probe_event, then the next tick
            const int pj = e.slot & 0xff;
            const size_t o = (size_t)pj * N + lp;
            X.ev_probe[lp] += 1;
            const unsigned long long idx_pe = S.G++;
            const int64_t k2 = X.p_arr[o] + 1;                               // index of the Probe's next tick in its table
            const int64_t a2 = tick_lookup(P.tabs->times + (size_t)P.tabs->probe_row[o] * (size_t)P.tabs->cap, P.tabs->cap, k2, overflow);
            X.p_arr[o] = k2;
            xpush(S, xev(t, idx_pe, XE_PSAMPLE, lp, 0, 0, (uint16_t)pj));
            if (a2 != kInfNs) {
                const unsigned long long idx_t = S.G++;
                xpush(S, xev(a2, idx_t, XE_PTICK, lp, 0, 0, (uint16_t)pj));
                X.PA[o] = a2 < t ? kInfNs : a2; X.seqP[o] = (uint32_t)idx_t; X.crtP[o] = t; X.rcP[o] = S.ctx_rc;
            } else X.PA[o] = kInfNs;
        } break;
        case XE_PSAMPLE: {                                                   // measure_callback (probe.py:51-66)
            const int pj = e.slot & 0xff;
            const size_t o = (size_t)pj * N + lp;
            X.ev_probe[N + lp] += 1;
            int64_t v = 0;
            switch (P.probe_metric[o]) {
                case kProbeDepth: v = X.buf[lp]; break;
                case kProbeActive: v = X.active[lp]; break;
                case kProbeAccepted: v = X.accepted[lp]; break;
                case kProbeDropped: v = X.dropped[lp]; break;
                case kProbeCompleted: v = X.completed[lp]; break;
                case kProbeReceived: v = X.received[lp]; break;
                case kProbeGenerated: v = X.generated[lp]; break;
                default: break;
            }
            const int64_t pn = X.p_n[o];
            if (pn < L.pcap) { const size_t q = ((size_t)pj * (size_t)L.pcap + (size_t)pn) * N + lp; L.probe_t[q] = t; L.probe_v[q] = v; } else overflow |= 1;
            X.p_n[o] = pn + 1;
        } break;
        default: break;
        }
        if (S.G >= S.n_init && !I.no_handover) {
            if (S.tc == INT64_MAX) {                                         // once: run-time events that share a pre-run event's key
                int64_t tc = INT64_MIN;
                for (int64_t i = 0; i < S.heap_len; ++i) {
                    const XEvent &h = S.heap[i];
                    if (!(h.slot & kXInitFlag) && h.idx < S.n_init && S.init_t[h.idx] == h.t && h.t > tc) tc = h.t;
                }
                S.tc = tc;
            }
            if (cur >= S.tc && cur <= end_ns && (S.heap_len == 0 || S.heap[0].t > cur)) { handover = true; break; }
        }
    }
    for (int k = 0; k < 15; ++k) if (evk[k]) atomicAdd(&tot->ev[k], evk[k]);
    if (n_completed) atomicAdd(&tot->completed, n_completed);
    if (n_received) atomicAdd(&tot->received, n_received);
    if (overflow) atomicOr(&tot->overflow, overflow);
    if (S.err) atomicOr(&tot->overflow, 16);
    if (only >= 0) atomicMax(&tot->final_time, (long long)cur);
    else { tot->cur_time = cur; if (cur > tot->final_time) tot->final_time = cur; }
    (void)start_ns;
    return handover;
}

}  // namespace hs
