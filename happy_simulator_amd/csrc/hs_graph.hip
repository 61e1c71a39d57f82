// hs_graph.hip -- general entity graphs on ONE heap (include/hs_engine.h "General entity graphs", ABI 15, LoadBalancer nodes ABI 16).
//
// The station engines (hs_station.hpp / hs_netstation.hpp) get their speed from a fixed LP shape; what the same entity classes
// can be wired into beyond that shape -- links with several senders, routers with any fan-out that also target Servers and
// routers, Servers behind Servers next to links, any number of Sources per Server -- has no LP decomposition that the lineage
// key orders, and the reference's own definition of the order is the heap: (time, _sort_index) with the index taken from two
// global creation counters (core/event.py:53-77, core/event_heap.py:48).  So this file IS that loop, on the device:
//   * one lane of one wavefront pops and invokes events exactly like `Simulation._execute_until` (core/simulation.py:449-505);
//   * the binary heap follows CPython's heapq sift procedures step for step, so equal (time, index) keys -- possible between
//     the pre-run counter and the run-time counter -- pop in heapq's layout order;
//   * the first kLdsHeap entries of the heap (every level a sift touches first) live in LDS, the rest in HBM; node parameters
//     and state are two 64-byte rows per node (L2-resident for the models this path is for);
//   * heap, Request pool and the Sink record log grow on demand: the kernel stops in front of the event that would not fit,
//     the host enlarges the buffer and launches again (a launch also ends after kBudget events, so no launch runs for minutes).
// Handlers restate the reference handler by handler (citations at each); arithmetic through hs_device.hpp (the same fixed IEEE
// operation sequences as every other engine), streams through the Philox streams of DESIGN.md section 3.
//
// Cost: ~2.4 us per event (~1 000 vector instructions of ONE lane; measured: not memory).  An exactness path for the graphs the parallel
// engines refuse; its throughput comes from heaps side by side: replicas (hs_graph_run_many) and a Simulation's disconnected parts
// (hs_graph_run_parts).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hs_engine.h"
#include "hs_device.hpp"
#include "hs_tables_api.hpp"
#include "hs_ring.hpp"

namespace hs {
namespace graph {

constexpr int kBatchStat = 4;                  // int64 per heap a batch launch reports: status, processed, pending events, the earliest's time
constexpr int kLdsNodes = 192, kLdsNodesBatch = 48;   // nodes whose parameters and state stay in LDS for a launch (24 KB / 6 KB)
constexpr int kLdsHeapBatch = 1024;            // ... of each of the heaps hs_graph_run_many runs side by side
constexpr int kLdsHeap = 4096;                 // heap entries in LDS: 4 096 x 32 B = 128 KB of the CU's 160 (+ 24 KB of nodes)
constexpr long long kBudget = 1ll << 21;       // events per launch (~2 s)

struct GEvent {                                // 32 bytes
    int64_t t;                                 // Event.time
    uint64_t idx;                              // Event._sort_index
    int32_t node;                              // target entity
    int32_t req;                               // the Request it carries (-1: none)
    uint32_t kind;                             // HS_EV_*
    uint32_t pad;
};

struct GParam {                                // 64 bytes, read-only
    uint64_t stream_base;
    double mean;                               // Source: rate; Server: mean service; link: mean jitter
    double lat_min;                            // link: ConstantLatency base
    double loss;                               // link: packet_loss_rate
    int64_t lim;                               // Source: stop_after ns (< 0 never); Server: queue capacity (< 0 unbounded)
    int32_t target;
    int32_t conc;                              // Server: max_concurrent; Source: n_clients of its ClientKeyEventProvider (0: none);
                                               // LoadBalancer: entries of its client -> backend-slot table (lim = its offset)
    int32_t rt_off, rt_cnt;                    // router / LoadBalancer: its targets; Source / Probe: rt_off = row of its tick table (-1: none)
    uint8_t kind, sub;                         // sub: Source arrival kind (hs_source_kind); Server / link latency kind; Probe metric; LB strategy
    uint8_t pad[6];
};

struct GState {                                // 64 bytes
    // Source: a = ArrivalTimeProvider.current_time, b = arrival draws, c = generated_count, d = payload Requests built
    // Server: a = stats_accepted, b = stats_dropped, c = completed, d = rejected
    // link:   a = entered, b = packets_sent, c = packets_dropped, d = jitter draws
    // router: a = stats_routed (= route draws);  Sink: a = events_received
    // Probe:  a = ticks taken from its table, c = samples
    // LoadBalancer: a = requests_received, b = requests_forwarded, c = requests_failed (= no_backend_available), d = in flight,
    //               svc_draws = RoundRobin._index (ConsistentHash: its fallback's, the key-less Requests);  Source: svc_draws = KEY draws
    int64_t a, b, c, d;
    double total_service;                      // Server._total_service_time
    uint64_t svc_draws;
    int32_t qhead, qtail;                      // FIFOQueue (components/queue_policy.py:75-114) as a list through Request::next
    int32_t qlen, active;
};

struct GRequest {                              // 32 bytes: the payload Event's identity + its context
    int64_t created;                           // context["created_at"] (load/source.py:76-79; forwarded unchanged, core/entity.py:100-105)
    uint64_t idx;                              // sort index of the queued payload Event (kept on retarget, queue_driver.py:86-90)
    double service_s;                          // service_time_s of the generator frame (server/server.py:246-247)
    int64_t client;                            // context["metadata"]["client_id"] (-1: none)
    int32_t next;                              // FIFO / free list
    int32_t hook;                              // LoadBalancer whose `_lb_response` hook rides on the Event (-1: none)
};

enum : int { kRunning = 0, kDone = 1, kGrowHeap = 2, kGrowReq = 4, kGrowRec = 8, kBadKind = 16, kGrowTicks = 32, kUndecided = 64 };

struct GVars {                                 // device scalars
    long long heap_len;
    unsigned long long counter;                // the heap's own counter (run-time sort indices, core/event_heap.py:48)
    unsigned long long global_counter;         // the process-wide counter (pre-run events and schedule()d Events)
    long long cur;                             // Simulation._current_time
    long long processed;
    long long by_kind[HS_EV_KINDS];
    long long completed, received;
    long long rec_n;
    int req_len, req_free;
    int booted;
    int status;
    long long sched_done;                      // scheduled entries already pushed
    long long heap_peak;
};

struct GCtl {                                  // kernel argument
    GEvent *heap; long long heap_cap;
    GRequest *reqs; int req_cap;
    int32_t *rec_node; int64_t *rec_t, *rec_cr; long long rec_cap;
    const GParam *P; GState *S; int n;
    const int32_t *rt_targets; long long *rt_taken;
    const int32_t *sched_node; const int64_t *sched_t; long long n_sched;
    // tick tables (hs_tables.hpp) of the time-varying Sources and the Probes: tick k of row r at ticks[r * tick_cap + k]; a row
    // holds tick_count[r] ticks -- up to two beyond the horizon it was computed for, or up to the stream's end (kInfNs)
    const int64_t *ticks; long long tick_cap; const int64_t *tick_count;
    const int32_t *key_table;                  // ConsistentHash.select(str(client id)) as a backend slot, per LoadBalancer (GParam::lim)
    GVars *V;
    uint64_t seed;
    int64_t start_ns, end_ns;
    long long budget;
    int part;                                  // 1: this heap holds PART of a Simulation (hs_graph_run_parts)
};

__device__ __forceinline__ bool ev_lt(const GEvent &a, const GEvent &b) {   // Event.__lt__, core/event.py:337-344
    if (a.t != b.t) return a.t < b.t;
    return a.idx < b.idx;
}

// The window's entries are addressed as LDS (address space 3), two 16-byte words each: a pointer that may be LDS or HBM compiles to FLAT
// loads and stores, which wait on both memory counters (113 + 118 of them in the loop before: every level of a sift).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
template <int W>
struct Heap {
    lds_u32x4 *lds; GEvent *glob; long long len;
    __device__ __forceinline__ GEvent get(long long i) const {
        if (i < W) {
            const u32x4 a = lds[2 * i], b = lds[2 * i + 1];
            GEvent e;
            e.t = (int64_t)(((uint64_t)a.y << 32) | (uint64_t)a.x); e.idx = ((uint64_t)a.w << 32) | (uint64_t)a.z;
            e.node = (int32_t)b.x; e.req = (int32_t)b.y; e.kind = b.z; e.pad = b.w;
            return e;
        }
        return glob[i];
    }
    __device__ __forceinline__ void set(long long i, const GEvent &e) {
        if (i < W) {
            u32x4 a, b;
            a.x = (uint32_t)(uint64_t)e.t; a.y = (uint32_t)((uint64_t)e.t >> 32); a.z = (uint32_t)e.idx; a.w = (uint32_t)(e.idx >> 32);
            b.x = (uint32_t)e.node; b.y = (uint32_t)e.req; b.z = e.kind; b.w = e.pad;
            lds[2 * i] = a; lds[2 * i + 1] = b;
        } else glob[i] = e;
    }
    // heapq.heappush: append, then _siftdown(heap, 0, len - 1)
    __device__ inline void push(const GEvent &e) {
        long long pos = len++;
        while (pos > 0) {
            const long long parent = (pos - 1) >> 1;
            const GEvent p = get(parent);
            if (!ev_lt(e, p)) break;
            set(pos, p);
            pos = parent;
        }
        set(pos, e);
    }
    // heapq.heappop: the last leaf replaces the root; _siftup walks the hole down to a leaf along the smaller child (the right
    // one unless left < right), then _siftdown bubbles the moved item back up
    __device__ inline GEvent pop() {
        const GEvent top = get(0);
        const GEvent last = get(--len);
        const long long n = len;
        if (n == 0) return top;
        long long pos = 0, child = 1;
        while (child < n) {
            GEvent c = get(child);
            const long long right = child + 1;
            if (right < n) {
                const GEvent r = get(right);
                if (!ev_lt(c, r)) { child = right; c = r; }
            }
            set(pos, c);
            pos = child;
            child = 2 * pos + 1;
        }
        while (pos > 0) {
            const long long parent = (pos - 1) >> 1;
            const GEvent p = get(parent);
            if (!ev_lt(last, p)) break;
            set(pos, p);
            pos = parent;
        }
        set(pos, last);
        return top;
    }
};

__device__ __forceinline__ double uniform_at(uint64_t seed, uint64_t sid, uint64_t k) {
    Stream s;
    s.init(seed, sid, k);
    return s.next_uniform();
}

// a Request-carrying Event aimed at `node`: which handler it lands in (Entity.handle_event of that class)
__device__ __forceinline__ uint32_t arrival_kind(const GParam *P, int node) {
    switch (P[node].kind) {
        case HS_NODE_SERVER: return HS_EV_ENQUEUE;
        case HS_NODE_SINK: return HS_EV_SINK;
        case HS_NODE_LINK: return HS_EV_LINK;
        case HS_NODE_ROUTER: return HS_EV_ROUTE;
        case HS_NODE_LB: return HS_EV_LB;
        default: return 0xffffffffu;
    }
}

__device__ __forceinline__ GEvent mk(int64_t t, uint64_t idx, uint32_t kind, int node, int req) {
    GEvent e;
    e.t = t; e.idx = idx; e.node = node; e.req = req; e.kind = kind; e.pad = 0;
    return e;
}
// ... one constructed BEFORE the run: its index comes from the process-wide counter (core/event.py:53-67), the run's own events count
// from zero again -- GEvent::pad tells the two apart (part runs: a timestamp group that mixes them is order-sensitive to ALL of a
// Simulation's events, hs_graph_run_parts)
__device__ __forceinline__ GEvent mk_pre(int64_t t, uint64_t idx, uint32_t kind, int node, int req) {
    GEvent e = mk(t, idx, kind, node, req);
    e.pad = 1;
    return e;
}

// ArrivalTimeProvider.next_arrival_time, constant-rate fast path (load/arrival_time_provider.py:72-82) with the target area of
// poisson_arrival.py:31 / constant_arrival.py:23
__device__ inline int64_t next_arrival(const GCtl &c, int n) {
    const GParam &p = c.P[n];
    GState &s = c.S[n];
    if (p.rt_off >= 0) {                       // a time-varying profile: tick number `generated` of the Source's table
        const int64_t a2 = c.ticks[(size_t)p.rt_off * (size_t)c.tick_cap + (size_t)s.c];
        s.a = a2;
        return a2;
    }
    double area = 1.0;
    if (p.sub == HS_SRC_POISSON) {
        area = exp1_from_uniform(uniform_at(c.seed, stream_id(p.stream_base, kStreamArrival), (uint64_t)s.b));
        s.b += 1;
    }
    const int64_t a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(s.a), __ddiv_rn(area, p.mean)));
    s.a = a2;
    return a2;
}

// NL: graphs of up to NL nodes keep their nodes' parameters and state in LDS for the launch (the loop's dependent chain goes
// through them several times per event: 64-cycle LDS round trips instead of L2's) -- `lnodes`: NL x (GParam + GState).
template <int W, int NL>
__device__ __forceinline__ void graph_loop(const GCtl &c0, GEvent *lheap, char *lnodes) {
    __shared__ unsigned long long s_by_kind[HS_EV_KINDS];
    GCtl c = c0;
    GVars &V = *c.V;
    const int lane = threadIdx.x;
    const bool nodes_in_lds = c.n <= NL;
    if (nodes_in_lds) {
        const long long n8 = (long long)c.n * (long long)(sizeof(GParam) / 8);
        static_assert(sizeof(GParam) == sizeof(GState), "one copy loop for both");
        const uint64_t *sp = reinterpret_cast<const uint64_t *>(c0.P), *ss = reinterpret_cast<const uint64_t *>(c0.S);
        uint64_t *dp = reinterpret_cast<uint64_t *>(lnodes), *ds = reinterpret_cast<uint64_t *>(lnodes + (size_t)NL * sizeof(GParam));
        for (long long i = lane; i < n8; i += 64) { dp[i] = sp[i]; ds[i] = ss[i]; }
        c.P = reinterpret_cast<const GParam *>(lnodes);
        c.S = reinterpret_cast<GState *>(lnodes + (size_t)NL * sizeof(GParam));
    }
    if (lane < HS_EV_KINDS) s_by_kind[lane] = 0ull;
    {   // the heap's head comes into LDS (all 64 lanes copy; 8 bytes per lane and step)
        const long long n8 = (V.heap_len < W ? V.heap_len : (long long)W) * (long long)(sizeof(GEvent) / 8);
        const uint64_t *src = reinterpret_cast<const uint64_t *>(c.heap);
        uint64_t *dst = reinterpret_cast<uint64_t *>(lheap);
        for (long long i = lane; i < n8; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (lane == 0) {
        int req_free = V.req_free, req_len = V.req_len;                     // (registers for the launch)
        Heap<W> H{(lds_u32x4 *)lheap, c.heap, V.heap_len};
        unsigned long long G = V.counter;
        int status = kRunning;
        if (!V.booted) {
            // Simulation.__init__ (core/simulation.py:145-154) + Source.start (load/source.py:120-140): the Sources in list order,
            // their first SourceEvents numbered by the process-wide counter
            unsigned long long g = 0;
            for (int i = 0; i < c.n; ++i) {
                if (c.P[i].kind != HS_NODE_SOURCE) continue;
                c.S[i].a = c.start_ns;                          // provider.current_time = start_time
                const int64_t t = next_arrival(c, i);
                if (t == kInfNs) continue;                      // "Rate is zero indefinitely. Source will not start." (source.py:137-139)
                H.push(mk_pre(t, g++, HS_EV_SOURCE, i, -1));    // (heap_cap >= 4 n: hs_graph_create)
            }
            for (int i = 0; i < c.n; ++i) {                     // then the probes, in list order (core/simulation.py:156-160)
                if (c.P[i].kind != HS_NODE_PROBE) continue;
                const int64_t t = c.ticks[(size_t)c.P[i].rt_off * (size_t)c.tick_cap];
                if (t == kInfNs) continue;
                H.push(mk_pre(t, g++, HS_EV_PROBE_TICK, i, -1));
            }
            V.global_counter = g; V.booted = 1; G = 0; V.cur = c.start_ns;
        }
        // Simulation.schedule (core/simulation.py:195-206): Events constructed outside the run
        while (status == kRunning && V.sched_done < c.n_sched) {
            if (H.len + 1 > c.heap_cap) { status |= kGrowHeap; break; }
            int r = req_free;
            if (r >= 0) req_free = c.reqs[r].next;
            else if (req_len < c.req_cap) r = req_len++;
            else { status |= kGrowReq; break; }
            const int node = c.sched_node[V.sched_done];
            const int64_t t = c.sched_t[V.sched_done];
            GRequest q; q.created = t; q.idx = V.global_counter; q.service_s = 0.0; q.client = -1; q.next = -1; q.hook = -1;
            c.reqs[r] = q;
            H.push(mk_pre(t, V.global_counter++, arrival_kind(c.P, node), node, r));
            V.sched_done++;
        }
        long long cur = V.cur, processed = 0, n_completed = 0, n_received = 0, rec_n = V.rec_n;
        long long peak = V.heap_peak;
        while (status == kRunning) {
            // core/simulation.py:472 tests the PREVIOUS event's time; a PART of a Simulation stops in front of the first event beyond
            // the end (which of the parts' first events the reference still processes is decided across all of them)
            if (!(H.len > 0 && (c.part ? H.get(0).t <= c.end_ns : cur <= c.end_ns))) { status = kDone; break; }
            if (processed >= c.budget) break;
            // room for whatever this event constructs (at most two pushes, one Request, one record)
            if (H.len + 2 > c.heap_cap) { status |= kGrowHeap; break; }
            if (req_free < 0 && req_len >= c.req_cap) { status |= kGrowReq; break; }
            if (rec_n >= c.rec_cap) { status |= kGrowRec; break; }
            if (c.ticks != nullptr) {                                              // the next tick of a table-driven stream must be in its table
                const GEvent top = H.get(0);
                if (top.kind == HS_EV_SOURCE || top.kind == HS_EV_PROBE_TICK) {
                    const GParam &tp = c.P[top.node];
                    if (tp.rt_off >= 0) {
                        const int64_t need = (top.kind == HS_EV_SOURCE ? c.S[top.node].c : c.S[top.node].a) + 1;
                        const int64_t cnt = c.tick_count[tp.rt_off];
                        if (need >= cnt && c.ticks[(size_t)tp.rt_off * (size_t)c.tick_cap + (size_t)(cnt - 1)] != kInfNs && top.t >= cur) {
                            status |= kGrowTicks; break;
                        }
                    }
                }
            }
            if (H.len > peak) peak = H.len;
            const GEvent e = H.pop();
            if (c.part && H.len > 0) {
                // two PENDING events of one nanosecond, one numbered before the run and one by the run: which comes first depends on how
                // many events the WHOLE Simulation had created by then, not only this part -- undecided here.  (Every such pair meets as
                // (popped, next) at some pop: a timestamp group is popped back to back.)
                const GEvent nx = H.get(0);
                if (nx.t == e.t && ((nx.pad ^ e.pad) & 1u)) { status |= kUndecided; break; }
            }
            if (e.t < cur) continue;                                               // time-travel drop, core/simulation.py:480-489
            cur = e.t;
            processed++;
            const int n = e.node;
            const int64_t t = e.t;
            if (e.kind >= (uint32_t)HS_EV_KINDS) { status |= kBadKind; break; }
            s_by_kind[e.kind] += 1ull;
            const GParam p = c.P[n];
            GState &s = c.S[n];
            switch (e.kind) {
            case HS_EV_SOURCE: {
                // Source.handle_event (load/source.py:142-180): the payload is constructed first, then the next SourceEvent;
                // `return [*payload_events, next_tick]`
                const bool payload = !(p.lim >= 0 && t > p.lim);                    // SimpleEventProvider.get_events :68
                int r = -1;
                unsigned long long idx_p = 0;
                if (payload) {
                    r = req_free;
                    if (r >= 0) req_free = c.reqs[r].next; else r = req_len++;
                    idx_p = G++;
                    GRequest q; q.created = t; q.idx = idx_p; q.service_s = 0.0; q.client = -1; q.next = -1; q.hook = -1;
                    if (p.conc > 0) {                                               // chash_example.py:83: one client id per Request
                        const double u = uniform_at(c.seed, stream_id(p.stream_base, kStreamKey), s.svc_draws);
                        s.svc_draws += 1;
                        q.client = (int64_t)__dmul_rn(u, (double)p.conc);
                    }
                    c.reqs[r] = q;
                    s.d += 1;
                }
                s.c += 1;                                                           // _generated_count (:159)
                const int64_t a2 = next_arrival(c, n);
                if (payload) H.push(mk(t, idx_p, arrival_kind(c.P, p.target), p.target, r));
                if (a2 != kInfNs) H.push(mk(a2, G++, HS_EV_SOURCE, n, -1));       // RuntimeError: the source is exhausted (:176-180)
            } break;
            case HS_EV_ENQUEUE: {
                // QueuedResource.handle_event -> Queue._handle_enqueue (components/queue.py:122-147)
                const int hook = c.reqs[e.req].hook;                               // Event.on_complete of THIS Event: one-shot (core/event.py:290-311)
                c.reqs[e.req].hook = -1;
                if (p.lim >= 0 && s.qlen >= p.lim) {                               // FIFOQueue.push refuses (queue_policy.py:94-98)
                    s.b += 1;
                    c.reqs[e.req].next = req_free; req_free = e.req;
                } else {
                    c.reqs[e.req].idx = e.idx;              // the queued payload IS this Event object
                    c.reqs[e.req].next = -1;
                    if (s.qtail >= 0) c.reqs[s.qtail].next = e.req; else s.qhead = e.req;
                    s.qtail = e.req;
                    s.a += 1;
                    const bool was_empty = s.qlen == 0;
                    s.qlen += 1;
                    if (was_empty) H.push(mk(t, G++, HS_EV_NOTIFY, n, -1));        // queue.py:144-146
                }
                // Event.invoke (core/event.py:277-283): the handler returned a plain list, so the completion hooks run now, behind
                // the handler's own events: the LoadBalancer's `_lb_response`
                if (hook >= 0) H.push(mk(t, G++, HS_EV_LB_RESP, hook, -1));
            } break;
            case HS_EV_NOTIFY:                                                      // QueueDriver._handle_notify (queue_driver.py:92-99)
                if (s.active < p.conc) H.push(mk(t, G++, HS_EV_POLL, n, -1));
                break;
            case HS_EV_POLL: {                                                      // Queue._handle_poll (queue.py:149-166)
                if (s.qlen == 0) break;
                const int r = s.qhead;
                s.qhead = c.reqs[r].next;
                if (s.qhead < 0) s.qtail = -1;
                s.qlen -= 1;
                H.push(mk(t, G++, HS_EV_DELIVER, n, r));
            } break;
            case HS_EV_DELIVER:                                                     // queue_driver.py:66-90: same payload, ORIGINAL index
                H.push(mk(t, c.reqs[e.req].idx, HS_EV_WORK, n, e.req));
                break;
            case HS_EV_WORK: {
                // Server.handle_queued_event up to its yield (server/server.py:202-250) via Event._start_process
                // (core/event.py:313-325): one continuation is built and invoked at once, the pushed one is the second
                (void)G++;
                if (!(s.active < p.conc)) {                                         // acquire() failed (server.py:223-234)
                    s.d += 1;
                    c.reqs[e.req].next = req_free; req_free = e.req;
                    break;
                }
                s.active += 1;
                double sv;
                if (p.sub == HS_LAT_EXPONENTIAL) {
                    const double u = uniform_at(c.seed, stream_id(p.stream_base, kStreamService), s.svc_draws);
                    s.svc_draws += 1;
                    const double sample = __ddiv_rn(exp1_from_uniform(u), __ddiv_rn(1.0, p.mean));   // expovariate(1 / mean)
                    sv = seconds_from_ns(ns_from_seconds(sample));
                } else sv = seconds_from_ns(ns_from_seconds(p.mean));
                c.reqs[e.req].service_s = sv;
                H.push(mk(t + ns_from_seconds(sv), G++, HS_EV_CONTINUATION, n, e.req));   // Instant + float seconds (temporal.py:222)
            } break;
            case HS_EV_CONTINUATION: {
                // the generator resumes (server/server.py:252-273): statistics, forward(event, downstream), then the
                // schedule_poll completion hook (queue_driver.py:79-84)
                s.active = s.active > 0 ? s.active - 1 : 0;
                s.c += 1; n_completed++;
                s.total_service = __dadd_rn(s.total_service, c.reqs[e.req].service_s);
                if (p.target >= 0) H.push(mk(t, G++, arrival_kind(c.P, p.target), p.target, e.req));
                else { c.reqs[e.req].next = req_free; req_free = e.req; }
                if (s.active < p.conc) H.push(mk(t, G++, HS_EV_POLL, n, -1));
            } break;
            case HS_EV_SINK: {                                                      // Sink.handle_event (components/common.py:36-44)
                c.rec_node[rec_n] = n; c.rec_t[rec_n] = t; c.rec_cr[rec_n] = c.reqs[e.req].created;
                rec_n++;
                s.a += 1; n_received++;
                c.reqs[e.req].next = req_free; req_free = e.req;
            } break;
            case HS_EV_ROUTE: {                                                     // RandomRouter.handle_event (random_router.py:32-45)
                const double u = uniform_at(c.seed, stream_id(p.stream_base, kStreamRoute), (uint64_t)s.a);
                s.a += 1;
                const int ri = (int)__dmul_rn(u, (double)p.rt_cnt);                 // targets[int(u * len(targets))]
                const int tg = c.rt_targets[p.rt_off + ri];
                c.rt_taken[p.rt_off + ri] += 1;
                H.push(mk(t, G++, arrival_kind(c.P, tg), tg, e.req));
            } break;
            case HS_EV_LINK: {
                // NetworkLink.handle_event up to its yield (components/network/link.py:114-154, _calculate_delay :190-216)
                (void)G++;                                                          // the continuation built by _start_process
                const int64_t entered = s.a;
                s.a = entered + 1;
                if (p.loss > 0.0 && uniform_at(c.seed, stream_id(p.stream_base, kStreamLoss), (uint64_t)entered) < p.loss) {   // link.py:131-138
                    s.c += 1;
                    c.reqs[e.req].next = req_free; req_free = e.req;
                    break;
                }
                double delay = seconds_from_ns(ns_from_seconds(p.lat_min));
                if (p.sub == HS_LAT_EXPONENTIAL) {
                    const double u = uniform_at(c.seed, stream_id(p.stream_base, kStreamLink), (uint64_t)s.d);
                    s.d += 1;
                    const double sample = __ddiv_rn(exp1_from_uniform(u), __ddiv_rn(1.0, p.mean));
                    delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(sample)));
                } else if (p.mean > 0.0) delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(p.mean)));   // ConstantLatency jitter: no draw
                if (!(delay > 0.0)) delay = 0.0;                                    // max(0.0, delay)
                H.push(mk(t + ns_from_seconds(delay), G++, HS_EV_LINK_CONT, n, e.req));
            } break;
            case HS_EV_LINK_CONT:                                                   // transit over (link.py:156-189): a NEW Event for the egress
                s.b += 1;
                if (p.target >= 0) H.push(mk(t, G++, arrival_kind(c.P, p.target), p.target, e.req));
                else { c.reqs[e.req].next = req_free; req_free = e.req; }
                break;
            case HS_EV_LB: {
                // LoadBalancer._forward_request (load_balancer.py:347-433), every backend healthy
                s.a += 1;                                                           // :349
                if (p.rt_cnt == 0) {                                                // no healthy backends, :352-366
                    s.c += 1;
                    c.reqs[e.req].next = req_free; req_free = e.req;
                    break;
                }
                const int64_t client = c.reqs[e.req].client;
                int slot = 0;
                if (p.sub == HS_LB_CONSISTENT_HASH) {
                    if (client >= 0 && client < (int64_t)p.conc) slot = c.key_table[p.lim + client];   // ConsistentHash.select(str(client_id))
                    else {                                                          // no key: `self._fallback.select(...)`, a RoundRobin of
                        slot = (int)(s.svc_draws % (uint64_t)p.rt_cnt);             // the strategy's own (strategies.py:362,420-421)
                        s.svc_draws += 1;
                    }
                } else if (p.sub == HS_LB_ROUND_ROBIN) {
                    slot = (int)(s.svc_draws % (uint64_t)p.rt_cnt);                 // backends[_index % len], _index += 1 (strategies.py:66-67)
                    s.svc_draws += 1;
                } else if (client >= 0 && client < (int64_t)p.rt_cnt) slot = (int)client;            // Random: the plugged random.choice
                s.d += 1;                                                           // _in_flight, :378-382
                c.rt_taken[p.rt_off + slot] += 1;                                   // BackendInfo.total_requests, :385-386
                s.b += 1;                                                           // :388
                // a NEW Event for the backend with the same context (created_at survives) + the response hook (:398-431)
                const int be = c.rt_targets[p.rt_off + slot];
                c.reqs[e.req].hook = n;
                H.push(mk(t, G++, arrival_kind(c.P, be), be, e.req));
            } break;
            case HS_EV_LB_RESP:                                                     // LoadBalancer._handle_response (load_balancer.py:435-473)
                if (s.d > 0) s.d -= 1;
                break;
            case HS_EV_PROBE_TICK: {
                // Source.handle_event with _ProbeEventProvider (instrumentation/probe.py:69-78): the daemon probe_event, then the next tick
                const unsigned long long idx_pe = G++;
                const int64_t k2 = s.a + 1;
                const int64_t a2 = c.ticks[(size_t)p.rt_off * (size_t)c.tick_cap + (size_t)k2];
                s.a = k2;
                H.push(mk(t, idx_pe, HS_EV_PROBE, n, -1));
                if (a2 != kInfNs) H.push(mk(a2, G++, HS_EV_PROBE_TICK, n, -1));
            } break;
            case HS_EV_PROBE: {                                                     // measure_callback (probe.py:51-66)
                const GState &tg = c.S[p.target];
                int64_t v = 0;
                switch (p.sub) {
                    case HS_PROBE_DEPTH: v = tg.qlen; break;
                    case HS_PROBE_ACTIVE: v = tg.active; break;
                    case HS_PROBE_ACCEPTED: v = tg.a; break;
                    case HS_PROBE_DROPPED: v = tg.b; break;
                    case HS_PROBE_COMPLETED: v = tg.c; break;
                    case HS_PROBE_RECEIVED: v = tg.a; break;                        // Sink.events_received
                    case HS_PROBE_GENERATED: v = tg.c; break;                       // Source.generated_count
                    default: break;
                }
                c.rec_node[rec_n] = n; c.rec_t[rec_n] = t; c.rec_cr[rec_n] = v;
                rec_n++;
                s.c += 1;
            } break;
            default: status |= kBadKind; break;
            }
        }
        V.heap_len = H.len; V.counter = G; V.cur = cur; V.processed += processed; V.rec_n = rec_n;
        for (int k = 0; k < HS_EV_KINDS; ++k) V.by_kind[k] += (long long)s_by_kind[k];
        V.req_free = req_free; V.req_len = req_len;
        V.completed += n_completed; V.received += n_received;
        V.heap_peak = peak;
        V.status = status;
    }
    __syncthreads();
    {   // ... and back
        const long long n8 = (V.heap_len < W ? V.heap_len : (long long)W) * (long long)(sizeof(GEvent) / 8);
        uint64_t *dst = reinterpret_cast<uint64_t *>(c.heap);
        const uint64_t *src = reinterpret_cast<const uint64_t *>(lheap);
        for (long long i = lane; i < n8; i += 64) dst[i] = src[i];
    }
    if (nodes_in_lds) {   // the nodes' state returns to its place
        const long long n8 = (long long)c.n * (long long)(sizeof(GState) / 8);
        const uint64_t *ss = reinterpret_cast<const uint64_t *>(lnodes + (size_t)NL * sizeof(GParam));
        uint64_t *ds = reinterpret_cast<uint64_t *>(c0.S);
        for (long long i = lane; i < n8; i += 64) ds[i] = ss[i];
    }
}

__global__ void __launch_bounds__(64) hs_graph_run(GCtl c) {
    __shared__ GEvent lheap[kLdsHeap];
    __shared__ __attribute__((aligned(16))) char lnodes[kLdsNodes * (sizeof(GParam) + sizeof(GState))];
    graph_loop<kLdsHeap, kLdsNodes>(c, lheap, lnodes);
}

// Independent graphs -- the replicas / sweep points of parallel/runner.py:82-142 -- side by side: one workgroup (one heap) each, one
// with a quarter of the lone run's LDS window (32 KB + 6 KB of nodes: four workgroups per CU, 1 024 heaps on the device at once; a heap that outgrows
// the window continues in HBM as it does behind the large one -- the window's size changes nothing the loop computes).
__global__ void __launch_bounds__(64) hs_graph_run_batch(const GCtl *cs, long long *stat) {
    __shared__ GEvent lheap[kLdsHeapBatch];
    __shared__ __attribute__((aligned(16))) char lnodes[kLdsNodesBatch * (sizeof(GParam) + sizeof(GState))];
    const GCtl c = cs[blockIdx.x];
    graph_loop<kLdsHeapBatch, kLdsNodesBatch>(c, lheap, lnodes);
    __syncthreads();
    if (threadIdx.x == 0) {                    // what the host decides on, in ONE array for the whole batch
        long long *o = stat + kBatchStat * blockIdx.x;
        o[0] = c.V->status; o[1] = c.V->processed; o[2] = c.V->heap_len;
        o[3] = c.V->heap_len > 0 ? c.heap[0].t : 0;                // (the earliest pending event: hs_graph_run_parts elects across parts)
    }
}

}  // namespace graph
}  // namespace hs

using namespace hs::graph;

struct hs_graph {
    hs_graph_config cfg{};
    int n = 0, n_rt = 0;
    GCtl ctl{};
    std::vector<GParam> params;
    std::vector<int32_t> sched_node; std::vector<int64_t> sched_t;
    int32_t *d_sched_node = nullptr; int64_t *d_sched_t = nullptr; long long d_sched_cap = 0;
    int32_t *d_rt_targets = nullptr;
    int32_t *d_key_table = nullptr;
    // tick tables: one row per time-varying Source and per distinct Probe interval, computed up to `tick_horizon`
    std::vector<hs::TickRow> rows;
    std::vector<double> row_rate;              // ticks per second a row may reach (sizes the table)
    hs::TickRow *d_rows = nullptr; int64_t *d_ticks = nullptr, *d_tick_count = nullptr; unsigned long long *d_tick_status = nullptr;
    int64_t tick_horizon = INT64_MIN, tick_cap = 0;
    hipStream_t stream = nullptr;              // created with the first launch that needs one (a replica run in a batch never does)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    char *slab = nullptr; size_t slab_bytes = 0, slab_alloc = 0;   // every buffer of hs_graph_create in ONE allocation (a replica costs one hipMalloc / hipFree)
    double last_run_ms = 0.0;
    long long launches = 0;
    bool ran = false;
    long long pending_events = 0; int64_t earliest_ns = 0;   // after a batch launch: the heap's length and its top's time
    bool undecided = false;                    // a part run met a timestamp group only the whole Simulation orders (hs_graph_run_parts)
    std::string error;
};

static thread_local std::string g_graph_error;

static int gfail(hs_graph *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf;
    g_graph_error = buf;
    return code;
}

#define HSG_HIP(g, expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return gfail(g, HS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

// Slabs of destroyed handles wait here for the next handle of about their size: a Simulation spread over thousands of parts, or a sweep of
// thousands of replicas, creates and frees as many handles back to back, and hipFree synchronises the device every time (0.2 ms).
// Bounded: 4 096 slabs / 4 GB; the rest goes back to the runtime.
struct SlabPool {
    std::mutex mu;
    std::vector<std::pair<size_t, char *>> free;          // (bytes, pointer) of ONE device (the pool serves the device of its first slab)
    size_t bytes = 0; int device = -1;
    char *take(int dev, size_t want, size_t *have) {
        std::lock_guard<std::mutex> lk(mu);
        if (dev != device) return nullptr;
        for (size_t i = free.size(); i-- > 0;) {
            if (free[i].first >= want && free[i].first <= want + want / 2 + 4096) {
                char *p = free[i].second; *have = free[i].first; bytes -= free[i].first;
                free[i] = free.back(); free.pop_back();
                return p;
            }
            if (free.size() - i > 64) break;              // (the newest 64: a sweep's handles are all of one size)
        }
        return nullptr;
    }
    bool give(int dev, size_t have, char *p) {
        std::lock_guard<std::mutex> lk(mu);
        if (device < 0) device = dev;
        if (dev != device || free.size() >= 4096 || bytes + have > (4ull << 30)) return false;
        free.emplace_back(have, p); bytes += have;
        return true;
    }
};
static SlabPool g_slabs;

static bool in_slab(const hs_graph *g, const void *p) {
    return g->slab && (const char *)p >= g->slab && (const char *)p < g->slab + g->slab_bytes;
}

template <typename T>
static int grow(hs_graph *g, T **buf, long long old_n, long long new_n) {
    T *nb = nullptr;
    HSG_HIP(g, hipMalloc(&nb, (size_t)new_n * sizeof(T)));
    if (*buf && old_n > 0) HSG_HIP(g, hipMemcpy(nb, *buf, (size_t)old_n * sizeof(T), hipMemcpyDeviceToDevice));
    if (*buf && !in_slab(g, *buf)) HSG_HIP(g, hipFree(*buf));
    *buf = nb;
    return HS_OK;
}

static int ensure_stream(hs_graph *g) {
    if (g->stream) return HS_OK;
    HSG_HIP(g, hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HSG_HIP(g, hipEventCreate(&g->ev_a));
    HSG_HIP(g, hipEventCreate(&g->ev_b));
    return HS_OK;
}

extern "C" {

const char *hs_graph_last_error(const hs_graph *g) { return g ? g->error.c_str() : g_graph_error.c_str(); }

void hs_graph_destroy(hs_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->cfg.device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    void *bufs[] = {g->ctl.heap, g->ctl.reqs, g->ctl.rec_node, g->ctl.rec_t, g->ctl.rec_cr, (void *)g->ctl.P, g->ctl.S,
                    g->d_rt_targets, g->d_key_table, g->ctl.rt_taken, g->d_sched_node, g->d_sched_t, g->ctl.V, g->d_rows, g->d_ticks, g->d_tick_count,
                    g->d_tick_status};
    for (void *b : bufs) if (b && !in_slab(g, b)) (void)hipFree(b);
    if (g->slab && !g_slabs.give(g->cfg.device, g->slab_alloc, g->slab)) (void)hipFree(g->slab);
    if (g->ev_a) (void)hipEventDestroy(g->ev_a);
    if (g->ev_b) (void)hipEventDestroy(g->ev_b);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

int hs_graph_create(const hs_graph_config *cfg, const hs_graph_nodes *nd, hs_graph **out) {
    if (!cfg || !nd || !out) return gfail(nullptr, HS_E_INVALID, "hs_graph_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(hs_graph_config)) return gfail(nullptr, HS_E_INVALID, "hs_graph_config.struct_size mismatch (ABI)");
    const int n = nd->n_nodes;
    if (n < 1) return gfail(nullptr, HS_E_INVALID, "n_nodes must be >= 1");
    if (!nd->kind || !nd->target) return gfail(nullptr, HS_E_INVALID, "kind and target are required");
    if (nd->n_rt < 0 || (nd->n_rt > 0 && !nd->rt_targets)) return gfail(nullptr, HS_E_INVALID, "rt_targets is required with n_rt > 0");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count < 1)
        return gfail(nullptr, HS_E_NO_DEVICE, "no HIP device is visible (the engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= dev_count) return gfail(nullptr, HS_E_INVALID, "device %d out of range", cfg->device);
    auto takes_requests = [&](int t) { const int k = nd->kind[t]; return k == HS_NODE_SERVER || k == HS_NODE_SINK || k == HS_NODE_LINK || k == HS_NODE_ROUTER || k == HS_NODE_LB; };
    int64_t kmax = 0;                                  // client ids any Source hands out: [0, kmax)
    for (int i = 0; i < n; ++i)
        if (nd->kind[i] == HS_NODE_SOURCE && nd->src_n_clients && nd->src_n_clients[i] > kmax) kmax = nd->src_n_clients[i];
    if (kmax > (1ll << 26)) return gfail(nullptr, HS_E_UNSUPPORTED, "n_clients above 2^26 is not supported (client -> backend table)");
    std::vector<int32_t> key_table;
    std::vector<GParam> P((size_t)n);
    std::vector<hs::TickRow> rows;
    std::vector<double> row_rate;
    double rate_sum = 0.0;
    int n_src = 0;
    bool seen_other = false, seen_probe = false;      // node order: Sources, then Probes (the pre-run sort indices), then the rest
    for (int i = 0; i < n; ++i) {
        GParam p{};
        p.kind = nd->kind[i];
        p.target = nd->target[i];
        p.stream_base = nd->stream_base ? nd->stream_base[i] : (uint64_t)i;
        p.conc = 1; p.lim = -1; p.rt_off = 0; p.rt_cnt = 0; p.mean = 0.0; p.lat_min = 0.0; p.loss = 0.0;
        p.rt_off = -1;
        if (p.target < -1 || p.target >= n) return gfail(nullptr, HS_E_INVALID, "node %d: target %d out of range", i, p.target);
        if (p.kind != HS_NODE_PROBE && p.target >= 0 && !takes_requests(p.target))
            return gfail(nullptr, HS_E_INVALID, "node %d: its target %d takes no Requests (a Source or a Probe)", i, p.target);
        if (p.kind == HS_NODE_SOURCE && (seen_other || seen_probe))
            return gfail(nullptr, HS_E_INVALID, "node %d: SOURCE nodes come first, in sources=[...] order", i);
        if (p.kind == HS_NODE_PROBE && seen_other)
            return gfail(nullptr, HS_E_INVALID, "node %d: PROBE nodes come behind the SOURCE nodes, in probes=[...] order", i);
        if (p.kind == HS_NODE_PROBE) seen_probe = true;
        else if (p.kind != HS_NODE_SOURCE) seen_other = true;
        switch (p.kind) {
        case HS_NODE_SOURCE: {
            if (!nd->src_rate) return gfail(nullptr, HS_E_INVALID, "src_rate is required");
            p.sub = nd->src_kind ? nd->src_kind[i] : (uint8_t)HS_SRC_POISSON;
            if (p.sub != HS_SRC_POISSON && p.sub != HS_SRC_CONSTANT) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: source kind %d is not lowered", i, (int)p.sub);
            p.mean = nd->src_rate[i];
            if (!(p.mean > 0.0) || !std::isfinite(p.mean)) return gfail(nullptr, HS_E_INVALID, "node %d: source rate must be > 0, got %g", i, p.mean);
            p.lim = nd->src_stop_after_ns ? nd->src_stop_after_ns[i] : -1;
            if (p.target < 0) return gfail(nullptr, HS_E_INVALID, "node %d: a Source needs a target", i);
            {
                const int64_t nc = nd->src_n_clients ? nd->src_n_clients[i] : 0;
                if (nc < 0) return gfail(nullptr, HS_E_INVALID, "node %d: src_n_clients < 0", i);
                p.conc = (int32_t)nc;
            }
            rate_sum += p.mean;
            ++n_src;
            const int pk = nd->src_profile_kind ? nd->src_profile_kind[i] : 0;
            if (pk != HS_PROF_CONSTANT) {
                if (pk != HS_PROF_LINEAR_RAMP && pk != HS_PROF_SPIKE) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: profile kind %d is not lowered", i, pk);
                if (!nd->src_profile_params) return gfail(nullptr, HS_E_INVALID, "src_profile_params is required with src_profile_kind");
                const double *q = nd->src_profile_params + 4 * (size_t)i;
                for (int j = 0; j < 4; ++j) if (!std::isfinite(q[j]) || q[j] < 0.0) return gfail(nullptr, HS_E_INVALID, "node %d: bad profile parameter %g", i, q[j]);
                hs::TickRow r{};
                r.kind = (uint32_t)pk; r.poisson = p.sub == HS_SRC_POISSON ? 1u : 0u;
                r.p0 = q[0]; r.p1 = q[1]; r.p2 = q[2]; r.p3 = q[3];
                r.seed = cfg->seed; r.sid = hs::stream_id(p.stream_base, hs::kStreamArrival); r.owner = i;
                p.rt_off = (int32_t)rows.size();
                rows.push_back(r);
                row_rate.push_back(p.mean);
            }
        } break;
        case HS_NODE_PROBE: {
            if (!nd->probe_metric || !nd->probe_interval_s) return gfail(nullptr, HS_E_INVALID, "probe_metric and probe_interval_s are required for probes");
            p.sub = nd->probe_metric[i];
            const double iv = nd->probe_interval_s[i];
            if (!(iv > 0.0) || !std::isfinite(iv)) return gfail(nullptr, HS_E_INVALID, "node %d: Probe interval must be positive", i);   // probe.py:29-30
            if (p.target < 0) return gfail(nullptr, HS_E_INVALID, "node %d: a Probe needs a target", i);
            const int tk = nd->kind[p.target];
            const bool ok = p.sub == HS_PROBE_GENERATED ? tk == HS_NODE_SOURCE : p.sub == HS_PROBE_RECEIVED ? tk == HS_NODE_SINK
                            : (p.sub <= HS_PROBE_COMPLETED && tk == HS_NODE_SERVER);
            if (!ok) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: metric %d is not an attribute of its target (node kind %d)", i, (int)p.sub, tk);
            const double rate = 1.0 / iv;                           // _ProbeProfile.rate (probe.py:31)
            int found = -1;
            for (size_t q = 0; q < rows.size() && found < 0; ++q)
                if (rows[q].kind == hs::kProfGeneralConstant && rows[q].p0 == rate) found = (int)q;
            if (found < 0) {
                hs::TickRow r{};
                r.kind = hs::kProfGeneralConstant; r.poisson = 0; r.p0 = rate; r.owner = i;
                found = (int)rows.size();
                rows.push_back(r);
                row_rate.push_back(rate);
            }
            p.rt_off = found;
        } break;
        case HS_NODE_SERVER: {
            p.conc = nd->concurrency ? nd->concurrency[i] : 1;
            if (p.conc < 1) return gfail(nullptr, HS_E_INVALID, "node %d: max_concurrent must be >= 1, got %d", i, p.conc);
            p.sub = nd->lat_kind ? nd->lat_kind[i] : (uint8_t)HS_LAT_CONSTANT;
            if (p.sub != HS_LAT_EXPONENTIAL && p.sub != HS_LAT_CONSTANT) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: service distribution kind %d is not lowered", i, (int)p.sub);
            p.mean = nd->lat_mean_s ? nd->lat_mean_s[i] : 0.0;
            if (!(p.mean >= 0.0) || !std::isfinite(p.mean)) return gfail(nullptr, HS_E_INVALID, "node %d: bad service mean %g", i, p.mean);
            if (p.sub == HS_LAT_EXPONENTIAL && !(p.mean > 0.0)) return gfail(nullptr, HS_E_INVALID, "node %d: exponential service needs mean > 0", i);
            p.lim = nd->queue_cap ? nd->queue_cap[i] : -1;
        } break;
        case HS_NODE_SINK: break;
        case HS_NODE_LINK: {
            p.sub = nd->lat_kind ? nd->lat_kind[i] : (uint8_t)HS_LAT_CONSTANT;
            if (p.sub != HS_LAT_EXPONENTIAL && p.sub != HS_LAT_CONSTANT) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: jitter kind %d is not lowered", i, (int)p.sub);
            p.mean = nd->lat_mean_s ? nd->lat_mean_s[i] : 0.0;
            if (p.sub == HS_LAT_EXPONENTIAL && !(p.mean > 0.0)) return gfail(nullptr, HS_E_INVALID, "node %d: exponential jitter needs mean > 0", i);
            if (!(p.mean >= 0.0) || !std::isfinite(p.mean)) return gfail(nullptr, HS_E_INVALID, "node %d: bad jitter mean %g", i, p.mean);
            p.lat_min = nd->link_lat_min_s ? nd->link_lat_min_s[i] : 0.0;
            if (!(p.lat_min >= 0.0) || !std::isfinite(p.lat_min)) return gfail(nullptr, HS_E_INVALID, "node %d: bad link latency %g", i, p.lat_min);
            p.loss = nd->link_loss_rate ? nd->link_loss_rate[i] : 0.0;
            if (!(p.loss >= 0.0 && p.loss <= 1.0)) return gfail(nullptr, HS_E_INVALID, "node %d: packet_loss_rate must be in [0, 1], got %g", i, p.loss);   // link.py:71-72
        } break;
        case HS_NODE_ROUTER: {
            if (!nd->rt_off || !nd->rt_cnt) return gfail(nullptr, HS_E_INVALID, "rt_off and rt_cnt are required for routers");
            p.rt_off = nd->rt_off[i]; p.rt_cnt = nd->rt_cnt[i];
            if (p.rt_cnt < 1) return gfail(nullptr, HS_E_INVALID, "node %d: a RandomRouter needs at least one target", i);
            if (p.rt_off < 0 || (long long)p.rt_off + p.rt_cnt > nd->n_rt) return gfail(nullptr, HS_E_INVALID, "node %d: router targets out of range", i);
            for (int q = 0; q < p.rt_cnt; ++q) {
                const int t = nd->rt_targets[p.rt_off + q];
                if (t < 0 || t >= n || !takes_requests(t)) return gfail(nullptr, HS_E_INVALID, "node %d: router target %d takes no Requests", i, t);
            }
        } break;
        case HS_NODE_LB: {
            if (!nd->rt_off || !nd->rt_cnt) return gfail(nullptr, HS_E_INVALID, "rt_off and rt_cnt are required for LoadBalancers");
            p.rt_off = nd->rt_off[i]; p.rt_cnt = nd->rt_cnt[i];
            if (p.rt_cnt < 0 || p.rt_off < 0 || (long long)p.rt_off + p.rt_cnt > nd->n_rt) return gfail(nullptr, HS_E_INVALID, "node %d: backends out of range", i);
            p.sub = nd->lb_strategy ? nd->lb_strategy[i] : (uint8_t)HS_LB_ROUND_ROBIN;                  // load_balancer.py:112: the default
            if (p.sub > HS_LB_RANDOM) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: load-balancing strategy %d is not lowered", i, (int)p.sub);
            for (int q = 0; q < p.rt_cnt; ++q) {
                const int t = nd->rt_targets[p.rt_off + q];
                if (t < 0 || t >= n || nd->kind[t] != HS_NODE_SERVER) return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: backend %d is not a Server (only Server backends are lowered)", i, t);
            }
            p.lim = -1; p.conc = 0;
            if (p.sub == HS_LB_CONSISTENT_HASH && p.rt_cnt > 0) {
                const int V = nd->lb_vnodes ? nd->lb_vnodes[i] : 100;
                if (V < 1) return gfail(nullptr, HS_E_INVALID, "node %d: virtual_nodes must be >= 1, got %d", i, V);   // strategies.py:355-356
                if (!nd->names || !nd->name_off) return gfail(nullptr, HS_E_INVALID, "names / name_off are required for a ConsistentHash LoadBalancer");
                std::string names;
                std::vector<int32_t> off(1, 0);
                for (int q = 0; q < p.rt_cnt; ++q) {
                    const int t = nd->rt_targets[p.rt_off + q];
                    const int nl = nd->name_off[t + 1] - nd->name_off[t];
                    if (nl < 1 || nl > 200) return gfail(nullptr, HS_E_INVALID, "node %d: backend %d needs a name (1 .. 200 bytes)", i, t);
                    names.append(nd->names + nd->name_off[t], (size_t)nl);
                    off.push_back((int32_t)names.size());
                }
                const std::vector<hs::ring::RingPoint> ring = hs::ring::build_ring(names.data(), off.data(), p.rt_cnt, V);
                p.lim = (int64_t)key_table.size(); p.conc = (int32_t)kmax;
                char key[32];
                for (int64_t cid = 0; cid < kmax; ++cid) {                          // ConsistentHash.select for key str(id): a pure function of the id
                    const int len = snprintf(key, sizeof key, "%lld", (long long)cid);
                    key_table.push_back(hs::ring::ring_select(ring, key, (size_t)len));
                }
            }
        } break;
        default: return gfail(nullptr, HS_E_UNSUPPORTED, "node %d: kind %d is not lowered", i, (int)p.kind);
        }
        P[(size_t)i] = p;
    }
    hs_graph *g = new hs_graph();
    g->cfg = *cfg; g->n = n; g->n_rt = nd->n_rt; g->params = P; g->rows = rows; g->row_rate = row_rate;
#define HSG_TRY(expr) do { int rc_ = (expr); if (rc_) { g_graph_error = g->error; hs_graph_destroy(g); return rc_; } } while (0)
#define HSG_HIPD(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { gfail(g, HS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); g_graph_error = g->error; hs_graph_destroy(g); return HS_E_HIP; } } while (0)
    HSG_HIPD(hipSetDevice(cfg->device));
    GCtl &c = g->ctl;
    c.n = n; c.seed = cfg->seed; c.start_ns = cfg->start_ns; c.budget = kBudget;
    c.heap_cap = std::max<long long>(cfg->heap_capacity > 0 ? cfg->heap_capacity : 0, (long long)n * 4 + 1024);
    c.req_cap = (int)std::min<long long>(std::max<long long>(cfg->request_capacity > 0 ? cfg->request_capacity : 0, (long long)n * 4 + 1024), 1ll << 30);
    c.rec_cap = cfg->record_capacity > 0 ? cfg->record_capacity : 65536;
    (void)rate_sum;
    // one allocation: the initialised arrays first (one copy from a host image), behind them the ones the run fills (a buffer that
    // has to grow later moves into an allocation of its own, grow())
    const size_t nrt = (size_t)(nd->n_rt > 0 ? nd->n_rt : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_P = take((size_t)n * sizeof(GParam)), o_S = take((size_t)n * sizeof(GState)), o_rt = take(nrt * sizeof(int32_t)),
                 o_key = take(key_table.size() * sizeof(int32_t)), o_taken = take(nrt * sizeof(long long)),
                 o_rows = take(rows.size() * sizeof(hs::TickRow)), o_V = take(sizeof(GVars));
    const size_t image_bytes = off;
    const size_t o_tcount = take(rows.size() * sizeof(int64_t)), o_tstatus = take(rows.empty() ? 0 : 2 * sizeof(unsigned long long)),
                 o_heap = take((size_t)c.heap_cap * sizeof(GEvent)), o_reqs = take((size_t)c.req_cap * sizeof(GRequest)),
                 o_rnode = take((size_t)c.rec_cap * sizeof(int32_t)), o_rt_ns = take((size_t)c.rec_cap * sizeof(int64_t)),
                 o_rcr = take((size_t)c.rec_cap * sizeof(int64_t));
    g->slab_bytes = off;
    g->slab = g_slabs.take(cfg->device, off, &g->slab_alloc);   // (a destroyed handle's slab of about this size, else a new one)
    if (!g->slab) { HSG_HIPD(hipMalloc((void **)&g->slab, g->slab_bytes)); g->slab_alloc = g->slab_bytes; }
    {
        std::vector<char> image(image_bytes, 0);
        std::memcpy(image.data() + o_P, P.data(), (size_t)n * sizeof(GParam));
        GState *S0 = reinterpret_cast<GState *>(image.data() + o_S);
        for (int i = 0; i < n; ++i) { S0[i].qhead = -1; S0[i].qtail = -1; }
        if (nd->n_rt > 0) std::memcpy(image.data() + o_rt, nd->rt_targets, (size_t)nd->n_rt * sizeof(int32_t));
        if (!key_table.empty()) std::memcpy(image.data() + o_key, key_table.data(), key_table.size() * sizeof(int32_t));
        if (!rows.empty()) std::memcpy(image.data() + o_rows, rows.data(), rows.size() * sizeof(hs::TickRow));
        GVars v{};
        v.req_free = -1; v.cur = cfg->start_ns;
        std::memcpy(image.data() + o_V, &v, sizeof v);
        HSG_HIPD(hipMemcpy(g->slab, image.data(), image_bytes, hipMemcpyHostToDevice));
    }
    c.P = reinterpret_cast<const GParam *>(g->slab + o_P);
    c.S = reinterpret_cast<GState *>(g->slab + o_S);
    g->d_rt_targets = reinterpret_cast<int32_t *>(g->slab + o_rt);
    c.rt_targets = g->d_rt_targets;
    g->d_key_table = key_table.empty() ? nullptr : reinterpret_cast<int32_t *>(g->slab + o_key);
    c.key_table = g->d_key_table;
    c.rt_taken = reinterpret_cast<long long *>(g->slab + o_taken);
    c.V = reinterpret_cast<GVars *>(g->slab + o_V);
    if (!rows.empty()) {
        g->d_rows = reinterpret_cast<hs::TickRow *>(g->slab + o_rows);
        g->d_tick_count = reinterpret_cast<int64_t *>(g->slab + o_tcount);
        g->d_tick_status = reinterpret_cast<unsigned long long *>(g->slab + o_tstatus);
    }
    c.heap = reinterpret_cast<GEvent *>(g->slab + o_heap);
    c.reqs = reinterpret_cast<GRequest *>(g->slab + o_reqs);
    c.rec_node = reinterpret_cast<int32_t *>(g->slab + o_rnode);
    c.rec_t = reinterpret_cast<int64_t *>(g->slab + o_rt_ns);
    c.rec_cr = reinterpret_cast<int64_t *>(g->slab + o_rcr);
    HSG_HIPD(hipDeviceSynchronize());
#undef HSG_TRY
#undef HSG_HIPD
    *out = g;
    return HS_OK;
}

int hs_graph_schedule(hs_graph *g, int32_t node, int64_t time_ns) {
    if (!g) return gfail(g, HS_E_INVALID, "null handle");
    if (node < 0 || node >= g->n) return gfail(g, HS_E_INVALID, "schedule: node %d out of range", node);
    if (g->params[(size_t)node].kind == HS_NODE_SOURCE)
        return gfail(g, HS_E_UNSUPPORTED, "schedule: node %d is a Source (only Requests for a Server, Sink, link or router are lowered)", node);
    g->sched_node.push_back(node);
    g->sched_t.push_back(time_ns);
    return HS_OK;
}

}  // extern "C"

// Tick tables up to `horizon` (two ticks beyond it, hs_tables.hip); the same prefix whatever the horizon, so a later, longer table
// continues the run that used the shorter one.
static int build_tables(hs_graph *g, int64_t horizon) {
    if (g->rows.empty() || horizon <= g->tick_horizon) return HS_OK;
    const double span_s = (double)(horizon - g->cfg.start_ns) / 1e9;
    int64_t cap = g->tick_cap > 0 ? g->tick_cap : 0;
    for (double r : g->row_rate) {
        const double mean = r * (span_s > 0 ? span_s : 0.0);
        const double want = mean + 10.0 * std::sqrt(mean + 1.0) + 72.0;
        if (want > 4e9) return gfail(g, HS_E_INVALID, "a tick table up to %lld ns would need %.3g ticks", (long long)horizon, want);
        if ((int64_t)want > cap) cap = (int64_t)want;
    }
    const long long budget = g->cfg.profile_budget > 0 ? g->cfg.profile_budget : hs::kDefaultLaneBudget;
    for (int attempt = 0; attempt < 8; ++attempt) {
        if ((double)g->rows.size() * (double)cap * 8.0 > 64e9) return gfail(g, HS_E_INVALID, "tick tables would need %.1f GB", (double)g->rows.size() * (double)cap * 8.0 / 1e9);
        if (cap != g->tick_cap || !g->d_ticks) {
            if (g->d_ticks) HSG_HIP(g, hipFree(g->d_ticks));
            g->d_ticks = nullptr;
            HSG_HIP(g, hipMalloc(&g->d_ticks, g->rows.size() * (size_t)cap * sizeof(int64_t)));
            g->tick_cap = cap;
        }
        { const int rc = ensure_stream(g); if (rc) return rc; }
        HSG_HIP(g, hs::tick_tables_launch(g->stream, g->d_rows, (int)g->rows.size(), g->cfg.start_ns, horizon, cap, g->d_ticks, g->d_tick_count,
                                          g->d_tick_status, budget, false));
        HSG_HIP(g, hipStreamSynchronize(g->stream));
        unsigned long long st[2] = {0ull, 0ull};
        HSG_HIP(g, hipMemcpy(st, g->d_tick_status, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] != 0ull)
            return gfail(g, HS_E_UNSUPPORTED, "node %lld: one tick of its time-varying Source / Probe needs more than 64 x %lld adaptive-Simpson "
                         "intervals (hs_graph_config.profile_budget raises the limit) -- refused instead of stalling the device",
                         (long long)st[0] - 2, budget);
        if (st[1] == 0ull) {
            g->tick_horizon = horizon;
            g->ctl.ticks = g->d_ticks; g->ctl.tick_cap = cap; g->ctl.tick_count = g->d_tick_count;
            return HS_OK;
        }
        cap *= 2;                                           // (a Poisson stream that ran ahead of its mean: a longer table)
    }
    return gfail(g, HS_E_OVERFLOW, "a tick table overflowed after eight doublings");
}

// What a run needs before its first launch: tick tables up to the end, the schedule()d entries on the device.
static int prepare_run(hs_graph *g, int64_t end_ns) {
    GCtl &c = g->ctl;
    // table-driven streams: up to this end when it is a real horizon, else (an auto-terminating run) a minute at a time
    int64_t table_h = end_ns;
    if (!g->rows.empty()) {
        const int64_t minute = 60ll * 1000000000ll;
        if (end_ns - g->cfg.start_ns > 64 * minute) table_h = std::max<int64_t>(g->tick_horizon, g->cfg.start_ns + minute);
        const int rc = build_tables(g, table_h);
        if (rc) return rc;
    }
    const long long ns = (long long)g->sched_node.size();
    if (ns > g->d_sched_cap) {
        if (g->d_sched_node) HSG_HIP(g, hipFree(g->d_sched_node));
        if (g->d_sched_t) HSG_HIP(g, hipFree(g->d_sched_t));
        g->d_sched_node = nullptr; g->d_sched_t = nullptr;
        g->d_sched_cap = ns + ns / 2 + 16;
        HSG_HIP(g, hipMalloc(&g->d_sched_node, (size_t)g->d_sched_cap * sizeof(int32_t)));
        HSG_HIP(g, hipMalloc(&g->d_sched_t, (size_t)g->d_sched_cap * sizeof(int64_t)));
    }
    if (ns > 0) {
        HSG_HIP(g, hipMemcpy(g->d_sched_node, g->sched_node.data(), (size_t)ns * sizeof(int32_t), hipMemcpyHostToDevice));
        HSG_HIP(g, hipMemcpy(g->d_sched_t, g->sched_t.data(), (size_t)ns * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    c.sched_node = g->d_sched_node; c.sched_t = g->d_sched_t; c.n_sched = ns;
    c.end_ns = end_ns;
    g->launches = 0;
    return HS_OK;
}

// What a launch left (the device is idle): enlarge what it ran out of; *done = the run reached its end.
static int after_launch_with(hs_graph *g, int status, long long processed, bool *done);
static int after_launch(hs_graph *g, bool *done) {
    GVars v;
    HSG_HIP(g, hipMemcpy(&v, g->ctl.V, sizeof v, hipMemcpyDeviceToHost));
    return after_launch_with(g, v.status, v.processed, done);
}
static int after_launch_with(hs_graph *g, int status, long long processed, bool *done) {
    GCtl &c = g->ctl;
    g->launches++;
    struct { int status; long long processed; } v{status, processed};
    if (v.status & kBadKind) return gfail(g, HS_E_INVALID, "an event of unknown kind reached the loop (internal error)");
    if (g->cfg.max_events > 0 && v.processed > g->cfg.max_events)
        return gfail(g, HS_E_UNSUPPORTED, "the run exceeds max_events = %lld events on the single-heap path (one lane, ~2.4 us per event); "
                     "raise max_events, or bring the graph into the shape the station engines take", (long long)g->cfg.max_events);
    if (v.status & kGrowHeap) {
        const long long nc = c.heap_cap * 2;
        int rc = grow(g, &c.heap, c.heap_cap, nc); if (rc) return rc;
        c.heap_cap = nc;
    }
    if (v.status & kGrowReq) {
        if (c.req_cap >= (1 << 30)) return gfail(g, HS_E_OVERFLOW, "more than 2^30 Requests alive at once");
        const int nc = c.req_cap * 2;
        int rc = grow(g, &c.reqs, c.req_cap, nc); if (rc) return rc;
        c.req_cap = nc;
    }
    if (v.status & kGrowTicks) {
        // a table-driven stream reached the end of its table: twice the span (never beyond the end the caller asked for + the two
        // ticks every table holds beyond its horizon)
        const int64_t span = g->tick_horizon - g->cfg.start_ns;
        int64_t nh = g->cfg.start_ns + (span > 0 ? 2 * span : 1000000000ll);
        if (nh <= g->tick_horizon) return gfail(g, HS_E_OVERFLOW, "the tick tables cannot grow any further");
        const int rc = build_tables(g, nh);
        if (rc) return rc;
    }
    if (v.status & kGrowRec) {
        const long long nc = c.rec_cap * 2;
        int rc;
        if ((rc = grow(g, &c.rec_node, c.rec_cap, nc))) return rc;
        if ((rc = grow(g, &c.rec_t, c.rec_cap, nc))) return rc;
        if ((rc = grow(g, &c.rec_cr, c.rec_cap, nc))) return rc;
        c.rec_cap = nc;
    }
    if (v.status & kUndecided) { g->undecided = true; *done = true; return HS_OK; }
    *done = v.status == kDone;
    return HS_OK;
}

extern "C" {

int hs_graph_run_until(hs_graph *g, int64_t end_ns) {
    if (!g) return gfail(g, HS_E_INVALID, "null handle");
    HSG_HIP(g, hipSetDevice(g->cfg.device));
    {
        const int rc = prepare_run(g, end_ns);
        if (rc) return rc;
    }
    { const int rc = ensure_stream(g); if (rc) return rc; }
    HSG_HIP(g, hipEventRecord(g->ev_a, g->stream));
    for (bool done = false; !done;) {
        hipLaunchKernelGGL(hs_graph_run, dim3(1), dim3(64), 0, g->stream, g->ctl);
        HSG_HIP(g, hipGetLastError());
        HSG_HIP(g, hipStreamSynchronize(g->stream));
        const int rc = after_launch(g, &done);
        if (rc) return rc;
    }
    HSG_HIP(g, hipEventRecord(g->ev_b, g->stream));
    HSG_HIP(g, hipStreamSynchronize(g->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g->ev_a, g->ev_b) == hipSuccess) g->last_run_ms = ms;
    g->ran = true;
    return HS_OK;
}

static int run_batch(hs_graph *const *gs, int32_t n, int64_t end_ns, int part) {
    if (!gs || n < 1) return gfail(nullptr, HS_E_INVALID, "no handles");
    for (int i = 0; i < n; ++i) {
        if (!gs[i]) return gfail(nullptr, HS_E_INVALID, "handle %d is null", i);
        if (gs[i]->cfg.device != gs[0]->cfg.device) return gfail(gs[i], HS_E_INVALID, "handle %d lives on another device", i);
    }
    {   // (a handle listed twice would run on one heap from two workgroups)
        std::vector<hs_graph *> sorted(gs, gs + n);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return gfail(gs[0], HS_E_INVALID, "a handle is listed twice");
    }
    hs_graph *g0 = gs[0];
    HSG_HIP(g0, hipSetDevice(g0->cfg.device));
    { const int rc = ensure_stream(g0); if (rc) return rc; }
    for (int i = 0; i < n; ++i) {
        gs[i]->ctl.part = part;
        const int rc = prepare_run(gs[i], end_ns);
        if (rc) { if (gs[i] != g0) gfail(g0, rc, "graph %d: %s", i, gs[i]->error.c_str()); return rc; }
    }
    GCtl *d_ctl = nullptr;
    HSG_HIP(g0, hipMalloc(&d_ctl, (size_t)n * (sizeof(GCtl) + kBatchStat * sizeof(long long))));
    long long *d_stat = reinterpret_cast<long long *>(d_ctl + n);
    std::vector<long long> h_stat;
    std::vector<int> pending((size_t)n);
    for (int i = 0; i < n; ++i) pending[(size_t)i] = i;
    std::vector<GCtl> h_ctl;
    int rc = HS_OK;
    hipError_t he = hipEventRecord(g0->ev_a, g0->stream);
    while (he == hipSuccess && rc == HS_OK && !pending.empty()) {
        h_ctl.clear();
        for (int i : pending) h_ctl.push_back(gs[i]->ctl);
        if ((he = hipMemcpy(d_ctl, h_ctl.data(), h_ctl.size() * sizeof(GCtl), hipMemcpyHostToDevice)) != hipSuccess) break;
        hipLaunchKernelGGL(hs_graph_run_batch, dim3((unsigned)pending.size()), dim3(64), 0, g0->stream, (const GCtl *)d_ctl, d_stat);
        if ((he = hipGetLastError()) != hipSuccess) break;
        if ((he = hipStreamSynchronize(g0->stream)) != hipSuccess) break;
        h_stat.resize((size_t)kBatchStat * pending.size());
        if ((he = hipMemcpy(h_stat.data(), d_stat, h_stat.size() * sizeof(long long), hipMemcpyDeviceToHost)) != hipSuccess) break;
        std::vector<int> left;
        size_t slot = 0;
        for (int i : pending) {
            bool done = false;
            const long long *st = &h_stat[(size_t)kBatchStat * slot++];
            rc = after_launch_with(gs[i], (int)st[0], st[1], &done);
            gs[i]->pending_events = st[2]; gs[i]->earliest_ns = st[3];
            if (rc) { if (gs[i] != g0) gfail(g0, rc, "graph %d: %s", i, gs[i]->error.c_str()); break; }
            if (done) gs[i]->ran = true; else left.push_back(i);
        }
        pending.swap(left);
    }
    if (he == hipSuccess && rc == HS_OK) {
        he = hipEventRecord(g0->ev_b, g0->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(g0->stream);
        float ms = 0.f;
        if (he == hipSuccess && hipEventElapsedTime(&ms, g0->ev_a, g0->ev_b) == hipSuccess)
            for (int i = 0; i < n; ++i) gs[i]->last_run_ms = ms;          // (the batch's wall time: the heaps ran side by side)
    }
    (void)hipFree(d_ctl);
    for (int i = 0; i < n; ++i) gs[i]->ctl.part = 0;
    if (rc) return rc;
    if (he != hipSuccess) return gfail(g0, HS_E_HIP, "batch launch: %s", hipGetErrorString(he));
    return HS_OK;
}

int hs_graph_run_many(hs_graph *const *gs, int32_t n, int64_t end_ns) { return run_batch(gs, n, end_ns, 0); }

int hs_graph_run_parts(hs_graph *const *gs, int32_t n, int64_t end_ns) {
    {
        const int rc = run_batch(gs, n, end_ns, 1);
        if (rc) return rc;
    }
    hs_graph *g0 = gs[0];
    for (int i = 0; i < n; ++i) if (gs[i]->undecided) return 1;
    // every part stands in front of its first event beyond end_ns: the reference processes the earliest of them (its loop tests the
    // PREVIOUS event's time, core/simulation.py:472) and nothing else
    int best = -1; int64_t best_t = 0; bool tie = false;
    for (int i = 0; i < n; ++i) {
        if (gs[i]->pending_events <= 0) continue;
        const int64_t t = gs[i]->earliest_ns;
        if (best < 0 || t < best_t) { best = i; best_t = t; tie = false; }
        else if (t == best_t) tie = true;              // (two parts' events on one nanosecond: their order is the whole Simulation's)
    }
    if (tie) return 1;
    if (best < 0) return HS_OK;
    hs_graph *g = gs[best];
    {
        const int rc = ensure_stream(g);
        if (rc) { gfail(g0, rc, "%s", g->error.c_str()); return rc; }
    }
    GCtl one = g->ctl;
    one.part = 1; one.budget = 1; one.end_ns = INT64_MAX;
    GVars before;
    HSG_HIP(g0, hipMemcpy(&before, g->ctl.V, sizeof before, hipMemcpyDeviceToHost));
    for (int attempt = 0; attempt < 64; ++attempt) {
        one.heap = g->ctl.heap; one.heap_cap = g->ctl.heap_cap; one.reqs = g->ctl.reqs; one.req_cap = g->ctl.req_cap;
        one.rec_node = g->ctl.rec_node; one.rec_t = g->ctl.rec_t; one.rec_cr = g->ctl.rec_cr; one.rec_cap = g->ctl.rec_cap;
        one.ticks = g->ctl.ticks; one.tick_cap = g->ctl.tick_cap; one.tick_count = g->ctl.tick_count;
        hipLaunchKernelGGL(hs_graph_run, dim3(1), dim3(64), 0, g->stream, one);
        HSG_HIP(g0, hipGetLastError());
        HSG_HIP(g0, hipStreamSynchronize(g->stream));
        bool done = false;
        const int rc = after_launch(g, &done);       // (enlarges what the one event needed)
        if (rc) { if (g != g0) gfail(g0, rc, "%s", g->error.c_str()); return rc; }
        GVars now;
        HSG_HIP(g0, hipMemcpy(&now, g->ctl.V, sizeof now, hipMemcpyDeviceToHost));
        if (now.processed > before.processed || now.heap_len <= 0) {
            return g->undecided ? 1 : HS_OK;      // (a neighbour of the other origin on the event's own nanosecond: the device's check)
        }
    }
    return gfail(g0, HS_E_OVERFLOW, "the event beyond the end could not be processed");
}

int hs_graph_get_summary(hs_graph *g, hs_summary *out) {
    if (!g || !out) return gfail(g, HS_E_INVALID, "null argument");
    HSG_HIP(g, hipSetDevice(g->cfg.device));
    GVars v;
    HSG_HIP(g, hipMemcpy(&v, g->ctl.V, sizeof v, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->events_processed = v.processed;
    for (int k = 0; k < HS_EV_KINDS; ++k) out->events_by_kind[k] = v.by_kind[k];
    out->final_time_ns = v.cur;
    out->requests_completed = v.completed;
    out->sink_records = v.received;
    out->last_run_ms = g->last_run_ms;
    out->kernel_ms = g->last_run_ms;
    out->launches = g->launches;
    return HS_OK;
}

int hs_graph_get_stats(hs_graph *g, hs_graph_stats *o) {
    if (!g || !o) return gfail(g, HS_E_INVALID, "null argument");
    HSG_HIP(g, hipSetDevice(g->cfg.device));
    std::vector<GState> S((size_t)g->n);
    HSG_HIP(g, hipMemcpy(S.data(), g->ctl.S, (size_t)g->n * sizeof(GState), hipMemcpyDeviceToHost));
    for (int i = 0; i < g->n; ++i) {
        const GState &s = S[(size_t)i];
        const int k = g->params[(size_t)i].kind;
        const bool src = k == HS_NODE_SOURCE, srv = k == HS_NODE_SERVER, lnk = k == HS_NODE_LINK;
        if (o->generated) o->generated[i] = src ? s.c : 0;
        if (o->payloads) o->payloads[i] = src ? s.d : 0;
        if (o->accepted) o->accepted[i] = srv ? s.a : 0;
        if (o->dropped) o->dropped[i] = srv ? s.b : 0;
        if (o->completed) o->completed[i] = srv ? s.c : 0;
        if (o->rejected) o->rejected[i] = srv ? s.d : 0;
        if (o->total_service_s) o->total_service_s[i] = srv ? s.total_service : 0.0;
        if (o->queue_depth) o->queue_depth[i] = srv ? s.qlen : 0;
        if (o->active) o->active[i] = srv ? s.active : 0;
        if (o->received) o->received[i] = k == HS_NODE_SINK ? s.a : 0;
        if (o->entered) o->entered[i] = lnk ? s.a : 0;
        if (o->packets_sent) o->packets_sent[i] = lnk ? s.b : 0;
        if (o->packets_dropped) o->packets_dropped[i] = lnk ? s.c : 0;
        if (o->routed) o->routed[i] = k == HS_NODE_ROUTER ? s.a : 0;
        if (o->lb) {
            const bool lb = k == HS_NODE_LB;
            int64_t *r = o->lb + 6 * (size_t)i;
            r[0] = lb ? s.a : 0; r[1] = lb ? s.b : 0; r[2] = lb ? s.c : 0; r[3] = lb ? s.c : 0; r[4] = lb ? s.d : 0;
            r[5] = lb ? (int64_t)s.svc_draws : 0;
        }
    }
    if (o->rt_taken && g->n_rt > 0)
        HSG_HIP(g, hipMemcpy(o->rt_taken, g->ctl.rt_taken, (size_t)g->n_rt * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HS_OK;
}

int64_t hs_graph_read_records(hs_graph *g, int32_t *node, int64_t *t_ns, int64_t *created_ns, int64_t cap) {
    if (!g) return gfail(g, HS_E_INVALID, "null handle");
    if (hipSetDevice(g->cfg.device) != hipSuccess) return gfail(g, HS_E_HIP, "hipSetDevice failed");
    GVars v;
    if (hipMemcpy(&v, g->ctl.V, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return gfail(g, HS_E_HIP, "reading the run's scalars failed");
    const long long m = std::min<long long>(v.rec_n, cap > 0 ? cap : 0);
    if (m > 0) {
        if (node && hipMemcpy(node, g->ctl.rec_node, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return gfail(g, HS_E_HIP, "record copy failed");
        if (t_ns && hipMemcpy(t_ns, g->ctl.rec_t, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return gfail(g, HS_E_HIP, "record copy failed");
        if (created_ns && hipMemcpy(created_ns, g->ctl.rec_cr, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return gfail(g, HS_E_HIP, "record copy failed");
    }
    return v.rec_n;
}

}  // extern "C"
