// hs_station.hpp -- the per-LP state machine: one station LP per lane.
//
// A station LP = [optional Source] -> Server(c, FIFO, capacity) -> [Sink].  It replaces, for that entity
// set, the reference's heap loop (`Simulation._execute_until`, core/simulation.py:449-505) and the
// Queue/Driver/Worker micro-protocol in front of every Server.  Only TWO kinds of real
// timestamps exist per LP -- the pending source tick A and up to c pending departures D[j]; every other
// reference event (Request@Server, QUEUE_NOTIFY, QUEUE_POLL, QUEUE_DELIVER, Request@worker,
// Request@Sink) happens at the timestamp of the tick or departure that caused it.  The LP therefore
// advances one TIMESTAMP GROUP at a time and counts each reference event as it "happens", so that
// `total_events_processed` and the per-kind histogram equal the reference's.
//
// Exactness at equal timestamps.  The reference orders same-time events by creation order
// (`_sort_index`, core/event.py:337-344).  Inside one LP that order is LP-local, so it is reproduced
// with a local creation counter `seq`:
//   * events pending from earlier groups (the tick, the departures) that share the group's timestamp
//     run first, in creation order;
//   * events created inside the group run after them in creation order (a FIFO), except the retargeted
//     payload, which keeps its old index and therefore runs immediately after its QUEUE_DELIVER
//     (components/queue_driver.py:86-90);
//   * the common case -- exactly one pending event at the group's timestamp and no new event landing on
//     the same nanosecond -- is a single chain with at most one event in flight and is executed as
//     straight-line code (the fast path); anything else goes through the general in-group FIFO.
#pragma once

#include "hs_device.hpp"

namespace hs {

// in-group event codes (low 3 bits) | slot << 3
enum : uint32_t { Q_ENQ = 1, Q_NOTIFY = 2, Q_POLL = 3, Q_DELIVER = 4, Q_TICK = 5, Q_CONT = 6, Q_SINK = 7 };

constexpr int kBlock = 256;     // LPs per workgroup (4 wavefronts)
constexpr int kQCap = 48;       // in-group FIFO capacity per LP (LDS)

struct StationParams {          // read-only, [n_lp] each
    const uint8_t *src_kind;
    const double *src_rate;
    const int64_t *src_stop;
    const int32_t *conc;
    const uint8_t *svc_kind;
    const double *svc_mean;
    const int64_t *qcap;
    const uint8_t *egress;
    const uint64_t *seed;
    const uint64_t *stream_base;
};

struct StationState {           // read-write; [n_lp] each unless noted
    int64_t *A;                 // pending tick time (kInfNs: none)
    uint32_t *seqA;             // creation stamp of the pending tick
    int64_t *crtA;              // simulation time at which the pending tick was created
    uint64_t *arr_k;            // arrival draws consumed
    int64_t *arr_time;          // ArrivalTimeProvider.current_time
    uint64_t *svc_k;            // service draws consumed
    int64_t *D;                 // [C][n_lp] pending departure times (kInfNs: free slot)
    uint32_t *seqD;             // [C][n_lp]
    int64_t *crtD;              // [C][n_lp] simulation time at which the departure was created (service start)
    double *svc_s;              // [C][n_lp] service_time_s of the job in the slot
    int64_t *crt;               // [C][n_lp] created_at of the job in the slot (C > 1 only)
    uint32_t *seq;              // local creation counter
    int64_t *buf;               // waiting requests (FIFO length, excludes jobs in service)
    int32_t *active;            // FixedConcurrency._active
    int64_t *generated, *accepted, *dropped, *completed, *rejected, *started, *received, *sink_w;
    double *total_service;
    uint32_t *q;                // in-group events left pending by an overshoot (<= 2 codes, 8 bit each, + count<<16)
    int64_t *grp_time;          // timestamp of that pending group
    int64_t *last_time;         // time of the LP's last processed event
    int64_t *events;            // events processed by this LP
    int64_t *ev_kind;           // [HS_EV_KINDS = 11][n_lp]
};

struct RecordLogs {
    int64_t *adm;               // [n_lp][cap] created_at of the k-th accepted request (FIFO backing store)
    int64_t *sink_t;            // [n_lp][cap] completion time of the m-th sink record
    int64_t *sink_created;      // [n_lp][cap] created_at of the m-th sink record (C > 1; C == 1 aliases adm)
    int64_t *sink_created_own;  // the separately allocated column (null when the alias is the only option)
    int64_t cap;
};

struct Totals {                 // engine-wide accumulators (device memory)
    unsigned long long ev[11];
    unsigned long long completed;
    unsigned long long received;
    long long final_time;       // max over LPs of last processed time (REPLICAS) / global current time (SINGLE)
    long long cur_time;         // SINGLE: Simulation._current_time
    int overflow;
    int qoverflow;
    unsigned int done;          // last-block ticket
    int pad;
};

struct Candidate {              // an LP's first event beyond end_ns (SINGLE-mode overshoot election)
    long long t;                // event time
    long long t_created;        // when it was created (proxy for the global sort index)
    int lp;
    int valid;
};

// ---------------------------------------------------------------------------------------------
template <int C>
struct Station {
    // parameters
    int lp, n;
    uint32_t src_kind, svc_kind, egress;
    int32_t conc;
    double rate, svc_mean, svc_lambda;
    int64_t stop_ns, qcap, svc_const_ns;
    double svc_const_s;
    // state
    int64_t A, crtA, arr_time, buf, generated, accepted, dropped, completed, rejected, started, received, sink_w;
    uint32_t seqA, seq;
    int32_t active;
    int64_t D[C];
    uint32_t seqD[C];
    int64_t crtD[C];
    double svc_s[C];
    int64_t crt[C];
    double total_service;
    int64_t last_time, grp_time;
    Stream arr, svc;
    // per-run deltas
    uint32_t ev[8];
    // logs
    int64_t *adm, *sink_t, *sink_created;
    int64_t cap;
    int overflow;
    // in-group FIFO (LDS), column `tid`
    uint8_t (*qmem)[kBlock];
    int tid;
    int qh, qn;
    int qoverflow;
    bool force_general;         // debug: route every group through the general FIFO path

    __device__ __forceinline__ void qpush(uint32_t code) {
        if (qn >= kQCap) { qoverflow = 1; return; }
        qmem[(qh + qn) % kQCap][tid] = (uint8_t)code;
        ++qn;
    }
    __device__ __forceinline__ uint32_t qpop() {
        const uint32_t c = qmem[qh][tid];
        qh = (qh + 1) % kQCap;
        --qn;
        return c;
    }

    // ---- ArrivalTimeProvider.next_arrival_time, constant-rate fast path (load/arrival_time_provider.py:72-82)
    __device__ __forceinline__ int64_t next_arrival() {
        double area;
        if (src_kind == 1) area = exp1_from_uniform(arr.next_uniform());  // Poisson: -log(1-u) (providers/poisson_arrival.py:31)
        else area = 1.0;                                                  // constant   (providers/constant_arrival.py:23)
        const double t_next = __dadd_rn(seconds_from_ns(arr_time), __ddiv_rn(area, rate));
        arr_time = ns_from_seconds(t_next);
        return arr_time;
    }

    // ---- service sample: get_latency(...).to_seconds() then `yield s` (server/server.py:246-250)
    __device__ __forceinline__ void sample_service(double &s, int64_t &dur_ns) {
        if (svc_kind == 0) {
            const double sample = __ddiv_rn(exp1_from_uniform(svc.next_uniform()), svc_lambda);  // expovariate(lambda)
            s = seconds_from_ns(ns_from_seconds(sample));   // Duration.from_seconds(sample).to_seconds()
            dur_ns = ns_from_seconds(s);                    // Instant + float: ns + int(s * 1e9)
        } else {
            s = svc_const_s;
            dur_ns = svc_const_ns;
        }
    }

    // ---- the reference handlers, one per event kind -----------------------------------------
    // Source.handle_event (load/source.py:142-180).  Returns bit0: payload created, bit1: next tick lands on `t`.
    __device__ __forceinline__ uint32_t do_tick(int64_t t) {
        ev[0]++;
        generated++;
        const bool payload = !(stop_ns >= 0 && t > stop_ns);             // SimpleEventProvider.get_events :68
        const int64_t a2 = next_arrival();
        uint32_t r = payload ? 1u : 0u;
        if (a2 == t) { r |= 2u; A = kInfNs; }
        else if (a2 < t) { A = kInfNs; }                                  // popped later as "time travel" and dropped (simulation.py:480-489)
        else { A = a2; seqA = seq++; crtA = t; }
        return r;
    }
    // QueuedResource.handle_event -> Queue._handle_enqueue (components/queue.py:122-147).  True: QUEUE_NOTIFY created.
    __device__ __forceinline__ bool do_enqueue(int64_t t) {
        ev[1]++;
        if (qcap >= 0 && buf >= qcap) { dropped++; return false; }        // FIFOQueue.push refuses (queue_policy.py:94-98)
        const bool was_empty = (buf == 0);
        if (accepted < cap) adm[accepted] = t; else overflow = 1;        // context["created_at"] = tick time
        accepted++;
        buf++;
        return was_empty;
    }
    // QueueDriver._handle_notify (components/queue_driver.py:92-99).  True: QUEUE_POLL created.
    __device__ __forceinline__ bool do_notify() { ev[2]++; return active < conc; }
    // Queue._handle_poll (components/queue.py:149-166).  True: QUEUE_DELIVER created.
    __device__ __forceinline__ bool do_poll() {
        ev[3]++;
        if (buf == 0) return false;
        buf--;
        return true;
    }
    // QUEUE_DELIVER @ driver (queue_driver.py:66-90) immediately followed by the retargeted payload @ worker:
    // Server.handle_queued_event up to its yield (server/server.py:202-250).  Returns slot+1 if the departure
    // lands on `t` itself (zero-length service), else 0.
    __device__ __forceinline__ uint32_t do_deliver_work(int64_t t) {
        ev[4]++;
        ev[5]++;
        const int64_t k = started++;
        if (active >= conc) { rejected++; return 0; }                    // acquire() failed (server.py:223-234)
        active++;
        double s; int64_t dur;
        sample_service(s, dur);
        int j = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; --i) if (D[i] == kInfNs) j = i;
        const int64_t d = t + dur;
        uint32_t same = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == j) {
            svc_s[i] = s;
            if (C > 1) crt[i] = (k < cap) ? adm[k] : 0;
            if (d == t) { D[i] = kInfNs - 1; same = (uint32_t)i + 1; }   // in-group continuation: parked, not pending
            else { D[i] = d; seqD[i] = seq++; crtD[i] = t; }
        }
        return same;
    }
    // generator resumes (server/server.py:252-273) + schedule_poll hook (queue_driver.py:79-84).
    // Returns bit0: Request@Sink created, bit1: QUEUE_POLL created.
    __device__ __forceinline__ uint32_t do_cont(int slot, int64_t t) {
        ev[6]++;
        double s = 0.0; int64_t cr = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == slot) { s = svc_s[i]; cr = crt[i]; D[i] = kInfNs; }
        active = active > 0 ? active - 1 : 0;
        completed++;
        total_service = __dadd_rn(total_service, s);
        uint32_t r = 0;
        if (egress == 1) {
            if (sink_w < cap) { sink_t[sink_w] = t; if (C > 1) sink_created[sink_w] = cr; }
            else overflow = 1;
            sink_w++;
            r |= 1u;
        }
        if (active < conc) r |= 2u;
        return r;
    }
    // Sink.handle_event (components/common.py:36-44); the record itself was staged by do_cont.
    __device__ __forceinline__ void do_sink() { ev[7]++; received++; }
    // A Source wired straight to a Sink/Counter (no Server in this LP): the payload IS the Sink's event.
    __device__ __forceinline__ void stage_direct_sink(int64_t t) {
        if (sink_w < cap) { sink_t[sink_w] = t; if (C > 1) sink_created[sink_w] = t; else adm[sink_w] = t; }
        else overflow = 1;
        sink_w++;
    }

    // ---- chains with at most one event in flight (fast path pieces) ---------------------------
    // returns true if the general FIFO must take over (a same-time continuation was created)
    __device__ __forceinline__ bool chain_from_poll(int64_t t) {
        if (!do_poll()) return false;
        const uint32_t same = do_deliver_work(t);
        if (same) { qpush(Q_CONT | ((same - 1) << 3)); return true; }
        return false;
    }
    __device__ __forceinline__ bool chain_from_enqueue(int64_t t) {
        if (!do_enqueue(t)) return false;
        if (!do_notify()) return false;
        return chain_from_poll(t);
    }

    // ---- roots: the first micro-event of a pending tick / departure; created events go to the FIFO
    __device__ __forceinline__ void root_tick(int64_t t) {
        const uint32_t r = do_tick(t);
        if (r & 1u) {
            if (svc_kind == 2) { if (egress == 1) { stage_direct_sink(t); qpush(Q_SINK); } }
            else qpush(Q_ENQ);
        }
        if (r & 2u) qpush(Q_TICK);
    }
    __device__ __forceinline__ void root_cont(int slot, int64_t t) {
        const uint32_t r = do_cont(slot, t);
        if (r & 1u) qpush(Q_SINK);
        if (r & 2u) qpush(Q_POLL);
    }

    // pending (pre-group) root at time t with the smallest creation stamp: -1 none, 0 tick, 1+slot departure
    __device__ __forceinline__ int pick_root(int64_t t) const {
        int best = -1;
        uint32_t bs = 0xffffffffu;
        if (A == t) { best = 0; bs = seqA; }
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (D[i] == t && (best < 0 || (int32_t)(seqD[i] - bs) < 0)) { best = 1 + i; bs = seqD[i]; }
        return best;
    }
    __device__ __forceinline__ void run_root(int which, int64_t t) {
        if (which == 0) root_tick(t); else root_cont(which - 1, t);
    }

    // general in-group FIFO drain
    __device__ __forceinline__ void drain(int64_t t) {
        while (qn > 0) {
            const uint32_t code = qpop();
            switch (code & 7u) {
                case Q_ENQ: if (do_enqueue(t)) qpush(Q_NOTIFY); break;
                case Q_NOTIFY: if (do_notify()) qpush(Q_POLL); break;
                case Q_POLL: if (do_poll()) qpush(Q_DELIVER); break;
                case Q_DELIVER: {
                    // do_poll already popped the buffer; deliver + work
                    const uint32_t same = do_deliver_work(t);
                    if (same) qpush(Q_CONT | ((same - 1) << 3));
                } break;
                case Q_TICK: root_tick(t); break;
                case Q_CONT: root_cont((int)(code >> 3), t); break;
                case Q_SINK: do_sink(); break;
                default: break;
            }
        }
    }

    // whole group at time t (t <= end): pending roots in creation order, then the FIFO
    __device__ __forceinline__ void run_group_general(int64_t t) {
        for (;;) {
            const int w = pick_root(t);
            if (w < 0) break;
            run_root(w, t);
        }
        drain(t);
    }

    __device__ __forceinline__ int64_t next_time() const {
        int64_t t = A;
#pragma unroll
        for (int i = 0; i < C; ++i) t = D[i] < t ? D[i] : t;
        return t;
    }

    __device__ __forceinline__ void run_group(int64_t t) {
        int n_at = (A == t) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < C; ++i) n_at += (D[i] == t) ? 1 : 0;
        if (n_at == 1 && !force_general) {
            // Fast path: one event in flight at a time.  Both kinds of root converge on ONE poll/deliver/work
            // site so that a wavefront whose lanes mix ticks and departures executes the (expensive) service
            // draw once per iteration, not once per branch.
            bool general = false, want_poll = false;
            if (A == t) {
                const uint32_t r = do_tick(t);
                if (svc_kind == 2) {             // Source -> Sink directly
                    if ((r & 1u) && egress == 1) { stage_direct_sink(t); do_sink(); }
                    if (r & 2u) { qpush(Q_TICK); general = true; }
                }
                else if (r & 2u) { if (r & 1u) qpush(Q_ENQ); qpush(Q_TICK); general = true; }
                else if (r & 1u) want_poll = do_enqueue(t) && do_notify();
            } else {
                int slot = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (D[i] == t) slot = i;
                const uint32_t r = do_cont(slot, t);
                if (r & 1u) do_sink();
                want_poll = (r & 2u) != 0;
            }
            if (want_poll) general = chain_from_poll(t);
            if (general) drain(t);
        } else {
            run_group_general(t);
        }
        last_time = t;
    }
};

}  // namespace hs
