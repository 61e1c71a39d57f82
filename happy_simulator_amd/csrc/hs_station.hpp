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
#include "hs_tables.hpp"

namespace hs {

// in-group event codes (low 3 bits) | slot << 3
enum : uint32_t { Q_PSAMPLE = 0, Q_ENQ = 1, Q_NOTIFY = 2, Q_POLL = 3, Q_DELIVER = 4, Q_TICK = 5, Q_CONT = 6, Q_SINK = 7 };
constexpr int kMaxProbes = 4;   // Probes per LP (Probe.on_many: several metrics of one entity, instrumentation/probe.py:119-164)
constexpr int kRootProbe = 100; // pick_root: the pending tick of probe j is kRootProbe + j
constexpr int kMaxXSrc = 3;     // further Sources feeding the LP's Server (slots 1 .. 3; slot 0 is the LP's first Source)
constexpr int kRootXSrc = 120;  // pick_root: the pending tick of further Source j is kRootXSrc + j
// arrival stream of further Source j of an LP: an entity of its own (include/hs_engine.h `src_more_kind`)
__device__ __forceinline__ uint64_t xsrc_stream_id(uint64_t lp_stream_base, int j) {
    return stream_id((1ull << 40) | (lp_stream_base << 2) | (uint64_t)j, kStreamArrival);
}
constexpr int kRootSched = 98;  // pick_root: the next Request injected with Simulation.schedule()
constexpr int kRootInj = 200;   // pick_root: kRootInj + 32 u + j = forward j of the run of forwards that arrive from upstream Server u at this nanosecond
constexpr int kMaxUp = 4;       // upstream Servers of one Server on the passes (more: the single-heap loop)
constexpr int kMaxInjRun = 32;  // ... such runs are one long (two upstream workers finishing on one nanosecond: two)
constexpr uint32_t kEgressServer = 4;   // HS_EGRESS_SERVER: the Server forwards to another Server (tandem queues)

constexpr int kBlock = 256;     // LPs per workgroup (4 wavefronts)
constexpr int kQCap = 48;       // in-group FIFO capacity per LP (LDS)
#ifndef HS_KRING                // (tuning builds override these; the shipped values are the measured optimum)
#define HS_KRING 24
#endif
#ifndef HS_KREFILL
#define HS_KREFILL 8
#endif
#ifndef HS_PC_SLEEP_P
#define HS_PC_SLEEP_P 2
#endif
#ifndef HS_PC_SLEEP_C
#define HS_PC_SLEEP_C 1
#endif
constexpr int kRing = HS_KRING;       // pre-drawn values buffered per stream per LP (LDS)
constexpr int kRefill = HS_KREFILL;   // values generated per wave-level refill (4 Philox blocks)

// Per-lane ring of pre-drawn stream values, column `tid` of an LDS array [kRing][kBlock].  The serial
// per-LP recursion consumes one value at a time; the expensive part of a draw (Philox block, hs_log,
// the division by the rate / lambda) is produced kRefill values at a time by ALL lanes of the wavefront
// in straight-line code (refill_*), instead of once per loop iteration inside divergent branches.
struct DrawRing {
    double (*mem)[kBlock];
    int tid, head, n;
    __device__ __forceinline__ void reset(double (*m)[kBlock], int t) { mem = m; tid = t; head = 0; n = 0; }
    __device__ __forceinline__ void push(double v) {
        int i = head + n;
        i = i >= kRing ? i - kRing : i;
        mem[i][tid] = v;
        ++n;
    }
    __device__ __forceinline__ double peek() const { return mem[head][tid]; }   // garbage when n == 0 (never used then)
    __device__ __forceinline__ void advance_if(bool p) {
        const int nh = (head + 1 == kRing) ? 0 : head + 1;
        head = p ? nh : head;
        n -= p ? 1 : 0;
    }
    __device__ __forceinline__ double pop() {
        const double v = mem[head][tid];
        head = (head + 1 == kRing) ? 0 : head + 1;
        --n;
        return v;
    }
};

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
    const uint8_t *prof_kind;       // time-varying arrival rate (load/profile.py): 0 constant, 1 linear ramp, 2 spike
    const double *prof_p;           // [4][n_lp]
    // tick times of the streams whose next tick is a numerical inversion -- Sources with such a profile, Probes -- produced before
    // the run by the tick-table kernel (hs_tables.hpp); null = the engine has none
    const TickTables *tabs;
    const uint8_t *probe_metric;    // [kMaxProbes][n_lp] Probes attached to this LP: kProbe* metric, 255 = none (slots fill from 0)
    const double *probe_rate;       // [kMaxProbes][n_lp] 1.0 / interval  (_ProbeProfile.rate, instrumentation/probe.py:27-35)
    // Simulation.schedule() (core/simulation.py:195-206): Requests injected before run(), per LP sorted by time (stable in
    // call order): LP lp owns sched_t[sched_off[lp] .. sched_off[lp + 1]).  null = none.
    const int64_t *sched_off;       // [n_lp + 1]
    const int64_t *sched_t;
    // sort index of every injected Request, written by the prologue (hs_exact.hpp); null = no prologue ran: an injected
    // Request then precedes every run-time event of its nanosecond (true once the run has constructed N_init events)
    const uint32_t *sched_idx;
    // cross-LP ties the creation times do not decide go to the Source the reference constructed first: the LP's position in
    // `sources=[...]` (sourceless LPs after them); null = LP order
    const int32_t *tie_rank;
    // ... [n_lp] those LP ranks, then [kMaxXSrc + 1][n_lp] the position of every SOURCE in that list by slot (a pending TICK
    // competes with its own Source's position: two lock-step Sources of different LPs tie on (time, creation time) at every
    // tick, and the LP whose first-listed Source is another one need not come first), then one word: the offset that puts a
    // Probe's tick behind every Source and every sourceless LP.  One array, so that the run kernels carry no further argument
    // through their loops (two more kernel arguments cost the headline kernel 2 % -- SGPR pressure, measured); cand_rank().
    // ... then [kMaxProbes][n_lp]: the rank of the PROBE in that slot (its position in `probes=[...]`, behind every Source and
    // every sourceless LP): a Probe's tick competes with its own list position.
    // further Sources of the LP (PF instantiations, general path): null = none
    const uint8_t *xsrc_kind;       // [kMaxXSrc][n_lp] 0 none, 1 Poisson, 2 constant
    const double *xsrc_rate;        // [kMaxXSrc][n_lp]
    const int64_t *xsrc_stop;       // [kMaxXSrc][n_lp]
};

// what a Probe samples with getattr(target, metric) (instrumentation/probe.py:51-66)
enum : uint32_t { kProbeDepth = 0, kProbeActive = 1, kProbeAccepted = 2, kProbeDropped = 3, kProbeCompleted = 4,
                  kProbeReceived = 5, kProbeGenerated = 6, kProbeNone = 255 };

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
    int64_t *ev_kind;           // [11][n_lp] station / network kinds
    // Probe (PF instantiation only)
    int64_t *PA;                // [kMaxProbes][n_lp] pending probe tick (kInfNs: none)
    uint32_t *seqP;             // [kMaxProbes][n_lp]
    int64_t *crtP, *p_arr, *p_n;   // [kMaxProbes][n_lp] creation time of the pending tick, its index in the Probe's tick table, samples taken
    int64_t *ev_probe;          // [2][n_lp] SourceEvent@Probe, probe_event
    int64_t *sched_i;           // [n_lp] index into sched_t of the LP's next scheduled Request (PF instantiation only)
    // LINEAGE of every pending event (the election of the one event beyond end_time, cand_less): the reference's heap is a FIFO
    // among the events of one nanosecond -- a group runs breadth-first from its roots (the events that were pending from
    // earlier) -- so of two events CREATED in one nanosecond the one fewer steps from its group's root was created first, then
    // the one whose root was created first (core/event.py:62-77,337-344; tools/election_rules.py).  Per pending event: how many
    // steps after the root of the group it was created in (dp*), and when that root was created (rc*).
    uint8_t *dpA; int64_t *rcA;   // [n_lp] the pending tick
    uint8_t *dpD; int64_t *rcD;   // [C][n_lp] the pending departures
    uint8_t *wkD;                 // [C][n_lp] ... and how many of their steps were retargeted payloads (Station::wk; tandem engines)
    int64_t *rcP;                 // [kMaxProbes][n_lp] the pending probe ticks (always one step from the previous tick)
    uint8_t *dpX; int64_t *rcX;   // [kMaxXSrc][n_lp] the pending ticks of the further Sources
    // ... and of the events waiting in the in-group FIFO (general path only; global memory, [kQCap][n_lp])
    uint8_t *qdep; int64_t *qrc;
    int64_t *enqpay;              // [kEnqPay][n_lp] network engines: ENQ payloads of hs_net_async's general path (null: stations only)
    // further Sources (PF instantiation only; null = none)
    int64_t *XA, *crtX, *x_arr, *x_n;   // [kMaxXSrc][n_lp] pending tick, its creation time, provider time, generated_count
    uint32_t *seqX;             // [kMaxXSrc][n_lp]
    uint64_t *x_k;              // [kMaxXSrc][n_lp] arrival draws consumed
};

struct RecordLogs {
    int64_t *adm;               // [cap][n_lp] created_at of the k-th accepted request (FIFO backing store)
    int64_t *sink_t;            // [cap][n_lp] completion time of the m-th sink record
    int64_t *sink_created;      // [cap][n_lp] created_at of the m-th sink record (C > 1; C == 1 aliases adm)
    int64_t *sink_created_own;  // the separately allocated column (null when the alias is the only option)
    int64_t cap;
    int64_t *probe_t, *probe_v; // [kMaxProbes][pcap][n_lp] sample time / sampled value
    int64_t pcap;
    // Network engines (round 6): the three record logs are [n_lp][cap] instead -- an LP's records are contiguous.  The station
    // kernels advance their lanes request by request in lock step, so a [cap][n_lp] row receives whole 128-byte lines; the LPs of
    // the asynchronous network engine are at different records at any time, a row's line was written 8 bytes at a time over a
    // stretch of the run longer than it survives in the L2 and went to HBM several times as partial sectors (WRITE_SIZE 3.3 GB
    // against 1.05 GB of records on the headline ring).  LP-major, the line an LP appends to stays in the L2 until it is full.
    int32_t lp_major;
};
__device__ __forceinline__ size_t log_at(const RecordLogs &L, int64_t k, int lp, int n) {
    return L.lp_major ? (size_t)lp * (size_t)L.cap + (size_t)k : (size_t)k * (size_t)n + (size_t)lp;
}

struct Totals {                 // engine-wide accumulators (device memory)
    unsigned long long ev[15];  // HS_EV_KINDS; the station / network engines fill 0..10 and 13..14 (probes)
    unsigned long long completed;
    unsigned long long received;
    long long final_time;       // max over LPs of last processed time (REPLICAS) / global current time (SINGLE)
    long long cur_time;         // SINGLE: Simulation._current_time
    int overflow;
    int qoverflow;
    unsigned int done;          // last-block ticket
    // tandem queues: an order between events of DIFFERENT LPs was needed that the lineage key does not decide (the two were created
    // in one nanosecond, equally many steps below roots that were created in one nanosecond as well -- lock-step constant
    // arrivals and services).  The reference decides it by comparing those roots' ancestry, arbitrarily far back; the engine
    // then repeats the run on the single-heap loop (hs_exact.hpp), which is the reference's algorithm.  bit 0: a forwarded Request
    // against an event of the downstream Server, bit 1: the election of the event beyond end_ns.
    // bit 2 (every general-path engine): a PRE-RUN event -- a first tick of a Source or Probe, a Request injected with
    // Simulation.schedule() -- shared its nanosecond with another pending event of its LP.  Only there can the reference's second
    // sort counter (run-time events are numbered from 0 again, core/simulation.py:77) put a run-time event BEFORE a pre-run one;
    // an engine that skipped the prologue (hs_engine.hip `lazy_prologue`) repeats the run with it.
    int undecided;
    unsigned long long dbg[4];  // asynchronous engine telemetry: sum of wave iterations, max, groups run, waves
    unsigned long long not_done; // shard rounds of hs_net_async: LPs that still have work at or before end_ns
    // Network engines, windows: the station whose timestamp group the election stopped inside (-1: none; StationState::q holds what is
    // left of the group, hs_net_resume finishes it), and why the next run_until cannot continue from this state (0: it can)
    int pend_lp, no_resume;
};

struct Candidate {              // an LP's first event beyond end_ns (SINGLE-mode overshoot election)
    long long t;                // event time
    long long t_created;        // when it was created
    long long rcrt;             // when the root of the group it was created in was created (StationState lineage)
    int depth;                  // steps from that root
    int lp;
    int valid;
    int rank;                   // last key: the position in the reference's construction order (`sources=[...]`, `probes=[...]`), cand_rank()
    int pad;                    // what the candidate is, for cand_rank: 0 departure / message / injected Request,
                                // 2 + slot the tick of the LP's Source in that slot, 8 + slot the tick of the Probe in that slot
    int pad2;                   // hs_net_window: the workgroup's other candidates with this one's key (bit 0: any, bit 1: a stand-in)
};

// the last election key of an LP's candidate (see StationParams::tie_rank)
__device__ __forceinline__ int cand_rank(const StationParams &P, int lp, int n, int pad) {
    if (P.tie_rank == nullptr)   // construction order = LP order: the LP's Sources by slot, Probes behind everything
        return pad >= 8 ? n * (kMaxXSrc + 1) + lp * kMaxProbes + (pad - 8) : lp * (kMaxXSrc + 1) + (pad >= 2 ? pad - 2 : 0);
    if (pad >= 8) return P.tie_rank[(size_t)(kMaxXSrc + 2) * (size_t)n + 1 + (size_t)(pad - 8) * (size_t)n + lp];
    if (pad >= 2) return P.tie_rank[(size_t)(pad - 1) * (size_t)n + lp];
    return P.tie_rank[lp];
}

// ---------------------------------------------------------------------------------------------
// PF: the LP may have a time-varying arrival profile (hs_profile.hpp).  A separate instantiation, because the numerical
// inversion needs a 4 KB per-lane stack and would otherwise tax the constant-rate kernel's registers.
// HSG(x, v): a per-LP configuration predicate `x` of the request-order loop that is the compile-time constant `v` in the UNI
// instantiation (every LP of the engine is Source.poisson -> Server(Exp) -> Sink; the host checks it, hs_engine_set_stations):
// 4 % of the headline kernel.
#define HSG(x, v) (UNI ? (v) : (x))
template <int C, bool PF = false, bool UNI = false>
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
    // lineage (StationState::dpA ...): of the pending events, and of the event being processed (cd steps from its group's root,
    // which was created at cr)
    int32_t dpA, dpD[C], cd;
    int64_t rcA, rcD[C], cr;
    int64_t rcP[kMaxProbes];
    int32_t dpX[kMaxXSrc];
    int64_t rcX[kMaxXSrc];
    uint8_t *qdep; int64_t *qrc;     // the in-group FIFO's lineage columns: entry `slot` of this LP at [slot * ls]
    // entity streams (DESIGN.md "Random streams"): draws consumed so far + the pre-drawn values
    uint32_t key0, key1, asid0, asid1, ssid0, ssid1;
    uint64_t arr_k, svc_k;
    DrawRing ra, rs;            // ra: E / rate per arrival draw;  rs: service_time_s per service draw
    ConstDiv div_rate, div_lambda;
    double inc_const;           // constant source: 1.0 / rate
    uint32_t prof_kind;         // != 0: a time-varying rate; tick k of the Source is tab_a[k] (hs_tables.hpp), no arrival draws here
    const int64_t *tab_a;
    const int64_t *tab_p[kMaxProbes];   // tick k of the Probe in slot j is tab_p[j][k]; p_arr[j] = index of its pending tick
    int64_t tab_cap;
    // Probe (PF): a daemon Source of its own (instrumentation/probe.py:81-164) whose ticks sample this LP
    uint32_t p_metric[kMaxProbes], seqP[kMaxProbes];
    int64_t PA[kMaxProbes], crtP[kMaxProbes], p_arr[kMaxProbes], p_n[kMaxProbes], pcap;
    int64_t *probe_t, *probe_v;     // slot j's log starts at probe_t + j * pcap * ls
    int n_probes;
    uint32_t evp[2];
    // further Sources feeding this LP's Server (PF): entities of their own, constant or Poisson rate
    uint32_t x_kind[kMaxXSrc], seqX[kMaxXSrc];
    double x_rate[kMaxXSrc];
    int64_t XA[kMaxXSrc], crtX[kMaxXSrc], x_arr[kMaxXSrc], x_n[kMaxXSrc], x_stop[kMaxXSrc];
    uint64_t x_k[kMaxXSrc], x_base;
    int n_xsrc;
    // Simulation.schedule() (PF): the next injected Request; it was constructed before run(), so it precedes every
    // run-time event of the same nanosecond
    int64_t SA, sc_i, sc_end;
    const int64_t *sc_t;
    const uint32_t *sc_idx;
    // per-run deltas
    uint32_t ev[8];
    // logs
    // logs: record k of this LP is at [k * ls] (ls = n_lp: the logs are [cap][n_lp], so the 64 lanes of a
    // wavefront appending their k-th records write 512 contiguous bytes)
    int64_t *adm, *sink_t, *sink_created;
    int64_t cap, ls;
    int overflow;
    // in-group FIFO (LDS), column `tid`
    uint8_t (*qmem)[kBlock];
    int tid;
    int qh, qn;
    int qoverflow;
    bool force_general;         // debug: route every group through the general FIFO path
    // ---- tandem queues: Server(downstream=<Server>) (components/server/server.py:271-272, core/entity.py:83-105) ----------------
    // A completion forwards the Request to the downstream Server's LP at the same instant.  The engine runs the LPs of a tandem in
    // PASSES, upstream first: pass p's LPs run to end_time while every forward they create goes to a log (time and created_at in the
    // LP's sink_t / sink_created columns, where a Sink's records would go); pass p + 1's LPs read their upstream LP's log as a list
    // of arrivals.  What the downstream LP cannot see from the list alone is where, inside a shared nanosecond, the reference's heap
    // puts the forwarded Request among the LP's own events.  The heap is a FIFO inside a nanosecond (StationState lineage): a group
    // runs breadth-first from its roots -- the events pending from earlier nanoseconds, in the order they were created -- so the
    // forwarded Request, `dep` steps below ITS root (the upstream continuation, if that was pending from earlier), runs after
    // everything fewer steps below any root and, among equals, in root order.  The log therefore carries the root's key -- creation
    // time, its own lineage, construction rank (cand_less) -- and `dep`; the downstream LP enters the root among its own roots by
    // that key (pick_root) as a root that does nothing but create a chain of `dep` placeholders ending in the Request@Server.
    bool trk;                   // the engine has tandem queues: every in-group event knows its root's key (rk_*, with cr)
    int32_t rk_dp, rk_rank;     // key of the root of the chain being processed: its steps from ITS group's root, construction rank,
    int64_t rk_rc;              // ... and that group's root's creation time (its own creation time is cr)
    int64_t cur_pay;            // created_at carried by the FIFO entry just popped (a forward on its way to the enqueue)
    int64_t t_start;            // TickTables::t_start
    // The steps `cd` count the retargeted payload (Request@worker) as a step of its own -- the convention of the election key --
    // but in the heap that payload keeps its OLD sort index and runs at once, inside its QUEUE_DELIVER's turn: it does not take a
    // place in the nanosecond's breadth-first order.  Where a forwarded Request lands among the downstream Server's events is a
    // matter of that order, so the payloads on the way are counted (`wk`) and taken off again (fw_dep: cd - wk places it)
    int32_t wk;
    int32_t wkD[C], rk_wk;      // ... on the way to each pending departure; of the root of the chain being processed
    const int32_t *tie_rank_p; int n_rank;   // cand_rank()'s table
    int64_t *fw_rc, *fw_rrc, *fw_rdr, *fw_dep;   // this LP's forward-log lineage columns, record m at [m * ls]
    int64_t *q_rrc, *q_rdr, *q_pay;              // this LP's FIFO root-key / payload columns, entry `slot` at [slot * ls]
    // forwards arriving from the upstream LPs U[u].up: records i .. n - 1 of THEIR logs (complete: their passes are over)
    struct UpList {
        int up;                 // the upstream LP (-1: none)
        int64_t i, n, IA;       // records consumed / published; IA: time of record i (kInfNs: none left)
        uint32_t mask;          // forwards of the current nanosecond's run already taken as roots (bit j: record i + j)
        int64_t *i_p;
        const int64_t *t, *created, *rc, *rrc, *rdr, *dep;
    };
    UpList U[kMaxUp];
    int n_up;
    int undecided;              // Totals::undecided bits 0 and 2, this LP
    // the Source slot the event being processed descends from, of the pending departures, and the FIFO's column (TickTables::rs_dep / rs_q)
    uint32_t lsrc, lsrcD[C];
    uint8_t *rs_qp, *rs_dp;     // null: not tracked (this LP's columns of TickTables::rs_q / rs_dep)

    // an event created by the one being processed: one step further from the group's root
    __device__ __forceinline__ void qpush(uint32_t code, int64_t pay = 0) {
        if (qn >= kQCap) { qoverflow = 1; return; }
        const int slot = (qh + qn) % kQCap;
        qmem[slot][tid] = (uint8_t)code;
        qdep[(size_t)slot * ls] = (uint8_t)(cd >= 254 ? 255 : cd + 1);
        qrc[(size_t)slot * ls] = cr;
        if constexpr (PF) {
            if (rs_qp != nullptr) rs_qp[(size_t)slot * ls] = (uint8_t)lsrc;
            if (trk) { q_rrc[(size_t)slot * ls] = rk_rc; q_rdr[(size_t)slot * ls] = rk_pack() | ((int64_t)(wk & 0x7f) << 56); q_pay[(size_t)slot * ls] = pay; }
        }
        ++qn;
    }
    __device__ __forceinline__ uint32_t qpop() {
        const uint32_t c = qmem[qh][tid];
        cd = qdep[(size_t)qh * ls];
        cr = qrc[(size_t)qh * ls];
        if constexpr (PF) {
            if (rs_qp != nullptr) lsrc = rs_qp[(size_t)qh * ls];
            if (trk) { rk_rc = q_rrc[(size_t)qh * ls]; rk_unpack(q_rdr[(size_t)qh * ls]); wk = (int32_t)(q_rdr[(size_t)qh * ls] >> 56); cur_pay = q_pay[(size_t)qh * ls]; }
        }
        qh = (qh + 1) % kQCap;
        --qn;
        return c;
    }
    // a root's own lineage in one word: steps (bits 0-7), construction rank (8-39), retargeted payloads among the steps (40-47)
    __device__ __forceinline__ int64_t rk_pack() const {
        return (int64_t)(uint32_t)(rk_dp & 0xff) | ((int64_t)(uint32_t)rk_rank << 8) | ((int64_t)(rk_wk & 0xff) << 40);
    }
    __device__ __forceinline__ void rk_unpack(int64_t w) {
        rk_dp = (int32_t)(w & 0xff); rk_rank = (int32_t)((w >> 8) & 0xffffffffll); rk_wk = (int32_t)((w >> 40) & 0xff);
    }
    // where the root stood in the breadth-first order of the nanosecond it was created in (Station::wk)
    static __device__ __forceinline__ int32_t rk_place(int64_t w) { return (int32_t)(w & 0xff) - (int32_t)((w >> 40) & 0xff); }
    __device__ __forceinline__ int32_t dp_next(int steps) const { return cd + steps > 255 ? 255 : cd + steps; }

    __device__ __forceinline__ void init_streams(uint64_t seed, uint64_t base, uint64_t ak, uint64_t sk,
                                                 double (*ring_a)[kBlock], double (*ring_s)[kBlock]) {
        key0 = (uint32_t)seed; key1 = (uint32_t)(seed >> 32);
        const uint64_t sa = stream_id(base, kStreamArrival), ss = stream_id(base, kStreamService);
        asid0 = (uint32_t)sa; asid1 = (uint32_t)(sa >> 32);
        ssid0 = (uint32_t)ss; ssid1 = (uint32_t)(ss >> 32);
        arr_k = ak; svc_k = sk;
        ra.reset(ring_a, tid); rs.reset(ring_s, tid);
        div_rate.init(rate);
        div_lambda.init(svc_lambda);
        inc_const = __ddiv_rn(1.0, rate);
    }
    // value of arrival draw k: E / rate with E = -log(1 - u)          (providers/poisson_arrival.py:31,
    //                                                                   load/arrival_time_provider.py:76)
    __device__ __forceinline__ double arr_value(double u) const { return div_rate.div(exp1_from_uniform(u)); }
    // value of service draw k: get_latency(...).to_seconds() of random.expovariate(lambda)
    //                                         (distributions/exponential.py:43, server/server.py:246-247)
    __device__ __forceinline__ double svc_value(double u) const {
        const double sample = div_lambda.div(exp1_from_uniform(u));
        if constexpr (UNI) return seconds_from_ns_d(ns_from_seconds_d(sample));   // (times below 2^51 ns: hs_engine uni_grid)
        return seconds_from_ns(ns_from_seconds(sample));      // Duration.from_seconds(sample).to_seconds()
    }
    // Append `blocks` Philox blocks (two draws each) to the ring; a ring that ends on an odd draw index
    // takes only the second half of its first block.  Needs room for 2 * blocks values.
    template <int BLOCKS>
    __device__ __forceinline__ void refill_arr() {
        const uint64_t kg = arr_k + (uint64_t)ra.n;
        const uint64_t b0 = kg >> 1;
        const bool odd = (kg & 1) != 0;
#pragma unroll
        for (int i = 0; i < BLOCKS; ++i) {
            const uint64_t b = b0 + (uint64_t)i;
            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), asid0, asid1, key0, key1);
            const double v0 = arr_value(res53(o.x, o.y)), v1 = arr_value(res53(o.z, o.w));
            if (!(i == 0 && odd)) ra.push(v0);
            ra.push(v1);
        }
    }
    template <int BLOCKS>
    __device__ __forceinline__ void refill_svc() {
        const uint64_t kg = svc_k + (uint64_t)rs.n;
        const uint64_t b0 = kg >> 1;
        const bool odd = (kg & 1) != 0;
#pragma unroll
        for (int i = 0; i < BLOCKS; ++i) {
            const uint64_t b = b0 + (uint64_t)i;
            const U4 o = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), ssid0, ssid1, key0, key1);
            const double v0 = svc_value(res53(o.x, o.y)), v1 = svc_value(res53(o.z, o.w));
            if (!(i == 0 && odd)) rs.push(v0);
            rs.push(v1);
        }
    }
    __device__ __forceinline__ bool timevarying() const { return PF && prof_kind != kProfConstant; }
    __device__ __forceinline__ bool wants_arr() const { return src_kind == 1 && A != kInfNs && ra.n == 0 && !timevarying(); }
    __device__ __forceinline__ bool wants_svc() const { return svc_kind == 0 && rs.n == 0; }
    // Wave-level top-up, called at a point where the whole wavefront is converged on the main loop: if any lane
    // has run dry, every lane with room generates kRefill more values (lanes consume at similar rates, so the
    // refills stay in step: ~93 % of the generated lanes-worth of values are used).
    __device__ __forceinline__ void top_up(bool act = true) {
        if (__any(act && wants_arr())) {
            if (src_kind == 1 && A != kInfNs && !timevarying() && ra.n <= kRing - kRefill) refill_arr<kRefill / 2>();
        }
        if (__any(act && wants_svc())) {
            if (svc_kind == 0 && rs.n <= kRing - kRefill) refill_svc<kRefill / 2>();
        }
    }

    // ---- ArrivalTimeProvider.next_arrival_time, constant-rate fast path (load/arrival_time_provider.py:72-82)
    __device__ __forceinline__ int64_t next_arrival() {
        if constexpr (PF) {
            if (prof_kind != kProfConstant) {      // general path (load/arrival_time_provider.py:84-144): the tick table; this is
                arr_time = tick_lookup(tab_a, tab_cap, generated, overflow);   // tick number `generated` (do_tick counted the one that runs)
                return arr_time;
            }
        }
        double inc;
        if (src_kind == 1) {                       // Poisson: -log(1-u) / rate
            if (ra.n == 0) refill_arr<1>();        // ran dry inside a group (rare): one block, in place
            inc = ra.pop();
            ++arr_k;
        } else inc = inc_const;                    // constant: 1.0 / rate   (providers/constant_arrival.py:23)
        const double t_next = __dadd_rn(seconds_from_ns(arr_time), inc);
        arr_time = ns_from_seconds(t_next);
        return arr_time;
    }

    // ---- service sample: get_latency(...).to_seconds() then `yield s` (server/server.py:246-250)
    __device__ __forceinline__ void sample_service(double &s, int64_t &dur_ns) {
        if (svc_kind == 0) {
            if (rs.n == 0) refill_svc<1>();
            s = rs.pop();
            ++svc_k;
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
        else { A = a2; seqA = seq++; crtA = t; dpA = dp_next(1); rcA = cr; }
        return r;
    }
    // QueuedResource.handle_event -> Queue._handle_enqueue (components/queue.py:122-147).  True: QUEUE_NOTIFY created.
    __device__ __forceinline__ bool do_enqueue(int64_t t) { return do_enqueue(t, t); }   // context["created_at"] = tick time
    __device__ __forceinline__ bool do_enqueue(int64_t t, int64_t created) {   // (a forwarded Request keeps its context: entity.py:100-105)
        ev[1]++;
        if (qcap >= 0 && buf >= qcap) { dropped++; return false; }        // FIFOQueue.push refuses (queue_policy.py:94-98)
        const bool was_empty = (buf == 0);
        if (accepted < cap) adm[accepted * ls] = created; else overflow = 1;
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
            if (C > 1) crt[i] = (k < cap) ? adm[k * ls] : 0;
            if (d == t) { D[i] = kInfNs - 1; same = (uint32_t)i + 1; }   // in-group continuation: parked, not pending
            else { D[i] = d; seqD[i] = seq++; crtD[i] = t; dpD[i] = dp_next(2); rcD[i] = cr; if constexpr (PF) { wkD[i] = wk + 1; lsrcD[i] = lsrc; } }   // QUEUE_DELIVER -> payload -> continuation
        }
        if (same) { ++cd; if constexpr (PF) ++wk; }                        // (the caller pushes the in-group continuation: deliver + 2)
        return same;
    }
    // generator resumes (server/server.py:252-273) + schedule_poll hook (queue_driver.py:79-84).
    // Returns bit0: Request@Sink created, bit1: QUEUE_POLL created.
    __device__ __forceinline__ uint32_t do_cont(int slot, int64_t t) {
        ev[6]++;
        double s = 0.0; int64_t created = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == slot) { s = svc_s[i]; created = crt[i]; D[i] = kInfNs; }
        active = active > 0 ? active - 1 : 0;
        completed++;
        total_service = __dadd_rn(total_service, s);
        uint32_t r = 0;
        if (egress == 1 || (PF && egress == kEgressServer)) {             // (forward(event, downstream): the same record, see do_sink)
            if (sink_w < cap) { sink_t[sink_w * ls] = t; if (C > 1) sink_created[sink_w * ls] = created; }
            else overflow = 1;
            sink_w++;
            r |= 1u;
        }
        if (active < conc) r |= 2u;
        return r;
    }
    // Sink.handle_event (components/common.py:36-44); the record itself was staged by do_cont.
    // A Server downstream instead (tandem): this is the moment the forwarded Request runs -- at the downstream LP, which reads
    // the record in its own pass; here its place in the nanosecond (root key, steps) joins the record and the record becomes
    // visible (`received` = forwards published; no event is counted on this LP).
    __device__ __forceinline__ void do_sink() {
        if constexpr (PF) {
            if (egress == kEgressServer) {
                if (received < cap) {
                    fw_rc[received * ls] = cr; fw_rrc[received * ls] = rk_rc; fw_rdr[received * ls] = rk_pack();
                    fw_dep[received * ls] = ((int64_t)cd << 8) | (int64_t)(cd - wk);          // steps by the key's count | places in the order
                }
                received++;
                return;
            }
        }
        ev[7]++; received++;
    }
    // A Source wired straight to a Sink/Counter (no Server in this LP): the payload IS the Sink's event.
    __device__ __forceinline__ void stage_direct_sink(int64_t t) {
        if (sink_w < cap) { sink_t[sink_w * ls] = t; if (C > 1) sink_created[sink_w * ls] = t; else adm[sink_w * ls] = t; }
        else overflow = 1;
        sink_w++;
    }

    // ---- Probe: Source.handle_event with _ProbeEventProvider, then the measurement callback ------------------
    __device__ __forceinline__ bool has_probe() const { return PF && n_probes > 0; }
    __device__ __forceinline__ int64_t probe_min() const {
        int64_t m = kInfNs;
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) if (j < n_probes && PA[j] < m) m = PA[j];
        return m;
    }
    __device__ __forceinline__ bool probe_at(int64_t t) const {
        bool any = false;
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j) any = any || (j < n_probes && PA[j] == t);
        return any;
    }
    __device__ __forceinline__ void root_probe(int j, int64_t t) {
        evp[0]++;
        qpush(Q_PSAMPLE | ((uint32_t)j << 3));                            // the daemon probe_event, created first
#pragma unroll
        for (int i = 0; i < kMaxProbes; ++i) if (i == j) {
            const int64_t k2 = p_arr[i] + 1;                              // ConstantArrivalTimeProvider over _ProbeProfile: the table
            const int64_t a2 = tick_lookup(tab_p[i], tab_cap, k2, overflow);
            p_arr[i] = k2;
            if (a2 <= t) PA[i] = kInfNs;
            else { PA[i] = a2; seqP[i] = seq++; rcP[i] = cr; crtP[i] = t; }
        }
    }
    __device__ __forceinline__ void do_probe_sample(int j, int64_t t) {
        evp[1]++;
        uint32_t metric = kProbeNone;
        int64_t pn = 0;
#pragma unroll
        for (int i = 0; i < kMaxProbes; ++i) if (i == j) { metric = p_metric[i]; pn = p_n[i]; p_n[i] = pn + 1; }
        int64_t v = 0;
        switch (metric) {
            case kProbeDepth: v = buf; break;
            case kProbeActive: v = active; break;
            case kProbeAccepted: v = accepted; break;
            case kProbeDropped: v = dropped; break;
            case kProbeCompleted: v = completed; break;
            case kProbeReceived: v = received; break;
            case kProbeGenerated: v = generated; break;
            default: break;
        }
        if (pn < pcap) { const int64_t o = ((int64_t)j * pcap + pn) * ls; probe_t[o] = t; probe_v[o] = v; } else overflow = 1;
    }

    // ---- Simulation.schedule(): the injected Event IS the Request@Server (QueuedResource.handle_event -> enqueue)
    __device__ __forceinline__ bool has_sched() const { return PF && SA != kInfNs; }
    __device__ __forceinline__ void root_sched(int64_t t) {
        ++sc_i;
        SA = sc_i < sc_end ? sc_t[sc_i] : kInfNs;
        if (do_enqueue(t)) qpush(Q_NOTIFY);
    }

    // ---- tandem: the forwards of the upstream Servers (see `trk` above)
    __device__ __forceinline__ int64_t inj_next() const {                          // earliest forward still to arrive
        int64_t m = kInfNs;
#pragma unroll
        for (int u = 0; u < kMaxUp; ++u) if (u < n_up && U[u].IA < m) m = U[u].IA;
        return m;
    }
    __device__ __forceinline__ bool has_inj() const { return PF && inj_next() != kInfNs; }
    // Two events of different origin, created at ca / cb in groups whose roots were created at ra / rb: is a pre-run event (a
    // Source's first tick, stamped t_start) involved in a way that leaves their order to sort indices no key here knows?  Either
    // one IS a first tick, or both were created in one nanosecond and one of them in a first tick's group (TickTables::t_start).
    __device__ __forceinline__ bool pre_run_tie(int64_t ca, int64_t cb, int64_t ra, int64_t rb) const {
        return ca == t_start || cb == t_start || (ca == cb && (ra == t_start || rb == t_start));
    }
    __device__ __forceinline__ int64_t inj_time(const UpList &L, int64_t k) const { return k < L.n ? L.t[k * ls] : kInfNs; }
    // cand_less on the roots of two forwards; `tie`: the whole key agrees (the roots' own ancestry would decide)
    __device__ __forceinline__ bool inj_key_less(const UpList &A, int64_t a, const UpList &B, int64_t b, bool &tie) const {
        tie = false;
        const int64_t ca = A.rc[a * ls], cb = B.rc[b * ls];
        const int64_t ra = A.rrc[a * ls], rb = B.rrc[b * ls];
        if (&A != &B && pre_run_tie(ca, cb, ra, rb)) tie = true;                   // (inside one list the list's order stands)
        if (ca != cb) return ca < cb;
        const int64_t da = A.rdr[a * ls], db = B.rdr[b * ls];
        if (rk_place(da) != rk_place(db)) return rk_place(da) < rk_place(db);
        if (ra != rb) return ra < rb;
        tie = true;
        return ((da >> 8) & 0xffffffffll) < ((db >> 8) & 0xffffffffll);
    }
    // the forward at time t whose root comes first, among those of every upstream list not taken yet: u_best < 0 none.  Inside one
    // list equal keys keep the list's order (= the upstream LP's processing order: exact); between lists a full tie is undecided.
    __device__ __forceinline__ void inj_pick(int64_t t, int &u_best, int &j_best) const {
        u_best = -1; j_best = -1;
#pragma unroll
        for (int u = 0; u < kMaxUp; ++u) {
            if (u >= n_up || U[u].IA != t) continue;
            int best = -1;
            for (int j = 0; j < kMaxInjRun && inj_time(U[u], U[u].i + j) == t; ++j) {
                if (U[u].mask & (1u << j)) continue;
                bool tie;
                if (best < 0 || (inj_key_less(U[u], U[u].i + j, U[u], U[u].i + best, tie) && !tie)) best = j;
            }
            if (best < 0) continue;
            if (u_best < 0) { u_best = u; j_best = best; continue; }
            bool tie = false, less = false;
#pragma unroll
            for (int v = 0; v < kMaxUp; ++v) if (v == u_best) less = inj_key_less(U[u], U[u].i + best, U[v], U[v].i + j_best, tie);
            if (tie) const_cast<Station *>(this)->undecided |= 1;
            if (less) { u_best = u; j_best = best; }
        }
    }
    // ... against one of this LP's own pending roots `w` (pick_root's code): true = the forward's root was created first
    __device__ __forceinline__ bool inj_before_own(const UpList &L, int64_t k, int w) const {
        int32_t dp, wkr; int64_t rc; int pad;
        own_root_key(w, dp, rc, pad, wkr);
        const int64_t ca = L.rc[k * ls], cb = root_crt(w);
        const int64_t ra = L.rrc[k * ls];
        if (pre_run_tie(ca, cb, ra, rc)) const_cast<Station *>(this)->undecided |= 1;
        if (ca != cb) return ca < cb;
        const int64_t da = L.rdr[k * ls];
        if (rk_place(da) != dp - wkr) return rk_place(da) < dp - wkr;
        if (ra != rc) return ra < rc;
        const_cast<Station *>(this)->undecided |= 1;                               // (the roots' own ancestry would decide: Totals::undecided)
        return (int32_t)((da >> 8) & 0xffffffffll) < rank_of(pad);
    }
    // lineage of the LP's own pending root `w` as the election sees it (make_candidate)
    __device__ __forceinline__ void own_root_key(int w, int32_t &dp, int64_t &rc, int &pad, int32_t &wkr) const {
        dp = 0; rc = INT64_MIN; pad = 0; wkr = 0;
        if (w == 0) { dp = dpA; rc = rcA; pad = 2; }
        else if (PF && w >= kRootXSrc && w < kRootInj) {
#pragma unroll
            for (int j = 0; j < kMaxXSrc; ++j) if (j == w - kRootXSrc) { dp = dpX[j]; rc = rcX[j]; }
            pad = 3 + (w - kRootXSrc);
        } else if (PF && w >= kRootProbe && w < kRootInj) {
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j == w - kRootProbe) rc = rcP[j];
            dp = 1; pad = 8 + (w - kRootProbe);
        } else if (PF && w == kRootSched) { dp = 0; rc = INT64_MIN; }
        else {
#pragma unroll
            for (int i = 0; i < C; ++i) if (i == w - 1) { dp = dpD[i]; rc = rcD[i]; wkr = wkD[i]; }
        }
    }
    __device__ __forceinline__ int rank_of(int pad) const {                        // cand_rank()
        if (tie_rank_p == nullptr)
            return pad >= 8 ? n_rank * (kMaxXSrc + 1) + lp * kMaxProbes + (pad - 8) : lp * (kMaxXSrc + 1) + (pad >= 2 ? pad - 2 : 0);
        if (pad >= 8) return tie_rank_p[(size_t)(kMaxXSrc + 2) * (size_t)n_rank + 1 + (size_t)(pad - 8) * (size_t)n_rank + lp];
        if (pad >= 2) return tie_rank_p[(size_t)(pad - 1) * (size_t)n_rank + lp];
        return tie_rank_p[lp];
    }
    // the forward's root, entered among this LP's roots: it does nothing here but head the chain that ends in the Request@Server
    __device__ __forceinline__ void root_inj(int uu, int j, int64_t t) {
#pragma unroll
        for (int u = 0; u < kMaxUp; ++u) {
            if (u != uu) continue;
            UpList &L = U[u];
            const int64_t k = L.i + j;
            const int64_t dv = L.dep[k * ls];
            const int32_t dep = (int32_t)(dv & 0xff), label = (int32_t)(dv >> 8);  // places in the breadth-first order | the key's steps
            cr = L.rc[k * ls]; rk_rc = L.rrc[k * ls]; rk_unpack(L.rdr[k * ls]);
            wk = label - dep; cd = wk;                                            // (the placeholders add `dep` steps: the Request arrives with `label`)
            if (dep < 1 || dep > 31) { qoverflow = 1; }                           // (a same-nanosecond chain deeper than the FIFO codes hold)
            else qpush(Q_ENQ | ((uint32_t)dep << 3), L.created[k * ls]);
            L.mask |= 1u << j;
            // the run is consumed once every forward of it has been taken
            int len = 0;
            while (len < kMaxInjRun && inj_time(L, L.i + len) == t) ++len;
            if (inj_time(L, L.i + len) == t && len == kMaxInjRun) qoverflow = 1;   // (a longer run than the mask holds)
            if (L.mask == (len >= 32 ? 0xffffffffu : ((1u << len) - 1u))) { L.i += len; L.mask = 0; L.IA = inj_time(L, L.i); }
        }
    }

    // ---- further Sources: Source.handle_event (load/source.py:142-180) of an entity of its own; its payload is one more
    // Request@Server.  Arrival times: ArrivalTimeProvider's constant-rate path, as next_arrival().
    __device__ __forceinline__ bool has_xsrc() const { return PF && n_xsrc > 0; }
    __device__ __forceinline__ int64_t xsrc_min() const {
        int64_t m = kInfNs;
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) if (j < n_xsrc && XA[j] < m) m = XA[j];
        return m;
    }
    __device__ __forceinline__ bool xsrc_at(int64_t t) const {
        bool any = false;
#pragma unroll
        for (int j = 0; j < kMaxXSrc; ++j) any = any || (j < n_xsrc && XA[j] == t);
        return any;
    }
    __device__ __forceinline__ void root_xsrc(int j, int64_t t) {
        ev[0]++;
#pragma unroll
        for (int i = 0; i < kMaxXSrc; ++i) if (i == j) {
            x_n[i]++;
            const bool payload = !(x_stop[i] >= 0 && t > x_stop[i]);     // SimpleEventProvider.get_events :68
            double area = 1.0;                                            // constant_arrival.py:23
            if (x_kind[i] == 1) {                                         // poisson_arrival.py:31
                Stream st;
                st.init(((uint64_t)key1 << 32) | key0, xsrc_stream_id(x_base, i), x_k[i]);
                area = exp1_from_uniform(st.next_uniform());
                x_k[i]++;
            }
            const int64_t a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(x_arr[i]), __ddiv_rn(area, x_rate[i])));
            x_arr[i] = a2;
            if (payload) qpush(Q_ENQ);
            if (a2 == t) { XA[i] = kInfNs; qpush(Q_TICK | ((uint32_t)(i + 1) << 3)); }
            else if (a2 < t) XA[i] = kInfNs;                              // popped later as "time travel" and dropped
            else { XA[i] = a2; seqX[i] = seq++; crtX[i] = t; dpX[i] = dp_next(1); rcX[i] = cr; }
        }
    }

    // ---- chains with at most one event in flight (fast path pieces) ---------------------------
    // returns true if the general FIFO must take over (a same-time continuation was created)
    // (cd = the QUEUE_POLL's steps from the group's root)
    __device__ __forceinline__ bool chain_from_poll(int64_t t) {
        if (!do_poll()) return false;
        ++cd;                                                             // the QUEUE_DELIVER it created
        const uint32_t same = do_deliver_work(t);
        if (same) { qpush(Q_CONT | ((same - 1) << 3)); return true; }
        return false;
    }
    __device__ __forceinline__ bool chain_from_enqueue(int64_t t) {   // (cd = the Request's steps from the group's root)
        if (!do_enqueue(t)) return false;
        ++cd;
        if (!do_notify()) return false;
        ++cd;
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
        if constexpr (PF) {
            // Totals::undecided bit 2: a pre-run root (its group's root stamp is "before the run") beside any other root of t
            int cnt = 0;
            bool pre = false;
            if (SA == t) { ++cnt; pre = true; if (sc_i + 1 < sc_end && sc_t[sc_i + 1] == t) ++cnt; }
            if (A == t) { ++cnt; pre = pre || rcA == INT64_MIN; }
#pragma unroll
            for (int i = 0; i < C; ++i) cnt += D[i] == t ? 1 : 0;
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j < n_probes && PA[j] == t) { ++cnt; pre = pre || rcP[j] == INT64_MIN; }
#pragma unroll
            for (int j = 0; j < kMaxXSrc; ++j) if (j < n_xsrc && XA[j] == t) { ++cnt; pre = pre || rcX[j] == INT64_MIN; }
            if (pre && (cnt >= 2 || (n_up > 0 && inj_next() == t))) const_cast<Station *>(this)->undecided |= 4;
        }
        if constexpr (PF) {
            if (SA == t) {
                if (sc_idx == nullptr) return kRootSched;
                best = kRootSched; bs = sc_idx[sc_i];            // its true sort index (after the prologue: hs_exact.hpp)
            }
        }
        if (A == t && (best < 0 || (int32_t)(seqA - bs) < 0)) { best = 0; bs = seqA; }
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (D[i] == t && (best < 0 || (int32_t)(seqD[i] - bs) < 0)) { best = 1 + i; bs = seqD[i]; }
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j)
                if (j < n_probes && PA[j] == t && (best < 0 || (int32_t)(seqP[j] - bs) < 0)) { best = kRootProbe + j; bs = seqP[j]; }
#pragma unroll
            for (int j = 0; j < kMaxXSrc; ++j)
                if (j < n_xsrc && XA[j] == t && (best < 0 || (int32_t)(seqX[j] - bs) < 0)) { best = kRootXSrc + j; bs = seqX[j]; }
            if (n_up > 0 && best > 0 && A == t && crtA == t_start) const_cast<Station *>(this)->undecided |= 1;   // (a first tick beside a
                                                                                                                // departure: pre_run_tie)
            if (n_up > 0 && inj_next() == t) {      // tandem: the roots of the forwards arriving now compete by the election key
                int u, j;
                inj_pick(t, u, j);
                if (u >= 0) {
                    bool first = best < 0;
#pragma unroll
                    for (int v = 0; v < kMaxUp; ++v) if (v == u && !first) first = inj_before_own(U[v], U[v].i + j, best);
                    if (first) best = kRootInj + 32 * u + j;
                }
            }
        }
        return best;
    }
    // creation time of pending root `which` (pick_root's code)
    __device__ __forceinline__ int64_t root_crt(int which) const {
        int64_t c = INT64_MIN;                                            // kRootSched: constructed before run()
        if (PF && which >= kRootInj) {
            const int u = (which - kRootInj) >> 5, j = (which - kRootInj) & 31;
            int64_t c = INT64_MIN;
#pragma unroll
            for (int v = 0; v < kMaxUp; ++v) if (v == u) c = U[v].rc[(U[v].i + j) * ls];
            return c;
        }
        if (which == 0) c = crtA;
        else if (PF && which >= kRootXSrc && which < kRootInj) {
#pragma unroll
            for (int j = 0; j < kMaxXSrc; ++j) if (j == which - kRootXSrc) c = crtX[j];
        } else if (PF && which >= kRootProbe && which < kRootInj) {
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j == which - kRootProbe) c = crtP[j];
        } else if (!(PF && which == kRootSched)) {
#pragma unroll
            for (int i = 0; i < C; ++i) if (i == which - 1) c = crtD[i];
        }
        return c;
    }
    __device__ __forceinline__ void run_root(int which, int64_t t) {
        cd = 0; cr = root_crt(which);                                     // a root: pending from an earlier nanosecond
        if constexpr (PF) {
            lsrc = 255u;                                                  // a tick descends from its own Source, a departure inherits
            if (which == 0) lsrc = 0u;
            else if (which >= kRootXSrc && which < kRootXSrc + kMaxXSrc) lsrc = 1u + (uint32_t)(which - kRootXSrc);
            else if (which >= 1 && which <= C) {
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == which - 1) lsrc = lsrcD[i];
            }
            if (trk) {
                if (which >= kRootInj) { root_inj((which - kRootInj) >> 5, (which - kRootInj) & 31, t); return; }
                int pad;
                own_root_key(which, rk_dp, rk_rc, pad, rk_wk);
                rk_rank = rank_of(pad);
                wk = 0;
            }
        }
        if (which == 0) root_tick(t);
        else if (PF && which >= kRootXSrc) root_xsrc(which - kRootXSrc, t);
        else if (PF && which >= kRootProbe) root_probe(which - kRootProbe, t);
        else if (PF && which == kRootSched) root_sched(t);
        else root_cont(which - 1, t);
    }

    // general in-group FIFO drain
    __device__ __forceinline__ void drain(int64_t t) {
        while (qn > 0) {
            const uint32_t code = qpop();
            switch (code & 7u) {
                case Q_ENQ: {
                    const uint32_t hop = PF ? (code >> 3) : 0u;          // tandem: a forward `hop` steps above its Request@Server
                    if (hop > 1) qpush(Q_ENQ | ((hop - 1) << 3), cur_pay);
                    else if (do_enqueue(t, hop == 1 ? cur_pay : t)) qpush(Q_NOTIFY);
                } break;
                case Q_NOTIFY: if (do_notify()) qpush(Q_POLL); break;
                case Q_POLL: if (do_poll()) qpush(Q_DELIVER); break;
                case Q_DELIVER: {
                    // do_poll already popped the buffer; deliver + work
                    const uint32_t same = do_deliver_work(t);
                    if (same) qpush(Q_CONT | ((same - 1) << 3));
                } break;
                case Q_TICK:
                    if (PF && (code >> 3) != 0) { if constexpr (PF) root_xsrc((int)(code >> 3) - 1, t); }
                    else root_tick(t);
                    break;
                case Q_CONT: root_cont((int)(code >> 3), t); break;
                case Q_SINK: do_sink(); break;
                case Q_PSAMPLE: if constexpr (PF) do_probe_sample((int)(code >> 3), t); break;
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
        if constexpr (PF) { if (has_probe()) { const int64_t pm = probe_min(); if (pm < t) t = pm; } if (SA < t) t = SA; }
        if constexpr (PF) { if (has_xsrc()) { const int64_t xm = xsrc_min(); if (xm < t) t = xm; } }
        if constexpr (PF) { if (n_up > 0) { const int64_t ia = inj_next(); if (ia < t) t = ia; } }
        return t;
    }

    __device__ __forceinline__ void run_group(int64_t t) {
        int n_at = (A == t) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < C; ++i) n_at += (D[i] == t) ? 1 : 0;
        if constexpr (PF) { if (has_probe() && probe_at(t)) n_at += 2; }  // a probe tick: always the general path
        if constexpr (PF) { if (SA == t) n_at += 2; }                     // so is a scheduled Request
        if constexpr (PF) { if (has_xsrc() && xsrc_at(t)) n_at += 2; }    // and a tick of one of the LP's further Sources
        if (n_at == 1 && !force_general) {
            // Fast path: one event in flight at a time.  Both kinds of root converge on ONE poll/deliver/work
            // site so that a wavefront whose lanes mix ticks and departures executes the (expensive) service
            // draw once per iteration, not once per branch.
            bool general = false, want_poll = false;
            cd = 0;
            if (A == t) {
                cr = crtA;
                if constexpr (PF) lsrc = 0u;
                const uint32_t r = do_tick(t);
                if (svc_kind == 2) {             // Source -> Sink directly
                    if ((r & 1u) && egress == 1) { stage_direct_sink(t); do_sink(); }
                    if (r & 2u) { qpush(Q_TICK); general = true; }
                }
                else if (r & 2u) { if (r & 1u) qpush(Q_ENQ); qpush(Q_TICK); general = true; }
                else if (r & 1u) { want_poll = do_enqueue(t) && do_notify(); cd = 3; }   // tick -> Request -> QUEUE_NOTIFY -> QUEUE_POLL
            } else {
                int slot = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (D[i] == t) slot = i;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == slot) { cr = crtD[i]; if constexpr (PF) lsrc = lsrcD[i]; }
                const uint32_t r = do_cont(slot, t);
                if (r & 1u) do_sink();
                want_poll = (r & 2u) != 0;
                cd = 1;                                                  // continuation -> QUEUE_POLL
            }
            if (want_poll) general = chain_from_poll(t);
            if (general) drain(t);
        } else {
            run_group_general(t);
        }
        last_time = t;
    }

    // ---- C == 1: one timestamp group per lane per call, as straight-line predicated code -----------------
    // Same semantics as run_group()'s fast path (exactly one pending event at `t`, at most one event in flight),
    // but without per-lane branches: a wavefront whose lanes mix ticks and departures runs ONE instruction
    // stream with selects instead of every branch body under a different exec mask.  Everything that would
    // leave that regime is detected BEFORE any state is committed (`slow`) and handed to run_group():
    // A == D ties, stop_after reached, a next tick that truncates onto / before `t`, a zero-length service,
    // a Source wired straight to a Sink.  `act` = this lane still has a group with t <= end_ns.
    // top_up() has run: every lane that may consume a stream value has at least one in its ring.
    __device__ __forceinline__ void step_c1(int64_t t, bool act) {
        static_assert(C == 1, "step_c1 is the single-slot specialisation");
        const bool tick = (A == t);
        const bool dep = !tick;
        // speculative next arrival (load/arrival_time_provider.py:72-82) and service sample (server.py:246-250)
        const double inc = (src_kind == 1) ? ra.peek() : inc_const;
        const int64_t a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(arr_time), inc));
        const double s_new = (svc_kind == 0) ? rs.peek() : svc_const_s;
        const int64_t dur = (svc_kind == 0) ? ns_from_seconds(s_new) : svc_const_ns;
        // which reference events happen in this group
        const bool enq_drop = qcap >= 0 && buf >= qcap;                 // FIFOQueue.push refuses (queue_policy.py:94-98)
        const bool acc = tick && !enq_drop;
        const bool notify = acc && buf == 0;                            // queue.py:124,144-146
        const int32_t active_dep = active > 0 ? active - 1 : 0;
        const bool poll = (notify && active < conc) || (dep && active_dep < conc);   // queue_driver.py:94-99 / :79-84
        const int64_t buf_enq = buf + (acc ? 1 : 0);
        const bool deliver = poll && buf_enq > 0;                       // queue.py:149-166
        const bool slow = act && (force_general || svc_kind == 2 ||
                                  (PF && (prof_kind != kProfConstant || (has_probe() && probe_at(t)) || (has_sched() && SA == t) ||
                                          (has_xsrc() && xsrc_at(t)) || n_up > 0)) ||     // (a rare root at t, forwards: run_group)
                                  (tick && D[0] == t) ||
                                  (tick && ((stop_ns >= 0 && t > stop_ns) || a2 <= t)) || (deliver && dur == 0));
        const bool fast = act && !slow;
        const bool tick_f = fast && tick, dep_f = fast && dep, acc_f = fast && acc;
        const bool poll_f = fast && poll, del_f = fast && deliver;
        // Source.handle_event
        ev[0] += tick_f; generated += tick_f;
        arr_time = tick_f ? a2 : arr_time;
        A = tick_f ? a2 : A;
        seqA = tick_f ? seq : seqA;
        // lineage: the next tick is one step from this one; a service that starts is six steps from a tick (Request, QUEUE_NOTIFY,
        // QUEUE_POLL, QUEUE_DELIVER, the payload, the continuation), four from the departure that freed the worker
        const int64_t root_c = tick ? crtA : crtD[0];
        rcA = tick_f ? crtA : rcA;
        dpA = tick_f ? 1 : dpA;
        rcD[0] = del_f ? root_c : rcD[0];
        dpD[0] = del_f ? (tick ? 6 : 4) : dpD[0];
        crtA = tick_f ? t : crtA;
        seq += tick_f ? 1u : 0u;
        const bool pop_a = tick_f && src_kind == 1;
        ra.advance_if(pop_a);
        arr_k += pop_a ? 1u : 0u;
        // Queue._handle_enqueue / QueueDriver._handle_notify
        ev[1] += tick_f;
        dropped += (tick_f && enq_drop) ? 1 : 0;
        if (acc_f) { if (accepted < cap) adm[accepted * ls] = t; else overflow = 1; }
        accepted += acc_f;
        ev[2] += (fast && notify) ? 1u : 0u;
        // generator resumes: statistics, Sink record, schedule_poll hook
        ev[6] += dep_f;
        completed += dep_f;
        total_service = dep_f ? __dadd_rn(total_service, svc_s[0]) : total_service;
        active = dep_f ? active_dep : active;
        const bool sink_f = dep_f && egress == 1;
        if (sink_f) { if (sink_w < cap) sink_t[sink_w * ls] = t; else overflow = 1; }
        sink_w += sink_f; ev[7] += sink_f; received += sink_f;
        // QUEUE_POLL, then QUEUE_DELIVER + the retargeted payload at the worker
        ev[3] += poll_f;
        buf = fast ? (buf_enq - (deliver ? 1 : 0)) : buf;
        ev[4] += del_f; ev[5] += del_f;
        started += del_f;
        active += del_f ? 1 : 0;
        svc_s[0] = del_f ? s_new : svc_s[0];
        D[0] = del_f ? (t + dur) : (dep_f ? kInfNs : D[0]);
        seqD[0] = del_f ? seq : seqD[0];
        crtD[0] = del_f ? t : crtD[0];
        seq += del_f ? 1u : 0u;
        const bool pop_s = del_f && svc_kind == 0;
        rs.advance_if(pop_s);
        svc_k += pop_s ? 1u : 0u;
        last_time = fast ? t : last_time;
        if (slow) run_group(t);
    }

    // ---- C == 1, unbounded FIFO: the LP in REQUEST order instead of event order ---------------------------
    // With one worker and a FIFO buffer, request k starts at S_k = max(a_k, D_{k-1}) and departs at
    // D_k = S_k + service_k, and the service draws are consumed in arrival order.  Every reference event of
    // request k happens at a_k (SourceEvent, Request@Server, QUEUE_NOTIFY iff the buffer was empty = request k-1
    // had started: S_{k-1} < a_k, QUEUE_POLL iff the worker was idle: D_{k-1} < a_k), at S_k (QUEUE_DELIVER,
    // Request@worker) or at D_k (ProcessContinuation, Request@Sink, the completion QUEUE_POLL), and is processed
    // by `_execute_until(T)` iff that time is <= T.  One iteration therefore handles one whole request -- one
    // arrival value, one service value, no tick/departure divergence between lanes -- and counts the same events
    // the event-order loop counts.  Anything whose outcome depends on the order of two events at the SAME
    // nanosecond (a_k == S_{k-1}, a_k == D_{k-1}, a next tick that truncates onto / before a_k, a zero-length
    // service) makes the lane `bail`: its state is reloaded and re-run by the event-order loop (step_c1), which
    // reproduces the reference's creation-order tie-breaking.  Equal timestamps that cannot change any outcome
    // (a_k == D_j, j < k-1: request k-1 is still waiting either way) are left alone.
    struct ReqCursor {
        int64_t T;                  // end of the window
        int64_t Sprev, Dprev;       // start / departure of the previous request in FIFO order (kInfNs: not started)
        int64_t nb;                 // requests that arrived in an earlier launch and are still waiting
        int64_t pendD, pendS;       // the request in service when the window ends
        double pend_s;
        int64_t lt;                 // time of the latest processed event
        uint32_t n_tick, n_notify, n_poll, n_start, n_dep;
        bool pend, blocked, bail, done;
        int64_t crtA0;              // creation time of the tick that was pending when the window began (lineage, req_finish)
        double arr_d;               // UNI: ArrivalTimeProvider.current_time as a binary64 (whole ns below 2^52: exact)
        // PF, an LP with ONE Probe (slot 0): its ticks inside the window, sampled from the request sequence (req_probe_*).  A sample
        // needs generated(T) -- known once an arrival beyond T is seen -- and completed(T) -- known once a departure beyond T is
        // seen; the two become known in either order, so each has its own pointer into the tick table and a third one finalises.
        int64_t pTa, pTd;           // time of the next tick of the arrival / the departure pointer (kInfNs: none inside the window)
        uint32_t ma, md, mf;        // ticks passed by the arrival pointer, the departure pointer, finalised
        int64_t p_prev;             // time of the last tick the arrival pointer passed (a tick that does not advance ends the Probe)
    };
    __device__ __forceinline__ bool req_eligible() const {
        return C == 1 && !force_general && qn == 0 && conc == 1 && qcap < 0 && stop_ns < 0 && svc_kind != 2 &&
               !(PF && (prof_kind != kProfConstant || n_probes > 1 || has_sched() || has_xsrc() || n_up > 0)) &&
               (egress == 0 || egress == 1) && !(buf > 0 && active == 0) && active <= 1;
    }
    // ---- a Probe in request order (one per LP).  Between two events of its LP a sample is a function of two counts:
    //     generated(T) = accepted(T) = #{a_k <= T},  completed(T) = #{D_k <= T},  started(T) = min(generated, completed + 1)
    // (one worker, FIFO, unbounded: request k starts at max(a_k, D_{k-1})), depth = generated - started, active = started -
    // completed.  A tick ON the nanosecond of an arrival or a departure is decided by sort indices: the lane bails to the
    // event-order loop, which is also where the pre-run first tick's coincidences are reported (Totals::undecided bit 2).
    __device__ __forceinline__ int64_t req_probe_time(uint32_t m) {                 // tick number p_arr[0] + m of the table
        return tick_lookup(tab_p[0], tab_cap, p_arr[0] + (int64_t)m, overflow);
    }
    __device__ __forceinline__ void req_probe_begin(ReqCursor &c) {
        c.ma = c.md = c.mf = 0;
        c.pTa = c.pTd = n_probes == 1 ? PA[0] : kInfNs;
        c.p_prev = INT64_MIN;
        if (c.pTa > c.T) c.pTa = c.pTd = kInfNs;
    }
    // the arrival pointer passes every tick before `limit` (the arrival being processed, or T + 1 at the end of the window)
    __device__ __forceinline__ void req_probe_arrivals(ReqCursor &c, bool on, int64_t limit, int64_t gen_now) {
        while (on && c.pTa < limit) {
            const int64_t o = ((int64_t)p_n[0] + (int64_t)c.ma) * ls;
            if (p_n[0] + (int64_t)c.ma < pcap) probe_v[o] = gen_now; else overflow = 1;
            c.p_prev = c.pTa;
            ++c.ma;
            int64_t nx = req_probe_time(c.ma);
            if (nx <= c.p_prev) { c.bail = true; nx = kInfNs; }          // (a tick that does not advance ends the Probe: event order)
            c.pTa = nx <= c.T ? nx : kInfNs;
        }
    }
    __device__ __forceinline__ void req_probe_departures(ReqCursor &c, bool on, int64_t limit, int64_t comp_now) {
        while (on && c.pTd < limit) {
            const int64_t o = ((int64_t)p_n[0] + (int64_t)c.md) * ls;
            if (p_n[0] + (int64_t)c.md < pcap) probe_t[o] = comp_now;
            ++c.md;
            const int64_t nx = req_probe_time(c.md);
            c.pTd = nx <= c.T ? nx : kInfNs;
        }
    }
    __device__ __forceinline__ void req_probe_finalise(ReqCursor &c, bool on) {
        while (on && c.mf < (c.ma < c.md ? c.ma : c.md)) {
            const int64_t pn = p_n[0] + (int64_t)c.mf;
            if (pn < pcap) {
                const int64_t o = pn * ls;
                __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the two halves were stored by this lane ...
                const int64_t g_rel = __hip_atomic_load(&probe_v[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (... past its L1)
                const int64_t c_rel = __hip_atomic_load(&probe_t[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t acc = accepted + g_rel, cn = completed + c_rel;
                const int64_t stt = acc < cn + 1 ? acc : cn + 1;
                int64_t v = 0;
                switch (p_metric[0]) {
                    case kProbeDepth: v = acc - stt; break;
                    case kProbeActive: v = stt - cn; break;
                    case kProbeAccepted: v = acc; break;
                    case kProbeDropped: v = dropped; break;
                    case kProbeCompleted: v = cn; break;
                    case kProbeReceived: v = egress == 1 ? received + c_rel : received; break;
                    case kProbeGenerated: v = generated + g_rel; break;
                    default: break;
                }
                probe_t[o] = req_probe_time(c.mf);
                probe_v[o] = v;
            }
            ++c.mf;
        }
    }
    __device__ __forceinline__ void req_count_departure(ReqCursor &c, bool p, int64_t d, double s) {
        total_service = p ? __dadd_rn(total_service, s) : total_service;
        if (p && HSG(egress == 1, true)) {
            const int64_t w = sink_w + (int64_t)c.n_dep;
            if (w < cap) sink_t[w * ls] = d; else overflow = 1;
        }
        c.n_dep += p ? 1u : 0u;
        c.lt = (p && d > c.lt) ? d : c.lt;
    }
    __device__ __forceinline__ void req_begin(ReqCursor &c, int64_t T) {
        c.T = T; c.nb = buf; c.lt = last_time; c.crtA0 = crtA; c.arr_d = (double)arr_time;
        c.n_tick = c.n_notify = c.n_poll = c.n_start = c.n_dep = 0;
        c.blocked = c.bail = c.done = false;
        const bool busy = active > 0;
        c.Sprev = busy ? crtD[0] : INT64_MIN;
        c.Dprev = busy ? D[0] : INT64_MIN;
        const bool dep0 = busy && D[0] <= T;      // the request already in service departs inside the window
        req_count_departure(c, dep0, D[0], svc_s[0]);
        c.pend = busy && !dep0;
        c.pendD = D[0]; c.pendS = crtD[0]; c.pend_s = svc_s[0];
        if constexpr (PF) {
            req_probe_begin(c);
            if (n_probes == 1 && busy) {                  // ticks before the departure that was pending: nothing has completed yet
                req_probe_departures(c, true, D[0], 0);
                c.bail = c.bail || c.pTd == D[0];         // (a tick on its nanosecond: event order)
            }
        }
    }
    __device__ __forceinline__ void req_step(ReqCursor &c, bool act) {
        const int64_t T = c.T;
        const bool bk = c.nb > 0 && !c.blocked;          // next in FIFO order: a request that is already waiting
        const bool arr = !bk && A <= T;                  // ... or the next arrival
        const bool fin = !bk && !arr;
        // arrival part at a_k = A
        const double inc = HSG(src_kind == 1, true) ? ra.peek() : inc_const;
        const double a2d = UNI ? ns_from_seconds_d(__dadd_rn(seconds_from_ns_d(c.arr_d), inc)) : 0.0;
        const int64_t a2 = UNI ? i64_from_whole_d(a2d) : ns_from_seconds(__dadd_rn(seconds_from_ns(arr_time), inc));
        const bool tie_a = arr && (A == c.Sprev || A == c.Dprev || a2 <= A);
        const bool notify = arr && c.Sprev < A;
        const bool idle = arr && c.Dprev < A;
        // start part at S_k
        const int64_t Sk = bk ? c.Dprev : (A > c.Dprev ? A : c.Dprev);
        const bool st = (bk || arr) && Sk <= T;
        const double s_new = HSG(svc_kind == 0, true) ? rs.peek() : svc_const_s;
        const int64_t dur = UNI ? i64_from_whole_d(ns_from_seconds_d(s_new)) : (HSG(svc_kind == 0, true) ? ns_from_seconds(s_new) : svc_const_ns);
        const int64_t Dk = Sk + dur;
        const bool dp = st && Dk <= T;                   // departure part at D_k
        bool tie_p = false;
        if constexpr (PF) {
            const bool pon = act && n_probes == 1;
            if (__any(pon)) {                            // the LP's Probe: ticks before this arrival / this departure (req_probe_*)
                req_probe_arrivals(c, pon && (arr || fin), arr ? A : T + 1, (int64_t)c.n_tick);
                req_probe_departures(c, pon && (st || fin), st ? Dk : T + 1, (int64_t)c.n_dep);
                tie_p = pon && ((arr && c.pTa == A) || (st && c.pTd == Dk));
            }
        }
        const bool bail = act && (tie_a || tie_p || (st && dur == 0));
        const bool go = act && !bail && !fin;
        c.bail = c.bail || bail;
        c.done = c.done || (act && fin);
        const bool arr_g = go && arr, st_g = go && st, dp_g = go && dp;
        // Source.handle_event + Queue._handle_enqueue (+ notify / poll when the buffer is empty / the worker idle)
        if (arr_g) {
            const int64_t w = accepted + (int64_t)c.n_tick;
            if (w < cap) adm[w * ls] = A; else overflow = 1;
        }
        c.n_tick += arr_g ? 1u : 0u;
        c.n_notify += (go && notify) ? 1u : 0u;
        c.n_poll += (go && idle) ? 1u : 0u;
        c.lt = (arr_g && A > c.lt) ? A : c.lt;
        crtA = arr_g ? A : crtA;
        arr_time = arr_g ? a2 : arr_time;
        if constexpr (UNI) c.arr_d = arr_g ? a2d : c.arr_d;
        const bool pop_a = arr_g && HSG(src_kind == 1, true);
        ra.advance_if(pop_a);
        arr_k += pop_a ? 1u : 0u;
        // QUEUE_DELIVER + Request@worker: the service sample is drawn at the start of service
        c.n_start += st_g ? 1u : 0u;
        c.lt = (st_g && Sk > c.lt) ? Sk : c.lt;
        const bool pop_s = st_g && HSG(svc_kind == 0, true);
        rs.advance_if(pop_s);
        svc_k += pop_s ? 1u : 0u;
        // ProcessContinuation + Request@Sink + completion poll
        req_count_departure(c, dp_g, Dk, s_new);
        // cursor
        c.pend = st_g ? !dp : c.pend;
        c.pendD = st_g ? Dk : c.pendD;
        c.pendS = st_g ? Sk : c.pendS;
        c.pend_s = st_g ? s_new : c.pend_s;
        // the request just looked at becomes "the previous request" -- also a waiting one that cannot start any
        // more (the worker stays busy beyond T): from then on nobody starts and every arrival finds a non-empty buffer
        const bool looked = go && (arr || bk);
        c.Sprev = looked ? (st ? Sk : kInfNs) : c.Sprev;
        c.Dprev = looked ? (st ? Dk : kInfNs) : c.Dprev;
        c.blocked = c.blocked || (go && bk && !st);
        c.nb -= (go && bk && st) ? 1 : 0;
        A = arr_g ? a2 : A;                              // last: `A` is read above
        if constexpr (PF) {
            const bool pon = act && !bail && n_probes == 1;
            if (__any(pon && c.mf < (c.ma < c.md ? c.ma : c.md))) req_probe_finalise(c, pon);
        }
    }
    // fold the window's deltas into the LP state exactly as the event-order loop would have left it
    __device__ __forceinline__ void req_finish(const ReqCursor &c) {   // (not const: the Probe's table look-ups can flag an overflow)
        ev[0] += c.n_tick; ev[1] += c.n_tick; ev[2] += c.n_notify; ev[3] += c.n_poll + c.n_dep;
        ev[4] += c.n_start; ev[5] += c.n_start; ev[6] += c.n_dep;
        generated += c.n_tick; accepted += c.n_tick; started += c.n_start; completed += c.n_dep;
        if (egress == 1) { ev[7] += c.n_dep; received += c.n_dep; sink_w += c.n_dep; }
        buf += (int64_t)c.n_tick - (int64_t)c.n_start;   // waiting: every arrival is admitted, every start takes one
        active = c.pend ? 1 : 0;
        D[0] = c.pend ? c.pendD : kInfNs;
        crtD[0] = c.pend ? c.pendS : crtD[0];
        svc_s[0] = c.pend ? c.pend_s : svc_s[0];
        // the Probe's ticks of the window (req_probe_*): two events each; the pending tick was created by the last of them, whose
        // own creation time (the tick before it) is the pending tick's group root
        uint32_t n_pt = 0;
        int64_t last_probe = INT64_MIN;
        if constexpr (PF) {
            n_pt = n_probes == 1 ? c.ma : 0u;
            if (n_pt != 0u) {
                evp[0] += n_pt; evp[1] += n_pt;
                const int64_t t_last = req_probe_time(n_pt - 1);
                last_probe = t_last;
                rcP[0] = n_pt >= 2u ? req_probe_time(n_pt - 2) : crtP[0];
                crtP[0] = t_last;
                const int64_t nx = req_probe_time(n_pt);
                PA[0] = nx > t_last ? nx : kInfNs;
                p_arr[0] += (int64_t)n_pt;
                p_n[0] += (int64_t)n_pt;
            }
        }
        // creation stamps: only their order matters (pick_root).  The pending departure was created at pendS, the
        // pending tick at crtA; a tick that also started the service created the next tick first.
        // (No two of this window's events shared a timestamp -- the lane would have bailed -- so times decide; stamps
        // that were not re-created in this window keep their order.)
        if ((c.n_tick | c.n_start | n_pt) != 0u) {
            const bool d_first = c.pend && c.pendS < crtA;
            uint32_t sA = d_first ? 1u : 0u, sD = d_first ? 0u : 1u, sP = 2u;
            if constexpr (PF) {
                if (n_probes == 1) {                      // the pending probe tick among them, by creation time (ties: it was there first
                    const int64_t tP = crtP[0];           // only if neither of the others was re-created after it)
                    const bool p_before_a = tP < crtA, p_before_d = !c.pend || tP < c.pendS;
                    sP = (p_before_a ? 0u : 1u) + ((c.pend && !p_before_d) ? 1u : 0u);
                    sA += p_before_a ? 1u : 0u;
                    sD += p_before_d ? 1u : 0u;
                    seqP[0] = seq + sP;
                }
            }
            seqA = seq + sA;
            seqD[0] = seq + sD;
            seq += 3u;
        }
        // Lineage of what is pending now (only the election beyond end_time reads it, so it is reconstructed HERE, from the
        // admission log and the service stream, instead of being carried through the loop).  Every tick was a root of its own group
        // (a tie made the lane bail), and in this regime every tick is admitted: adm[k] is the time of tick k.
        if (c.n_tick != 0u) {       // the pending tick was created by tick number accepted - 1, itself created at tick accepted - 2
            dpA = 1;
            rcA = accepted >= 2 ? adm[(accepted - 2) * ls] : c.crtA0;
        }
        if (c.pend && c.n_start != 0u) {      // the request in service is number m = started - 1; it started at pendS
            const int64_t m = started - 1;
            const int64_t a_m = m < cap ? adm[m * ls] : 0;
            if (c.pendS == a_m) {             // ... on arrival: six steps from tick m, which was created at tick m - 1
                dpD[0] = 6;
                rcD[0] = m >= 1 ? adm[(m - 1) * ls] : c.crtA0;
            } else {                          // ... when request m - 1 left: four steps from that continuation, created when IT started
                double s_prev = svc_const_s;
                if (HSG(svc_kind == 0, true)) {
                    Stream st;
                    st.init(((uint64_t)key1 << 32) | key0, ((uint64_t)ssid1 << 32) | ssid0, (uint64_t)(m - 1));
                    s_prev = svc_value(st.next_uniform());
                }
                dpD[0] = 4;
                rcD[0] = c.pendS - ns_from_seconds(s_prev);
            }
        }
        last_time = c.lt > last_probe ? c.lt : last_probe;
    }
};

}  // namespace hs
#undef HSG
