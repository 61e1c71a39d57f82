// hs_netstation.hpp -- station LPs that exchange requests over links: conservative time windows.
//
// Extends the station LP of hs_station.hpp to networks built from the reference's components:
//
//   [Source] -> Server_i -> { nothing | Sink_i | NetworkLink -> Server_j | RandomRouter([Sink_i | NetworkLink ...]) }
//
// (components/random_router.py:32-45, components/network/link.py:114-189).  This is the engine-side
// replacement of the reference's partitioned execution (`parallel/coordinator.py:75-172`: EXECUTE all
// partitions to T+W, EXCHANGE outboxes, ADVANCE), with one LP per lane instead of one partition per thread:
//
//   * A request travelling over a link is a MESSAGE {arrival ns, send ns, created_at ns, link}.  Its transit
//     time is >= the link's constant latency, so with W = min over links of that latency every message sent
//     inside a window [T, T+W) arrives at or after T+W: all LPs can process a whole window independently
//     (`PartitionLink.min_latency > 0` plays the same role in the reference, parallel/link.py:41-45).
//   * Messages are appended to the DESTINATION LP's incoming bag with one device-scope atomic slot
//     reservation; bags are double-buffered by window parity, and the owner merges the previous window's
//     bag at the start of the next launch (kernel boundary = the exchange barrier).
//   * The link's continuation event (transit over) and the Request it creates at the egress Server both
//     happen at the arrival timestamp; both are processed -- and counted -- by the destination LP.
//
// Same-timestamp order follows creation order as in hs_station.hpp; a message's creation stamp is its send
// time.  Local events precede a message created in the same nanosecond (cross-LP creation order inside one
// nanosecond is the one piece of the reference's global sort index that is not reconstructed).
#pragma once

#include "hs_station.hpp"

// HSU(x, v): a per-LP configuration predicate `x` that is the compile-time constant `v` in the UNI instantiation of NetStation
// (every station of the network is Source.poisson -> Server(Exp, c = 1, unbounded) -> RandomRouter -> one NetworkLink with
// exponential jitter and no loss; the host checks it, hs_engine_set_network).  Those predicates otherwise live as lane masks in
// SGPR pairs that spill to VGPR lanes, and every use is a select: the specialised kernel is 14 % faster on the headline ring.
#define HSU(x, v) (UNI ? (v) : (x))
namespace hs {

enum : uint32_t { EG_NONE = 0, EG_SINK = 1, EG_LINK = 2, EG_ROUTER = 3 };

// Losses decided by a TABLE (round 6: PartitionLink.packet_loss, parallel/coordinator.py:203-205 -- the coordinator's ONE
// `random.Random(seed)` drawn in exchange order, which the host replays over the run's cross-partition sends,
// happy_simulator_amd/parallel.py): packet number e of link l is lost iff bit drop_off[l] + e of drop_bits is set; every packet
// that enters such a link is logged so that the host can order the sends of a run.  One object in device memory behind ONE
// pointer of NetParams: the asynchronous kernels keep every kernel argument they touch in scalar registers, and the six words
// this would add to the argument structs cost hs_net_async<1, false, true> 5 % on the headline ring (spills), for a path it
// never takes.
struct LossTables {
    const int64_t *drop_off;      // [n_links + 1] bit offsets
    const uint32_t *drop_bits;
    int64_t *send_log;            // {send time ns, network-wide link id, packet number} of every packet that entered such a link
    unsigned long long *send_log_n;   // ... since the last reset, in no particular order
    int64_t send_log_cap;
    int *overflow_word;           // Totals::overflow (bit 32: a loss table or the send log is too short)
};

struct NetParams {
    const uint8_t *egress;        // [n_lp] EG_*
    const int32_t *rt0, *rt1;     // [n_lp] router targets in RandomRouter(targets=[...]) order: -1 = the LP's Sink, else link
    const int32_t *rt2, *rt3;     // [n_lp] ... the third and fourth target of routers with more than two
    const uint8_t *rt_cnt;        // [n_lp] len(targets), 1..4: the route draw picks targets[int(u * len(targets))]
    const int32_t *link_of;       // [n_lp] EG_LINK: the link
    const uint64_t *route_base;   // [n_lp] stream base of the router entity
    int32_t n_links;
    const int32_t *link_dst;      // [n_links] destination LP
    const double *link_lat_min;   // [n_links] ConstantLatency(latency) seconds
    const uint8_t *link_jit_kind; // [n_links] 0 = ExponentialLatency(link_jit_mean) jitter, 1 = ConstantLatency(link_jit_mean) jitter (0: none)
    const double *link_jit_mean;  // [n_links]
    const uint64_t *link_base;    // [n_links] stream base of the link entity
    const double *link_loss;      // [n_links] NetworkLink.packet_loss_rate (0 = lossless); kLossTable: decided by the table below
    const int32_t *link_gid;      // [n_links] network-wide link id (tie-break key); null = the index itself
    // Losses decided by a TABLE (round 6: PartitionLink.packet_loss): one pointer, see LossTables (null: no link has one)
#ifndef HS_NO_LOSS_TABLES   // (scratch build: what the argument costs the asynchronous kernels)
    const struct LossTables *loss_tables;
#endif
    // asynchronous engine (hs_net_async): incoming links of every LP (CSR) and each link's transit floor in ns
    const int32_t *in_off;        // [n_lp + 1]
    const int32_t *in_links;      // [n_links] link ids grouped by destination LP
    const int64_t *link_lat_ns;   // [n_links] from_seconds(to_seconds(from_seconds(lat_min))): no transit is shorter
};

// A network partitioned over several engines (one per GPU): links whose destination station lives on another
// engine put their messages into an OUTBOX row per destination rank instead of the destination's bag; the host
// exchanges the rows between the ranks after every window (torch.distributed all-to-all over RCCL) and the
// receiver injects them into its bags (hs_shard_inject).  Window ends follow the global virtual time:
// wend = min(end, max(prev_wend + 1, GVT) + W - 1), GVT = all-reduce(min) over the ranks' earliest pending work.
struct ShardCtl {
    int64_t *wend_slots;          // [2] window end of launch k at [k & 1]; null = unsharded (host passes wend)
    const int64_t *gvt_in;        // GVT after the previous window (all-reduced)
    int64_t *gvt_out;             // this rank's earliest pending work after this window (atomicMin)
    int64_t end_ns, W, lp_base;
    int64_t *outbox;              // [world][row] : row = {count, kMsgWords x int64 per message ...}
    int64_t *cand_out;            // [8] {valid, t, t_created, global lp, steps from its group's root, that root's creation time,
                                  //      construction rank among THIS shard's entities, what the candidate is (Candidate::pad)}
                                  //      of the rank's first event beyond end_ns (the election's key)
    const int32_t *link_rank;     // [n_links] rank that owns the link's destination station
    int32_t msg_cap, row, rank, world;
    // LIVE exchange (round 6): no launch boundary per round -- a station whose link leaves the shard appends to the link's queue in
    // the DESTINATION rank's memory (peer-mapped: hipIpcOpenMemHandle / xGMI peer access) and publishes the link's (bound, tail)
    // word there, with system-scope stores, while both ranks' hs_net_async kernels run; the receiver polls its own memory and
    // writes the positions it has taken where the sender reads them.  `live` != 0: per rank r the base of its link-queue records,
    // words and positions, and per link the index it has in its destination rank's link table.
    int32_t live;
    int64_t *const *peer_rec;     // [world] NetState::aq_rec of rank r
    int64_t *const *peer_ea;      // [world] NetState::aq_ea
    unsigned long long *const *peer_head;   // [world] NetState::aq_head
    const int32_t *peer_link;     // [n_links] the link's index at its destination rank (-1: the destination is here)
};

struct NetState {
    uint64_t *route_k;            // [n_lp] route draws consumed
    int64_t *routed;              // [n_lp] RandomRouter.stats_routed
    uint64_t *link_k;             // [n_links] jitter draws consumed (touched only by the link's source LP)
    int64_t *link_in;             // [n_links] requests that entered the link (Request@Link events)
    int64_t *link_sent;           // [n_links] of those, the ones not lost (link_in - link_sent = packets_dropped); also the
                                  //           sequence number of the link's asynchronous queue
    int64_t *link_packets;        // [n_links] NetworkLink.packets_sent (touched only by the destination LP)
    int64_t *next_time;           // [n_lp] earliest pending local event or bagged message
    // current bag (owner only)
    int32_t *bag_cnt;             // [n_lp]
    int64_t *bag_t, *bag_ts, *bag_cr;   // [n_lp][bag_cap]
    int64_t *bag_lin;             // [n_lp][bag_cap] the message's lineage (lin_pack)
    int32_t *bag_link;            // [n_lp][bag_cap]
    // incoming bags, double-buffered by window parity: [2][n_lp] / [2][n_lp][bag_cap]
    int32_t *in_cnt;
    int64_t *in_t, *in_ts, *in_cr, *in_lin;
    int32_t *in_link;
    int32_t bag_cap;
    // asynchronous engine: one single-producer / single-consumer message queue per link.  Every word below is written
    // with write-through (sc1) agent-scope stores and read with agent-scope loads -- the data is its own flag
    // (cdna_hip_programming.md section 6, Guideline 16, recipe R2); the producer drains its stores (vmcnt(0)) between
    // the payload, `aq_tail` and `aq_ea`, so a consumer that sees a value of `aq_ea` also sees every message below it.
    // [n_links][aq_cap] records of four words {arrival ns, send ns, created_at ns, lineage (lin_pack)}: one message = one 32-byte
    // sector (round 3; rounds 1-2 kept four arrays, i.e. four write-through stores to four different lines per message and four
    // lines read per message: 3.7-5.2 x the algorithmic HBM traffic)
    int64_t *aq_rec;
    unsigned long long *aq_tail;     // [n_links] messages appended so far (producer)
    unsigned long long *aq_head;     // [n_links] messages taken so far (consumer; the producer reads it for flow control)
    int64_t *aq_ea;                  // [n_links] ONE word per link = (bound << 20 | messages appended so far mod 2^20):
                                     //   bound: every message NOT yet appended arrives at or after pk_base + bound ns
                                     //   (2^44 - 1 = never); a consumer gets a consistent (bound, tail) pair from a single load
    int64_t pk_base;                 // = start_ns
    // pre-sent departures (hs_net_async, one-worker stations): see NetStation::early_upto
    int64_t *early_upto;             // [n_lp]
    int64_t *d_pre;                  // [n_lp]
    int64_t *pend_pay;               // [n_lp] windows: the ENQ payload of the group the last election stopped inside (NetStation::pending_pack)
    int32_t aq_cap;               // entries per link queue, a power of two (slot = sequence number & (aq_cap - 1))
    int32_t aq_on;                   // 1 inside hs_net_async / its final launch: send_link uses the queues
};

// agent-scope accesses of the words LPs of different workgroups exchange (write-through store / cache-bypassing load)
__device__ __forceinline__ void ag_store(int64_t *p, int64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ag_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one link-queue record = four words = two 16-byte agent-scope stores into one 32-byte sector (NetState::aq_rec)
__device__ __forceinline__ void ag_store_rec(int64_t *rec, int64_t w0, int64_t w1, int64_t w2, int64_t w3) {
#ifdef HS_REC_WORDS
    ag_store(rec, w0); ag_store(rec + 1, w1); ag_store(rec + 2, w2); ag_store(rec + 3, w3);
#else
    typedef long long v2i64 __attribute__((ext_vector_type(2)));
    const v2i64 lo = {w0, w1}, hi = {w2, w3};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(rec), "v"(lo), "v"(hi) : "memory");
#endif
}
__device__ __forceinline__ int64_t ag_load(const int64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and into ANOTHER rank's memory (ShardCtl::live): system scope
__device__ __forceinline__ void sys_store_rec(int64_t *rec, int64_t w0, int64_t w1, int64_t w2, int64_t w3) {
    typedef long long v2i64 __attribute__((ext_vector_type(2)));
    const v2i64 lo = {w0, w1}, hi = {w2, w3};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" ::"v"(rec), "v"(lo), "v"(hi) : "memory");
}
// Round 6: a link whose two ends are lanes of ONE wavefront (63 of 64 links of the ring) moves its records at WORKGROUP scope: the
// CU's own L1 / L2 path is coherent for them, so the record is a plain write-back line -- two records fill one 64-byte line, written
// to HBM once when the L2 evicts it -- instead of two 16-byte write-through transactions per record and a fetch past the L2 per
// read (measured on the headline ring: WRITE_SIZE 2.8 GB against 1.05 GB of records and log appends).  The order between the
// sender's stores and the receiver's loads is the iteration boundary's vmcnt(0) drain, as for the device-scope path.
__device__ __forceinline__ void wg_store_rec(int64_t *rec, int64_t w0, int64_t w1, int64_t w2, int64_t w3) {
    typedef long long v2i64 __attribute__((ext_vector_type(2)));
    const v2i64 lo = {w0, w1}, hi = {w2, w3};
    asm volatile("global_store_dwordx4 %0, %1, off sc0\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0" ::"v"(rec), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ int64_t wg_load(const int64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the bound word of a link: everything the producer made visible before it published this value must be seen by the
// loads that FOLLOW this one (the tail, the payload) -- an acquire, not just program order: two relaxed loads of different
// addresses may be serviced in either order
__device__ __forceinline__ int64_t ag_load_acquire(const int64_t *p) {
#ifdef HS_NO_ACQ
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ unsigned long long ag_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the packed (bound, tail) word of a link queue (NetState::aq_ea)
constexpr int kPkTailBits = 20;
constexpr unsigned long long kPkTailMask = (1ull << kPkTailBits) - 1ull;
constexpr int64_t kPkNever = (1ll << 44) - 1;                   // bound field: nothing will ever be appended again
__device__ __forceinline__ int64_t pk_pack(int64_t ea, unsigned long long tail, int64_t base) {
    int64_t rel = ea == kInfNs ? kPkNever : ea - base;
    rel = rel < 0 ? 0 : (rel > kPkNever - 1 && ea != kInfNs ? kPkNever - 1 : rel);   // clamping DOWN is conservative
    return (int64_t)(((unsigned long long)rel << kPkTailBits) | (tail & kPkTailMask));
}
__device__ __forceinline__ int64_t pk_ea(int64_t w, int64_t base) {
    const int64_t rel = (int64_t)((unsigned long long)w >> kPkTailBits);
    return rel == kPkNever ? kInfNs : base + rel;
}
__device__ __forceinline__ unsigned long long pk_tail(int64_t w, unsigned long long head) {   // head <= tail < head + 2^20
    return head + (((unsigned long long)w - head) & kPkTailMask);
}

// Is packet number `entered` of link l lost?  NetworkLink.packet_loss_rate (components/network/link.py:131-138): u of the link's
// LOSS stream, one draw per packet that enters.  kLossTable: the table of NetParams::drop_off (PartitionLink.packet_loss).
constexpr double kLossTable = 2.0;
__device__ __forceinline__ bool link_loses(const NetParams &np, uint64_t seed, int32_t l, int64_t entered, int64_t t_send) {
    const double loss = np.link_loss[l];
    if (loss <= 1.0) {
        Stream ls;
        ls.init(seed, stream_id(np.link_base[l], kStreamLoss), (uint64_t)entered);
        return ls.next_uniform() < loss;
    }
#ifdef HS_NO_LOSS_TABLES
    (void)t_send;
    return false;
#else
    const LossTables lt = *np.loss_tables;
    const unsigned long long pos = atomicAdd(lt.send_log_n, 1ull);
    if (pos < (unsigned long long)lt.send_log_cap) {
        lt.send_log[3 * pos] = t_send; lt.send_log[3 * pos + 1] = np.link_gid ? np.link_gid[l] : (int64_t)l; lt.send_log[3 * pos + 2] = entered;
    } else atomicOr(lt.overflow_word, 32);
    const int64_t b0 = lt.drop_off[l], nb = lt.drop_off[l + 1] - b0;
    if (entered >= nb) { atomicOr(lt.overflow_word, 32); return false; }
    const int64_t b = b0 + entered;
    return ((lt.drop_bits[b >> 5] >> (b & 31)) & 1u) != 0u;
#endif
}

constexpr int kEnqPay = 8;   // ENQ payload FIFO depth (general path only)
constexpr int kMsgWords = 5; // int64 words per message in an outbox / inbox row: arrival, send time, created_at, dst << 32 | link, lineage

// LINEAGE of a message (the NetworkLink's continuation event, components/network/link.py:114-154): how many steps after the root
// of the group it was created in, and when that root was created (hs_station.hpp StationState::dpA) -- one word:
// steps << 56 | "the root was constructed before run()" << 55 | (send time - root's creation time).
__device__ __forceinline__ int64_t lin_pack(int32_t steps, int64_t root_crt, int64_t t_send) {
    const uint64_t d = (uint64_t)(steps > 255 ? 255 : steps) << 56;
    if (root_crt == INT64_MIN) return (int64_t)(d | (1ull << 55));
    return (int64_t)(d | ((uint64_t)(t_send - root_crt) & ((1ull << 55) - 1ull)));
}
__device__ __forceinline__ int32_t lin_steps(int64_t lin) { return (int32_t)((uint64_t)lin >> 56); }
__device__ __forceinline__ int64_t lin_root(int64_t lin, int64_t t_send) {
    return (((uint64_t)lin >> 55) & 1ull) ? INT64_MIN : t_send - (int64_t)((uint64_t)lin & ((1ull << 55) - 1ull));
}

// ---- FAST instantiation (hs_net_async only) ------------------------------------------------------------------------
// The asynchronous engine runs a wavefront's event groups in a divergent loop with only a few lanes active per trip
// (measured on the 65 536-station ring: ~3 of 64), so whatever a group costs is paid almost per LANE.  Three things kept
// that cost high and are removed here without touching the results:
//   * random draws (Philox + hs_log) ran inline in the divergent branches -> the E = -log(1-u) values of the arrival,
//     service and link-jitter streams and the router's choices are pre-drawn by the WHOLE wavefront into small per-lane
//     rings (LDS / a register of bits) at a converged point (top_up), exactly as hs_station.hpp does;
//   * the bag of pending messages lived in global memory (a chain of ~1 us loads per message) -> LDS columns;
//   * the state of the LP's outgoing link (counters, parameters) lived in global memory -> registers.
// Draws are pure functions of (stream, index) and are consumed in the same order, so nothing observable changes.
constexpr int kLBag = 8;     // LDS bag entries per LP (a full bag leaves messages in their queue: async_receive)
#ifndef HS_LOOK
#define HS_LOOK 4
#endif
constexpr int kLookMax = HS_LOOK;   // completions / services a sender's bound looks ahead over (pre-drawn route / service draws)
constexpr int kNRing = 8;    // pre-drawn values per stream per LP
struct NetFastLds {
    double (*ring_a)[kBlock];
    double (*ring_s)[kBlock];
    double (*ring_j)[kBlock];
    int64_t (*bag_t)[kBlock];
    int64_t (*bag_ts)[kBlock];
    int64_t (*bag_cr)[kBlock];
    int64_t (*bag_lin)[kBlock];
    int32_t (*bag_link)[kBlock];
    int64_t (*crc)[kBlock];       // created_at of the NEXT kNRing requests to start: ordinals [started, started + kNRing), slot = ordinal mod kNRing
};

// PF: the station may carry a Probe, a time-varying arrival profile or Requests injected with Simulation.schedule() -- the
// rare roots.  The windowed engine always has them; the asynchronous engine has a second instantiation for networks
// that use any of them, so that the common one keeps its registers.
template <int C, bool FAST = false, bool PF = !FAST, bool UNI = false>
struct NetStation {
    // parameters
    int lp, n;
    uint32_t src_kind, svc_kind, egress;
    int32_t conc, rt0, rt1, rt2, rt3, rtk, link_of;   // rtk = len(router.targets)
    double rate, svc_lambda, svc_const_s;
    int64_t stop_ns, qcap, svc_const_ns;
    uint64_t seed, route_base;
    // state
    int64_t A, crtA, arr_time, buf, generated, accepted, dropped, completed, rejected, started, received, routed;
    double arr_d;                 // UNI: arr_time as a binary64 (exact: whole ns below 2^52)
    uint32_t seqA, seq;
    int32_t active;
    int64_t D[C], crtD[C], crt[C];
    uint32_t seqD[C];
    double svc_s[C];
    double total_service;
    int64_t last_time;
    // lineage of the pending events and of the event being processed (hs_station.hpp)
    int32_t dpA, dpD[C], cd;
    int64_t rcA, rcD[C], cr, rcP[kMaxProbes];
    uint8_t *qdep; int64_t *qrc;  // the in-group FIFO's lineage columns (global memory): entry `slot` of this LP at [slot * ls]
    Stream arr, svc, rte;
    uint32_t ev[11];
    // Probe attached to this station (instrumentation/probe.py:81-164), as in hs_station.hpp: a daemon Source of its own
    // whose ticks sample one attribute (PF instantiations).
    uint32_t p_metric[kMaxProbes], seqP[kMaxProbes];
    int64_t PA[kMaxProbes], crtP[kMaxProbes], p_arr[kMaxProbes], p_n[kMaxProbes], pcap;   // p_arr: index of the pending tick in tab_p
    const int64_t *tab_p[kMaxProbes];   // the Probes' tick tables (hs_tables.hpp)
    int64_t *probe_t, *probe_v;     // slot j's log starts at probe_t + j * pcap * ls
    int n_probes;
    uint32_t evp[2];
    // further Sources feeding this station's Server (PF), as in hs_station.hpp: entities of their own, constant or Poisson rate.
    // Their state stays in the engine's arrays (a rare root: registers are the scarce resource of these kernels -- with it in
    // registers hs_net_window<1> went from 451 to 512 VGPRs plus scratch spills); only the earliest pending tick is cached.
    const StationParams *xp;
    const StationState *xx;
    int64_t xs_min;
    uint64_t x_base;
    int n_xsrc, x_n_lp;
    // time-varying arrival rate of this station's Source (load/profile.py); 0 = constant.  Tick k of such a Source is tab_a[k]
    // (the tick table, hs_tables.hpp); it draws nothing here.
    uint32_t prof_kind;
    const int64_t *tab_a;
    int64_t tab_cap;
    // Simulation.schedule(): Requests injected before run() (hs_station.hpp); they precede every run-time event of their ns
    int64_t SA, sc_i, sc_end;
    int undecided;                // Totals::undecided bit 2, this LP (run_group)
    const int64_t *sc_t;
    const uint32_t *sc_idx;       // true sort indices of the injected Requests (after the prologue, hs_exact.hpp), or null
    // logs
    int64_t *adm, *sink_t, *sink_created;   // record k at [k * lgs] (RecordLogs::lp_major: 1 for network engines, else n_lp)
    int64_t cap, ls, lgs;                   // ls: stride of the other per-LP columns ([slot][n_lp]: in-group FIFO lineage, probe logs)
    int overflow, qoverflow, bagoverflow;
    // network
    const NetParams *np;
    const NetState *ns;
    const ShardCtl *sc;
    int64_t sent_min;             // earliest arrival among the messages this LP sent in this window
    bool sent_async;              // asynchronous engine: a message was appended since the last publication of aq_ea
    int64_t undrained;            // asynchronous engine: earliest possible arrival among messages left in a queue (bag full)
    int send_idx;
    int32_t bag_n;
    int64_t bmin;                 // min over the bag's arrival times (kInfNs: empty)
    int bh;                       // FAST: the LDS bag is a circular buffer SORTED by arrival time; logical entry i sits in slot
                                  // (bh + i) & (kLBag - 1), entry 0 is the earliest: taking the next message is O(1), no scans
    // FAST: pre-drawn values (ring head / count per stream), router choices as bits, the one outgoing link in registers
    NetFastLds fl;
    int ha, na, hs_, nsv, hj, nj, rn;
    uint32_t rbits;
    int32_t fl_link;              // the LP's only outgoing link (-1: none or two -> the global-memory path)
    int32_t fl_dst;
    bool fl_remote;               // ... whose destination lives on another shard: messages go to that rank's outbox row
    uint32_t fl_jit;              // 0 = exponential jitter
    double fl_delay0, fl_lam, fl_loss, inc_const;
    int64_t fl_in, fl_sent;
    // PRE-SENDING (C == 1, FAST, the one outgoing link in registers, lossless).  With one worker and a FIFO buffer the whole
    // future of a request is fixed the moment it is admitted: it starts at S = max(arrival, departure of the request before
    // it) and leaves at D = S + its service time -- service draw number `ordinal`, route draw number `ordinal`, the link's next
    // jitter draw: all pure functions of indices that are known now.  So the message it will become is appended to the link's
    // queue AT ADMISSION (or at the start of service when the pre-drawn rings do not reach that far), with the same
    // {arrival, send time D, created_at} it would carry if it were sent when the completion is processed -- the completion then
    // only counts the reference's events.  The receiver sees the same messages in the same order, just earlier in wall-clock
    // time, and the sender's bound on the link starts from the departure of the LAST admitted request instead of its clock:
    // the lookahead grows by the station's whole backlog.
    //   early_upto  requests with ordinal < early_upto have had their way out decided and (if it is the link) their message
    //               appended; completed <= early_upto <= accepted
    //   D_pre       departure time of request early_upto - 1 (any time <= now once that request has completed)
    //   fl_q        messages appended to the link's queue so far (the queue's sequence number; fl_sent is the statistic)
    int64_t early_upto, D_pre, fl_q, end_ns;
    bool presend;
    Stream jit;
    // FAST: the created_at window.  Slot j & 7 of `crc` holds request j for the ordinals [started, win_hi) -- a contiguous prefix of
    // the next kNRing requests to start.  An admission inside [started, started + kNRing) writes its slot directly (and extends
    // the prefix when it is the next ordinal); what a deeper queue admitted comes back from the admission log in HBM at the TOP
    // of an iteration (window_fill: one round trip per iteration at a point where the wavefront is converged and the previous
    // iteration's stores have drained).  History: a synchronous read at delivery time stalled 64 % of the group trips; a
    // "prefetch" issued at the start and committed an iteration later was no better -- the compiler needs the loaded register for
    // the conditional assignment right away and gfx9 has one counter for loads and stores, so the trip waited for the load AND
    // for every write-through store before it (measured: ~2 400 of a trip's ~7 600 cycles).
    int64_t win_hi;
    unsigned long long tail_hint; // asynchronous engine: the in-wavefront sender's appended count (0: none)
    int32_t fi_link;              // the LP's only incoming link (-1: none or several): its packets_sent counter in a register
    int64_t fi_packets;
    unsigned long long fi_head;   // ... and this LP's position in that link's queue (the only writer of aq_head[fi_link])
    int32_t fi_unpub;             // ... messages taken since aq_head[fi_link] was last written (head_taken)
    bool fl_local, fi_local;      // hs_net_async: the outgoing / the incoming link's other end is a lane of this wavefront (wg_store_rec)
    // in-group FIFO (an LDS column) + ENQ payloads (an LDS column, or global memory in hs_net_async: entry k at enq[k * enq_stride])
    uint8_t (*qmem)[kBlock];
    int64_t *enq;
    size_t enq_stride;
    int tid, qh, qn, ph, pn;
#ifdef HS_RINGSTAT   // scratch statistics build (never defined in the shipped library)
    int stat_gl, stat_slow;
#endif
#ifdef HS_CYC2       // scratch build: cycles inside step1 (classification / arrival side / departure side / start of service)
    unsigned long long cy2[4];
#define HS_CY2(k) { const unsigned long long now_ = __builtin_readcyclecounter(); cy2[k] += now_ - cy_t; cy_t = now_; }
#elif defined(HS_MARK)
#define HS_CY2(k) asm volatile("; HSMARK step1_" #k ::: "memory");
#else
#define HS_CY2(k)
#endif

    // an event created by the one being processed: one step further from the group's root (hs_station.hpp)
    __device__ __forceinline__ void qpush(uint32_t code) {
        if (qn >= kQCap) { qoverflow = 1; return; }
        const int slot = (qh + qn) % kQCap;
        qmem[slot][tid] = (uint8_t)code;
        qdep[(size_t)slot * ls] = (uint8_t)(cd >= 254 ? 255 : cd + 1);
        qrc[(size_t)slot * ls] = cr;
        ++qn;
    }
    __device__ __forceinline__ uint32_t qpop() {
        const uint32_t c = qmem[qh][tid];
        cd = qdep[(size_t)qh * ls];
        cr = qrc[(size_t)qh * ls];
        qh = (qh + 1) % kQCap;
        --qn;
        return c;
    }
    __device__ __forceinline__ int32_t dp_next(int steps) const { return cd + steps > 255 ? 255 : cd + steps; }
    __device__ __forceinline__ void push_enq(int64_t created) {
        if (pn >= kEnqPay) { qoverflow = 1; return; }
        enq[(size_t)((ph + pn) % kEnqPay) * enq_stride] = created;
        ++pn;
        qpush(Q_ENQ);
    }
    __device__ __forceinline__ int64_t pop_enq_payload() {
        const int64_t v = enq[(size_t)ph * enq_stride];
        ph = (ph + 1) % kEnqPay;
        --pn;
        return v;
    }

    // ---- bag (pending inbound messages of this LP; global memory, owner-only)
    __device__ __forceinline__ size_t bidx(int i) const { return (size_t)lp * ns->bag_cap + i; }
    // FAST: the bag lives in LDS (kLBag entries); when it is full, further messages wait in their link's queue, whose
    // capacity follows bag_capacity (async_receive / `undrained`)
    __device__ __forceinline__ int bag_capacity() const {
        if constexpr (FAST) return kLBag < ns->bag_cap ? kLBag : ns->bag_cap;
        else return ns->bag_cap;
    }
    static_assert((kLBag & (kLBag - 1)) == 0, "the LDS bag is addressed with a mask");
    __device__ __forceinline__ int bslot(int i) const { return (bh + i) & (kLBag - 1); }
    __device__ __forceinline__ int64_t bg_t(int i) const { if constexpr (FAST) return fl.bag_t[bslot(i)][tid]; else return ns->bag_t[bidx(i)]; }
    __device__ __forceinline__ int64_t bg_ts(int i) const { if constexpr (FAST) return fl.bag_ts[bslot(i)][tid]; else return ns->bag_ts[bidx(i)]; }
    __device__ __forceinline__ int64_t bg_cr(int i) const { if constexpr (FAST) return fl.bag_cr[bslot(i)][tid]; else return ns->bag_cr[bidx(i)]; }
    __device__ __forceinline__ int32_t bg_link(int i) const { if constexpr (FAST) return fl.bag_link[bslot(i)][tid]; else return ns->bag_link[bidx(i)]; }
    __device__ __forceinline__ int64_t bg_lin(int i) const { if constexpr (FAST) return fl.bag_lin[bslot(i)][tid]; else return ns->bag_lin[bidx(i)]; }
    __device__ __forceinline__ void bg_set(int i, int64_t t, int64_t ts, int64_t created, int32_t l, int64_t lin) {
        if constexpr (FAST) { const int k = bslot(i); fl.bag_t[k][tid] = t; fl.bag_ts[k][tid] = ts; fl.bag_cr[k][tid] = created; fl.bag_link[k][tid] = l; fl.bag_lin[k][tid] = lin; }
        else { const size_t d = bidx(i); ns->bag_t[d] = t; ns->bag_ts[d] = ts; ns->bag_cr[d] = created; ns->bag_link[d] = l; ns->bag_lin[d] = lin; }
    }
    // FAST: insert in arrival-time order (stable: behind equal times).  Messages of one link arrive almost in send order,
    // so the new one usually goes last: one comparison.
    __device__ __forceinline__ void bag_insert(int64_t t, int64_t ts, int64_t created, int32_t l, int64_t lin) {
        int p = bag_n;
        while (p > 0 && bg_t(p - 1) > t) { bg_set(p, bg_t(p - 1), bg_ts(p - 1), bg_cr(p - 1), bg_link(p - 1), bg_lin(p - 1)); --p; }
        bg_set(p, t, ts, created, l, lin);
        ++bag_n;
        bmin = t < bmin ? t : bmin;
    }
    // earliest arrival in the bag, kept in a register (`bmin`): next_time() runs several times per step
    __device__ __forceinline__ int64_t bag_scan_min() const {
        if constexpr (FAST) return bag_n > 0 ? bg_t(0) : kInfNs;
        int64_t m = kInfNs;
        for (int i = 0; i < bag_n; ++i) { const int64_t t = bg_t(i); m = t < m ? t : m; }
        return m;
    }
    __device__ __forceinline__ int64_t bag_min() const { return bmin; }
    __device__ __forceinline__ void bag_remove(int i) {
        if constexpr (FAST) {
            if (i == 0) bh = (bh + 1) & (kLBag - 1);                  // the earliest message: the common case
            else for (int k = i; k + 1 < bag_n; ++k) bg_set(k, bg_t(k + 1), bg_ts(k + 1), bg_cr(k + 1), bg_link(k + 1), bg_lin(k + 1));
            --bag_n;
            bmin = bag_scan_min();
            return;
        }
        const int last = bag_n - 1;
        if (i != last) bg_set(i, bg_t(last), bg_ts(last), bg_cr(last), bg_link(last), bg_lin(last));
        bag_n = last;
        bmin = bag_scan_min();
    }

    // ---- pre-drawn values (FAST).  Stream::k counts GENERATED draws; consumed = k - ring count (store_net).
    // The rings hold the values as the handlers use them: the arrival increment E / rate, the service time
    // to_seconds(from_seconds(E / lambda)), the jitter to_seconds(from_seconds(E / lambda_link)) -- the divisions and ns
    // truncations run here, for 64 lanes at once, not in the divergent group loop.
    // UNI (the host checks it, hs_engine_set_network): every time is a whole number of ns in [0, 2^51), exact in binary64 -- the time
    // algebra stays in fp64 registers (hs_device.hpp ns_from_seconds_d: v_trunc_f64 instead of the f64 <-> i64 conversion sequences)
    __device__ __forceinline__ int64_t ns_i(double x) const { return UNI ? i64_from_whole_d(ns_from_seconds_d(x)) : ns_from_seconds(x); }
    __device__ __forceinline__ double sec_rt(double x) const {           // Duration.from_seconds(x).to_seconds()
        return UNI ? seconds_from_ns_d(ns_from_seconds_d(x)) : seconds_from_ns(ns_from_seconds(x));
    }
    __device__ __forceinline__ double svc_value(double e) const { return sec_rt(__ddiv_rn(e, svc_lambda)); }
    __device__ __forceinline__ void refill_a(int m) {
        for (int i = 0; i < m; ++i) {
            const double e = exp1_from_uniform(arr.next_uniform());
            fl.ring_a[(ha + na) & (kNRing - 1)][tid] = __ddiv_rn(e, rate); ++na;
        }
    }
    __device__ __forceinline__ void refill_s(int m) {
        for (int i = 0; i < m; ++i) {
            fl.ring_s[(hs_ + nsv) & (kNRing - 1)][tid] = svc_value(exp1_from_uniform(svc.next_uniform())); ++nsv;
        }
    }
    __device__ __forceinline__ void refill_j(int m) {
        for (int i = 0; i < m; ++i) {
            const double sample = __ddiv_rn(exp1_from_uniform(jit.next_uniform()), fl_lam);
            fl.ring_j[(hj + nj) & (kNRing - 1)][tid] = sec_rt(sample); ++nj;
        }
    }
    // the wave-level refills of top_up: four values in straight-line code (Stream::next4) -- on one wavefront per SIMD the four
    // logarithms and quotients overlap instead of waiting for one another
    __device__ __forceinline__ void refill_a4() {
        double u[4];
        arr.next4(u);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = __ddiv_rn(exp1_from_uniform(u[i]), rate);
#pragma unroll
        for (int i = 0; i < 4; ++i) fl.ring_a[(ha + na + i) & (kNRing - 1)][tid] = u[i];
        na += 4;
    }
    __device__ __forceinline__ void refill_a2() {
        double u[2];
        arr.next2(u);
#pragma unroll
        for (int i = 0; i < 2; ++i) u[i] = __ddiv_rn(exp1_from_uniform(u[i]), rate);
#pragma unroll
        for (int i = 0; i < 2; ++i) fl.ring_a[(ha + na + i) & (kNRing - 1)][tid] = u[i];
        na += 2;
    }
    __device__ __forceinline__ void refill_s2() {
        double u[2];
        svc.next2(u);
#pragma unroll
        for (int i = 0; i < 2; ++i) u[i] = svc_value(exp1_from_uniform(u[i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) fl.ring_s[(hs_ + nsv + i) & (kNRing - 1)][tid] = u[i];
        nsv += 2;
    }
    __device__ __forceinline__ void refill_j2() {
        double u[2];
        jit.next2(u);
#pragma unroll
        for (int i = 0; i < 2; ++i) u[i] = sec_rt(__ddiv_rn(exp1_from_uniform(u[i]), fl_lam));
#pragma unroll
        for (int i = 0; i < 2; ++i) fl.ring_j[(hj + nj + i) & (kNRing - 1)][tid] = u[i];
        nj += 2;
    }
    __device__ __forceinline__ void refill_s4() {
        double u[4];
        svc.next4(u);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = svc_value(exp1_from_uniform(u[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) fl.ring_s[(hs_ + nsv + i) & (kNRing - 1)][tid] = u[i];
        nsv += 4;
    }
    __device__ __forceinline__ void refill_j4() {
        double u[4];
        jit.next4(u);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = sec_rt(__ddiv_rn(exp1_from_uniform(u[i]), fl_lam));
#pragma unroll
        for (int i = 0; i < 4; ++i) fl.ring_j[(hj + nj + i) & (kNRing - 1)][tid] = u[i];
        nj += 4;
    }
    // RandomRouter (components/random_router.py:32-45, Philox-plugged): targets[int(u * len(targets))].  Pre-drawn decisions are
    // kept two bits each in `rbits` (16 of them), oldest in the low bits.
    __device__ __forceinline__ int32_t rt_target(int idx) const { return idx == 0 ? rt0 : idx == 1 ? rt1 : idx == 2 ? rt2 : rt3; }
    __device__ __forceinline__ int route_draw() { return (int)__dmul_rn(rte.next_uniform(), (double)rtk); }
    __device__ __forceinline__ void refill_r(int m) {
        for (int i = 0; i < m; ++i) { rbits |= (uint32_t)(route_draw() & 3) << (2 * rn); ++rn; }
    }
    __device__ __forceinline__ double arr_inc() {                     // E / rate of the next Poisson arrival
        if constexpr (FAST) {
            if (na == 0) refill_a(2);
            const double v = fl.ring_a[ha][tid];
            ha = (ha + 1) & (kNRing - 1); --na;
            return v;
        } else {
            return __ddiv_rn(exp1_from_uniform(arr.next_uniform()), rate);
        }
    }
    __device__ __forceinline__ double svc_s_next() {                  // service time of the next start, seconds
        if constexpr (FAST) {
            if (nsv == 0) refill_s(2);
            const double v = fl.ring_s[hs_][tid];
            hs_ = (hs_ + 1) & (kNRing - 1); --nsv;
            return v;
        } else return svc_value(exp1_from_uniform(svc.next_uniform()));
    }
    __device__ __forceinline__ int route_idx() {
        if constexpr (FAST) {
            if (rn == 0) refill_r(2);
            const int idx = (int)(rbits & 3u);
            rbits >>= 2; --rn;
            return idx;
        } else return route_draw();
    }
    __device__ __forceinline__ uint64_t arr_consumed() const { return FAST ? arr.k - (uint64_t)na : arr.k; }
    __device__ __forceinline__ uint64_t svc_consumed() const { return FAST ? svc.k - (uint64_t)nsv : svc.k; }
    __device__ __forceinline__ uint64_t rte_consumed() const { return FAST ? rte.k - (uint64_t)rn : rte.k; }
    // Wave-level top-up at a converged point: when some lane has run dry, every lane with room draws 4 more values
    // (2 Philox blocks) -- 64 lanes at the price the divergent loop would pay for one
    // (`need` = what one iteration may consume per stream: the groups-per-iteration cap)
    // (`parts`: 1 = the arrival and service streams, 2 = the jitter stream and the router's decisions -- hs_net_async runs the two
    //  halves around the point at which the incoming link's word has arrived, each hiding one memory round trip)
    __device__ __forceinline__ void top_up(bool act, int need, int parts = 3) {
        if constexpr (FAST) {
            const bool wa = HSU(src_kind == 1, true) && A != kInfNs && !(PF && prof_kind != kProfConstant), ws = HSU(svc_kind == 0, true), wj = fl_link >= 0 && HSU(fl_jit == 0, true);
            const bool wr = HSU(egress == EG_ROUTER, true);
            // A top-up costs the wavefront what its NEEDIEST lane draws, and one is due in almost every iteration (some lane of 64 is
            // always below the mark): so it is one Philox block = two values for every lane with room, and a second block only when a
            // lane is down to fewer than two values (HS_TOPUP4: four values at a time, as until round 5).
#ifdef HS_TOPUP4
            if (parts & 1) {
                if (__any(act && wa && na < need)) { if (act && wa && na <= kNRing - 4) refill_a4(); }
                if (__any(act && ws && nsv < need)) { if (act && ws && nsv <= kNRing - 4) refill_s4(); }
            }
            if (parts & 2) {
                if (__any(act && wj && nj < need)) { if (act && wj && nj <= kNRing - 4) refill_j4(); }
#else
            if (parts & 1) {
                if (__any(act && wa && na < need)) {
                    const bool low = act && wa && na < 2;
                    if (act && wa && na <= kNRing - 2) refill_a2();
                    if (__any(low)) { if (low) refill_a2(); }
                }
                if (__any(act && ws && nsv < need)) {
                    const bool low = act && ws && nsv < 2;
                    if (act && ws && nsv <= kNRing - 2) refill_s2();
                    if (__any(low)) { if (low) refill_s2(); }
                }
            }
            if (parts & 2) {
                if (__any(act && wj && nj < need)) {
                    const bool low = act && wj && nj < 2;
                    if (act && wj && nj <= kNRing - 2) refill_j2();
                    if (__any(low)) { if (low) refill_j2(); }
                }
#endif
                if (__any(act && wr && rn < need)) { if (act && wr && rn <= 8) refill_r(8); }
            }
        }
    }

    __device__ __forceinline__ int64_t next_arrival() {
        if constexpr (PF) {
            if (prof_kind != kProfConstant) {      // general path: tick number `generated` of the Source's table
                arr_time = tick_lookup(tab_a, tab_cap, generated, overflow);
                return arr_time;
            }
        }
        const double inc = HSU(src_kind == 1, true) ? arr_inc() : __ddiv_rn(1.0, rate);
        const double t_next = __dadd_rn(seconds_from_ns(arr_time), inc);
        arr_time = ns_from_seconds(t_next);
        arr_d = (double)arr_time;
        return arr_time;
    }
    __device__ __forceinline__ void sample_service(double &s, int64_t &dur_ns) {
        if (HSU(svc_kind == 0, true)) {
            s = svc_s_next();
            dur_ns = ns_from_seconds(s);
        } else { s = svc_const_s; dur_ns = svc_const_ns; }
    }

    // ---- Probe: Source.handle_event with _ProbeEventProvider, then the measurement callback (hs_station.hpp)
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

    // ---- Simulation.schedule(): the injected Event IS the Request@Server
    __device__ __forceinline__ bool has_sched() const { return PF && SA != kInfNs; }
    __device__ __forceinline__ void root_sched(int64_t t) {
        ++sc_i;
        SA = sc_i < sc_end ? sc_t[sc_i] : kInfNs;
        if (do_enqueue(t, t)) qpush(Q_NOTIFY);
    }

    // ---- handlers (see hs_station.hpp for the reference citations of the shared ones)
    __device__ __forceinline__ uint32_t do_tick(int64_t t) {
        ev[0]++; generated++;
        const bool payload = !(HSU(stop_ns >= 0, false) && t > stop_ns);
        const int64_t a2 = next_arrival();
        uint32_t r = payload ? 1u : 0u;
        if (a2 == t) { r |= 2u; A = kInfNs; }
        else if (a2 < t) { A = kInfNs; }
        else { A = a2; seqA = seq++; crtA = t; dpA = dp_next(1); rcA = cr; }
        return r;
    }
    __device__ __forceinline__ bool do_enqueue(int64_t t, int64_t created) {
        (void)t;
        ev[1]++;
        if (HSU(qcap >= 0, false) && buf >= qcap) { dropped++; return false; }
        const bool was_empty = (buf == 0);
        if (accepted < cap) adm[accepted * lgs] = created; else overflow = 1;
        if constexpr (FAST) window_admit(created);
        accepted++; buf++;
        return was_empty;
    }
    __device__ __forceinline__ bool do_notify() { ev[2]++; return active < conc; }
    __device__ __forceinline__ bool do_poll() {
        ev[3]++;
        if (buf == 0) return false;
        buf--;
        return true;
    }
    // FAST created_at window (see win_hi): at the top of an iteration, fetch what entered [started, started + kNRing) from the log
    __device__ __forceinline__ void window_fill(bool act) {
        if constexpr (FAST) {
            const int64_t lim = started + kNRing;
            const int64_t hi = accepted < lim ? accepted : lim;
            for (int i = 0; i < kNRing; ++i) {
                const int64_t j = win_hi + i;
                const bool need = act && j < hi;
                if (!__any(need)) break;
                if (need) fl.crc[j & (kNRing - 1)][tid] = (j < cap) ? adm[j * lgs] : 0;
            }
            if (act && hi > win_hi) win_hi = hi;
        }
    }
    // window_fill in two halves around the wave-level refill of the stream rings (hs_net_async): the first two entries a lane needs are
    // LOADED here and stored there, so that their memory round trip runs behind the refill's arithmetic instead of in front of it
    // (the LP's own admission log: nothing between the two halves touches it, `accepted`, `started` or `win_hi`); a lane that needs
    // more than two (rare) takes the rest in window_fill as before.
    __device__ __forceinline__ int window_issue(bool act, int64_t &v0, int64_t &v1) const {
        int cnt = 0;
        v0 = 0; v1 = 0;
        if constexpr (FAST) {
            const int64_t lim = started + kNRing;
            const int64_t hi = accepted < lim ? accepted : lim;
            if (act && win_hi < hi) { v0 = (win_hi < cap) ? adm[win_hi * lgs] : 0; cnt = 1; }
            if (act && win_hi + 1 < hi) { v1 = (win_hi + 1 < cap) ? adm[(win_hi + 1) * lgs] : 0; cnt = 2; }
        }
        return cnt;
    }
    __device__ __forceinline__ void window_commit(bool act, int cnt, int64_t v0, int64_t v1) {
        if constexpr (FAST) {
            if (cnt >= 1) fl.crc[win_hi & (kNRing - 1)][tid] = v0;
            if (cnt >= 2) fl.crc[(win_hi + 1) & (kNRing - 1)][tid] = v1;
            win_hi += cnt;
            window_fill(act);
        }
    }
    // created_at of request k, which starts now
    __device__ __forceinline__ int64_t window_take(int64_t k) {
        int64_t created;
        if (k < win_hi) created = fl.crc[k & (kNRing - 1)][tid];
        else created = (k < cap) ? adm[k * lgs] : 0;                   // (more than kNRing starts since the last fill: synchronous)
        if (win_hi <= k) win_hi = k + 1;
        return created;
    }
    // an admission (ordinal `accepted`, before the increment) inside the window
    __device__ __forceinline__ void window_admit(int64_t created) {
        if (accepted - started < kNRing) {
            fl.crc[accepted & (kNRing - 1)][tid] = created;
            if (win_hi == accepted) win_hi = accepted + 1;
        }
    }
    // `known_created`: created_at of the head request when the caller still has it in a register (the request
    // that was enqueued by this very chain into an empty buffer); otherwise it is read back from the log.
    __device__ __forceinline__ uint32_t do_deliver_work(int64_t t, bool have_created, int64_t known_created) {
        ev[4]++; ev[5]++;
        const int64_t k = started++;
        if (active >= conc) { rejected++; if constexpr (FAST) { if (win_hi <= k) win_hi = k + 1; } return 0; }
        active++;
        double s; int64_t dur;
        sample_service(s, dur);
        int64_t created;
        if (have_created) created = known_created;
        else if constexpr (FAST) created = window_take(k);             // no global round trip
        else created = (k < cap) ? adm[k * lgs] : 0;
        if constexpr (FAST) { if (win_hi <= k) win_hi = k + 1; }
        int j = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; --i) if (D[i] == kInfNs) j = i;
        const int64_t d = t + dur;
        uint32_t same = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == j) {
            svc_s[i] = s; crt[i] = created;
            if (d == t) { D[i] = kInfNs - 1; same = (uint32_t)i + 1; }
            else { D[i] = d; seqD[i] = seq++; crtD[i] = t; dpD[i] = dp_next(2); rcD[i] = cr; }   // QUEUE_DELIVER -> payload -> continuation
        }
        if (same) ++cd;                                                   // (the caller pushes the in-group continuation: deliver + 2)
        return same;
    }

    // NetworkLink.handle_event up to its yield (components/network/link.py:114-154, _calculate_delay :190-216)
    // executed for a request that enters link `l` at time t; the continuation becomes a message to the egress LP.
    __device__ __forceinline__ bool link_is_remote(int32_t l) const {
        return sc->wend_slots != nullptr && sc->link_rank[l] != sc->rank;
    }
    // a message for a station of another shard: append to that rank's outbox row (the host exchanges the rows)
    __device__ __forceinline__ void outbox_append(int32_t l, int32_t dst, int64_t t_arr, int64_t t, int64_t created, int64_t lin) {
        int64_t *row = sc->outbox + (size_t)sc->link_rank[l] * sc->row;
        const unsigned long long pos = atomicAdd((unsigned long long *)row, 1ull);
        if (pos < (unsigned long long)sc->msg_cap) {
            int64_t *m = row + 1 + kMsgWords * pos;
            const int64_t gid = np->link_gid ? np->link_gid[l] : l;
            m[0] = t_arr; m[1] = t; m[2] = created; m[3] = ((int64_t)dst << 32) | gid; m[4] = lin;
        } else bagoverflow = 1;
    }
    // LIVE exchange: message number `seq` of a link that leaves the shard goes straight into the link's queue in the destination
    // rank's memory (the caller publishes the link's word behind a drain of these stores, as for a local queue)
    __device__ __forceinline__ int64_t *live_rec(int32_t l, unsigned long long seq) const {
        const int r = sc->link_rank[l];
        const size_t slot = (size_t)sc->peer_link[l] * ns->aq_cap + (size_t)(seq & (unsigned long long)(ns->aq_cap - 1));
        return sc->peer_rec[r] + 4 * slot;
    }
    __device__ __forceinline__ void live_append(int32_t l, unsigned long long seq, int64_t t_arr, int64_t t_send, int64_t created, int64_t lin) {
        sys_store_rec(live_rec(l, seq), t_arr, t_send, created, lin);
        sent_async = true;
    }
    __device__ __forceinline__ void live_publish(int32_t l, int64_t word) const {
        __hip_atomic_store(sc->peer_ea[sc->link_rank[l]] + sc->peer_link[l], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ unsigned long long live_head(int32_t l) const {
        return __hip_atomic_load(sc->peer_head[sc->link_rank[l]] + sc->peer_link[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ int64_t link_sent_of(int32_t l) const {   // the link queue's sequence number
        if constexpr (FAST) { if (l == fl_link) return fl_q; }
        return ns->link_sent[l];
    }
    // append {arrival, send time, created_at} to the LP's one outgoing link (queue or, on a shard, the outbox row)
    __device__ __forceinline__ void fl_append(int64_t t_arr, int64_t t_send, int64_t created, int64_t lin) {
        sent_min = t_arr < sent_min ? t_arr : sent_min;
        ++fl_q;
        if (HSU(fl_remote, false)) {
            if (sc->live) { live_append(fl_link, (unsigned long long)fl_q - 1ull, t_arr, t_send, created, lin); return; }
            outbox_append(fl_link, fl_dst, t_arr, t_send, created, lin);
            return;
        }
        const size_t slot = (size_t)fl_link * ns->aq_cap + (size_t)(((unsigned long long)fl_q - 1) & (unsigned long long)(ns->aq_cap - 1));
#ifndef HS_NO_WG_SCOPE
        if (fl_local) wg_store_rec(&ns->aq_rec[4 * slot], t_arr, t_send, created, lin);
        else
#endif
            ag_store_rec(&ns->aq_rec[4 * slot], t_arr, t_send, created, lin);
        sent_async = true;
    }
    // steps from a Server's continuation to the NetworkLink's continuation it causes: Request@Link, the link's continuation
    // (+ Request@RandomRouter in front of them)
    __device__ __forceinline__ int link_steps() const { return HSU(egress == EG_ROUTER, true) ? 3 : 2; }
    // Decide the way out of request `ordinal` (departure D, created_at `created`) ahead of time; `bit_off` = its distance, in
    // completions, from the next one.  False when the pre-drawn route / jitter values do not reach that far.
    __device__ __forceinline__ bool pre_send(int64_t ordinal, int bit_off, int64_t D, int64_t created, int64_t S_start) {
        const bool router = HSU(egress == EG_ROUTER, true);
        if (router && bit_off >= rn) return false;
        const int32_t target = HSU(egress == EG_SINK, false) ? -1 : HSU(egress == EG_LINK, false) ? link_of
                             : router ? rt_target((int)((rbits >> (2 * bit_off)) & 3u)) : -2;
        // (a message that leaves beyond end_ns is NOT decided ahead of time: what this launch leaves behind -- early_upto, the bounds that
        //  skip pre-sent requests -- must not depend on where it stops, so that the next run_until continues from it; round 6)
        if (target >= 0 && D > end_ns) return false;
        if (target >= 0) {
            if (HSU(fl_jit == 0, true) && nj == 0) return false;
            double delay = fl_delay0;
            if (HSU(fl_jit == 0, true)) { delay = __dadd_rn(delay, fl.ring_j[hj][tid]); hj = (hj + 1) & (kNRing - 1); --nj; }
            if (!(delay > 0.0)) delay = 0.0;
            fl_append(D + ns_i(delay), D, created, lin_pack(link_steps(), S_start, D));   // (the continuation at D: a root, created at S_start)
        }
        early_upto = ordinal + 1;
        D_pre = D;
        return true;
    }
    // the LP's only outgoing link with its state in registers and the jitter E pre-drawn (FAST; queue path only)
    __device__ __forceinline__ void send_link_fast(int64_t t, int64_t created) {
        ev[8]++;
        if (HSU(presend, true) && completed - 1 < early_upto) { fl_in++; fl_sent++; return; }   // its message was appended ahead of time
        const int64_t entered = fl_in++;
        if (HSU(fl_loss > 0.0, false)) { if (link_loses(*np, seed, fl_link, entered, t)) return; }
        ++fl_sent;
        double delay = fl_delay0;
        if (HSU(fl_jit == 0, true)) {
            if (nj == 0) refill_j(2);
            delay = __dadd_rn(delay, fl.ring_j[hj][tid]);
            hj = (hj + 1) & (kNRing - 1); --nj;
        }
        if (!(delay > 0.0)) delay = 0.0;
        fl_append(t + ns_i(delay), t, created, lin_pack(dp_next(link_steps()), cr, t));
    }
    __device__ __forceinline__ void send_link(int32_t l, int64_t t, int64_t created) {
        if constexpr (FAST) { if (l == fl_link) { send_link_fast(t, created); return; } }
        ev[8]++;
        const int64_t entered = ns->link_in[l]++;
        const double loss = np->link_loss[l];
        if (loss > 0.0) {
            // packet loss is decided before anything else (link.py:131-138: `random.random() < packet_loss_rate`, here u
            // of the link's LOSS stream, one draw per packet that enters); the generator ends without yielding
            if (link_loses(*np, seed, l, entered, t)) return;
        }
        ns->link_sent[l]++;
        double delay = seconds_from_ns(ns_from_seconds(np->link_lat_min[l]));          // ConstantLatency
        if (np->link_jit_kind[l] == 0) {
            Stream js;
            js.init(seed, stream_id(np->link_base[l], kStreamLink), ns->link_k[l]);
            ns->link_k[l]++;
            const double lam = __ddiv_rn(1.0, np->link_jit_mean[l]);
            const double sample = __ddiv_rn(exp1_from_uniform(js.next_uniform()), lam);
            delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(sample)));        // + jitter
        } else delay = __dadd_rn(delay, seconds_from_ns(ns_from_seconds(np->link_jit_mean[l])));   // ConstantLatency jitter (0: none)
        if (!(delay > 0.0)) delay = 0.0;                                               // max(0.0, delay)
        const int64_t t_arr = t + ns_from_seconds(delay);
        sent_min = t_arr < sent_min ? t_arr : sent_min;
        const int32_t dst = np->link_dst[l];                 // network-wide station index
        const int64_t lin = lin_pack(dp_next(link_steps()), cr, t);
        if (link_is_remote(l)) {                             // destination lives on another engine
            if (sc->live) live_append(l, (unsigned long long)ns->link_sent[l] - 1ull, t_arr, t, created, lin);
            else outbox_append(l, dst, t_arr, t, created, lin);
            return;
        }
        if (ns->aq_on) {
            // asynchronous engine: append to the link's queue (this LP is its only producer); link_sent[l] is the
            // sequence number of this message
            // (room for this group's messages was checked before the group started: async_can_send)
            const unsigned long long seq = (unsigned long long)ns->link_sent[l];
            const size_t slot = (size_t)l * ns->aq_cap + (size_t)((seq - 1) & (unsigned long long)(ns->aq_cap - 1));
            ag_store_rec(&ns->aq_rec[4 * slot], t_arr, t, created, lin);
            sent_async = true;          // the caller publishes aq_tail (= link_sent) after draining these stores
            return;
        }
        const int32_t dl = dst - (int32_t)sc->lp_base;       // local index on this engine
        const size_t cslot = (size_t)send_idx * n + dl;
        const int pos = atomicAdd(&ns->in_cnt[cslot], 1);
        if (pos < ns->bag_cap) {
            const size_t b = cslot * ns->bag_cap + pos;
            ns->in_t[b] = t_arr; ns->in_ts[b] = t; ns->in_cr[b] = created; ns->in_link[b] = l; ns->in_lin[b] = lin;
        } else bagoverflow = 1;
    }
    __device__ __forceinline__ int64_t gid_of(int64_t l) const { return np->link_gid ? np->link_gid[l] : l; }

    // generator resumes (server/server.py:252-273): statistics only
    __device__ __forceinline__ int64_t do_cont_core(int slot, int64_t t) {
        (void)t;
        ev[6]++;
        double s = 0.0; int64_t created = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) if (i == slot) { s = svc_s[i]; created = crt[i]; D[i] = kInfNs; }
        active = active > 0 ? active - 1 : 0;
        completed++;
        total_service = __dadd_rn(total_service, s);
        return created;
    }
    // the forwarded request's way out of the LP: Sink / RandomRouter / NetworkLink (all at time t)
    __device__ __forceinline__ void do_egress(int64_t t, int64_t created) {
        const bool pre = FAST && C == 1 && HSU(presend, true) && completed - 1 < early_upto;   // way out decided ahead of time
        do_egress_inner(t, created);
        if (FAST && C == 1 && !pre) { if (early_upto < completed) { early_upto = completed; D_pre = t; } }
    }
    __device__ __forceinline__ void do_egress_inner(int64_t t, int64_t created) {
        int32_t target = -2;             // -2 nothing, -1 sink, >= 0 link
        if (HSU(egress == EG_SINK, false)) target = -1;
        else if (HSU(egress == EG_LINK, false)) target = link_of;
        else if (HSU(egress == EG_ROUTER, true)) {  // RandomRouter.handle_event (components/random_router.py:32-45)
            ev[10]++; routed++;
            const int idx = route_idx();
            target = rt_target(idx);
        }
        if (target == -1) {              // Sink.handle_event (components/common.py:36-44)
            ev[7]++;
            if (received < cap) { sink_t[received * lgs] = t; sink_created[received * lgs] = created; } else overflow = 1;
            received++;
        } else if (target >= 0) send_link(target, t, created);
    }
    // full continuation: statistics, egress chain, schedule_poll hook.  Returns true if QUEUE_POLL is created.
    __device__ __forceinline__ bool do_cont(int slot, int64_t t) {
        const int64_t created = do_cont_core(slot, t);
        do_egress(t, created);
        return active < conc;
    }
    // NetworkLink continuation at the egress side (link.py:156-189): transit over, a new Request for the Server
    __device__ __forceinline__ int64_t do_msg(int i, int64_t t) {
        (void)t;
        ev[9]++;
        const int64_t created = bg_cr(i);
        if (FAST && bg_link(i) == fi_link) fi_packets++;
        else ns->link_packets[bg_link(i)]++;
        bag_remove(i);
        return created;
    }

    // (cd = the QUEUE_POLL's steps from the group's root)
    __device__ __forceinline__ bool chain_from_poll(int64_t t, bool have_created, int64_t created) {
        if (!do_poll()) return false;
        ++cd;                                                             // the QUEUE_DELIVER it created
        const uint32_t same = do_deliver_work(t, have_created, created);
        if (same) { qpush(Q_CONT | ((same - 1) << 3)); return true; }
        return false;
    }

    // ---- asynchronous engine -----------------------------------------------------------------
    // take delivery of everything the incoming links hold; returns min over those links of aq_ea (kInfNs: no in-links)
    // one incoming link (FAST): its id and this LP's queue position are in registers, so the only global round trip is
    // the (bound, tail) pair.  (Loading that pair ahead of time, before this wavefront's publication drains, is NOT safe:
    // the in-wavefront scan hands a lane its neighbour's bound from the neighbour's CURRENT state, and the messages that
    // state has already sent are only guaranteed to be behind a tail read after the neighbour's drains.)
    // (It may be loaded at the very top of the iteration, before the stream rings are topped up -- that is after every
    // publication of the previous iteration: async_peek.)
    // This LP has taken the messages below `head` of its one incoming link (FAST: the position lives in a register).  The
    // producer reads the published position only when its queue LOOKS full (async_can_send), so the word is written when that
    // can be the case -- the queue, measured from the last published position, is at least half full -- and when the launch
    // ends (store_net); a producer that blocks on a stale position therefore gets the exact one at this LP's next receive.
    // (Round 6: one write-through store per receive -- ~10 M partial-line writes on the headline ring -- was a tenth of the
    // kernel's HBM traffic for a word that was read a few hundred times.)
    __device__ __forceinline__ void head_taken(int l, unsigned long long head, unsigned long long tail) {
        fi_unpub += (int32_t)(head - fi_head);
        fi_head = head;
        if ((long long)(tail - head) + fi_unpub >= (long long)(ns->aq_cap >> 1)) { ag_store(&ns->aq_head[l], head); fi_unpub = 0; }
    }
#ifndef HS_NO_WG_SCOPE
    __device__ __forceinline__ int64_t rec_load(const int64_t *p) const { return fi_local ? wg_load(p) : ag_load(p); }
#else
    __device__ __forceinline__ int64_t rec_load(const int64_t *p) const { return ag_load(p); }
#endif
    __device__ __forceinline__ int64_t async_peek() const {
        if constexpr (FAST) { if (fi_link >= 0) return ag_load(&ns->aq_ea[fi_link]); }
        return 0;
    }
    __device__ __forceinline__ int64_t async_receive_one(int64_t w) {
        const int l = fi_link;                                        // w: (bound, tail) in one word: a consistent pair; the payload
        const int64_t ea = pk_ea(w, ns->pk_base);                     // loads below depend on it (loop bounds), so they follow it
        undrained = kInfNs;
        unsigned long long head = fi_head;
        unsigned long long tail = pk_tail(w, head);
        // a sender in the same wavefront hands over its tail through a shuffle (the scan's bound is computed from its
        // CURRENT state: everything that state has appended must be taken, whether or not the word has landed yet; the
        // payloads are complete -- the sender drained them before the iteration ended)
        if (tail_hint > tail) tail = tail_hint;
        if (head != tail) {
            const int bcap = bag_capacity();
            for (; head < tail && bag_n < bcap; ++head) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                const int64_t ta = rec_load(&ns->aq_rec[4 * slot]);
                bag_insert(ta, rec_load(&ns->aq_rec[4 * slot + 1]), rec_load(&ns->aq_rec[4 * slot + 2]), l, rec_load(&ns->aq_rec[4 * slot + 3]));
            }
            head_taken(l, head, tail);
            if (head < tail) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                undrained = rec_load(&ns->aq_rec[4 * slot + 1]) + np->link_lat_ns[l];
            }
        }
        return ea;
    }
    // async_receive_one in two halves (one incoming link, its word `w` has arrived): the first two messages of [head, tail) that fit
    // into the bag are LOADED here and inserted there; the second half of the stream refill runs between the two.
    __device__ __forceinline__ int async_receive_issue(int64_t w, int64_t (&r)[8], unsigned long long &tail_out) const {
        const int l = fi_link;
        const unsigned long long head = fi_head;
        unsigned long long tail = pk_tail(w, head);
        if (tail_hint > tail) tail = tail_hint;
        tail_out = tail;
        const int room = bag_capacity() - bag_n;
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            r[4 * m] = r[4 * m + 1] = r[4 * m + 2] = r[4 * m + 3] = 0;
            if (head + (unsigned long long)m < tail && m < room) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)((head + (unsigned long long)m) & (unsigned long long)(ns->aq_cap - 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) r[4 * m + q] = rec_load(&ns->aq_rec[4 * slot + q]);
                cnt = m + 1;
            }
        }
        return cnt;
    }
    __device__ __forceinline__ int64_t async_receive_commit(int64_t w, int cnt, const int64_t (&r)[8], unsigned long long tail) {
        const int l = fi_link;
        const int64_t ea = pk_ea(w, ns->pk_base);
        undrained = kInfNs;
        unsigned long long head = fi_head;
        if (head != tail) {
            if (cnt >= 1) { bag_insert(r[0], r[1], r[2], l, r[3]); ++head; }
            if (cnt >= 2) { bag_insert(r[4], r[5], r[6], l, r[7]); ++head; }
            const int bcap = bag_capacity();
            for (; head < tail && bag_n < bcap; ++head) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                const int64_t ta = rec_load(&ns->aq_rec[4 * slot]);
                bag_insert(ta, rec_load(&ns->aq_rec[4 * slot + 1]), rec_load(&ns->aq_rec[4 * slot + 2]), l, rec_load(&ns->aq_rec[4 * slot + 3]));
            }
            head_taken(l, head, tail);
            if (head < tail) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                undrained = rec_load(&ns->aq_rec[4 * slot + 1]) + np->link_lat_ns[l];
            }
        }
        return ea;
    }
    __device__ __forceinline__ int64_t async_receive(int64_t w_peek) {
        if constexpr (UNI) return async_receive_one(w_peek);          // (every station has exactly one incoming link)
        if constexpr (FAST) { if (fi_link >= 0) return async_receive_one(w_peek); }
        int64_t H = kInfNs;
        undrained = kInfNs;
        const int a = np->in_off[lp], b = np->in_off[lp + 1];
        for (int q = a; q < b; ++q) {
            const int l = np->in_links[q];
            const int64_t w = ag_load(&ns->aq_ea[l]);                  // (bound, tail) in one word
            const int64_t ea = pk_ea(w, ns->pk_base);
            H = ea < H ? ea : H;
            unsigned long long head = ns->aq_head[l];                  // ours
            const unsigned long long tail = pk_tail(w, head);
            if (head == tail) continue;
            const int bcap = bag_capacity();
            for (; head < tail && bag_n < bcap; ++head) {
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                const int64_t ta = ag_load(&ns->aq_rec[4 * slot]);
                bag_insert(ta, ag_load(&ns->aq_rec[4 * slot + 1]), ag_load(&ns->aq_rec[4 * slot + 2]), l, ag_load(&ns->aq_rec[4 * slot + 3]));
            }
            ag_store(&ns->aq_head[l], head);
            if (head < tail) {
                // the bag is full: what stays in the queue was sent no earlier than its first entry (send times do not
                // decrease along a queue), so it arrives no earlier than that + the link's transit floor
                const size_t slot = (size_t)l * ns->aq_cap + (size_t)(head & (unsigned long long)(ns->aq_cap - 1));
                const int64_t lb = ag_load(&ns->aq_rec[4 * slot + 1]) + np->link_lat_ns[l];
                undrained = lb < undrained ? lb : undrained;
            }
        }
        return H;                         // the caller caps it with `undrained`
    }
    // back-pressure: an LP processes events only while each of its outgoing queues can take what one timestamp group
    // may send (one message per completion, at most C completions per group; a pre-sending station up to two per group)
    // `head_seen` caches the consumer's published position: it only moves forward, so a stale value errs on the safe
    // side and the (cache-bypassing) reload is needed only when the queue looks full
    __device__ __forceinline__ bool async_can_send(int32_t l, unsigned long long &head_seen, bool remote = false) const {
        if (l < 0) return true;
        const unsigned long long sent = (unsigned long long)link_sent_of(l);
        if (sent - head_seen + 2ull * C <= (unsigned long long)ns->aq_cap) return true;
        head_seen = remote ? live_head(l) : ag_load(&ns->aq_head[l]);      // (LIVE exchange: the consumer is on another rank)
        return sent - head_seen + 2ull * C <= (unsigned long long)ns->aq_cap;
    }
    // Shortest duration among the next `free` services to start: service draws svc.k .. svc.k + free - 1, not consumed
    // (pure functions of the draw index).  With `free` idle workers that many requests can be in service before any
    // completion, so the first completion of a not-yet-started request is no earlier than its start + this.
    __device__ __forceinline__ int64_t peek_service_ns(int free) const {
        if (HSU(svc_kind != 0, false)) return svc_const_ns;
        Stream c = svc;                   // FAST: positioned behind the pre-drawn values, which come first
        int64_t m = kInfNs;
        for (int i = 0; i < free; ++i) {
            double sv;
            if (FAST && i < nsv) sv = fl.ring_s[(hs_ + i) & (kNRing - 1)][tid];
            else sv = svc_value(exp1_from_uniform(c.next_uniform()));
            const int64_t d = ns_i(sv);
            m = d < m ? d : m;
        }
        return m;
    }

    // ---- look-ahead over the pre-drawn decisions (C == 1, FAST) ----------------------------------------------------
    // A sender's bound on link l is the earliest time it can still SEND on l.  Completions leave in start order (one worker),
    // the router's choice for the k-th completion from now is route draw number routed + k -- already drawn (`rbits`) -- and the
    // service times of the next requests to start are pre-drawn too (`ring_s`).  So the bound need not stop at the next
    // completion: it is the lower bound of the first completion whose route draw says `l`,
    //     busy:  D + s_1 + ... + s_q          idle:  (next possible start) + s_1 + ... + s_(q+1)
    // (q = completions before it that go elsewhere; s_i = service time of the i-th next request to start: the next service
    // cannot start before the previous one ends, whatever arrives).  Truncating the sums is conservative.
    __device__ __forceinline__ int skip_to_link(int32_t l) const {
        if (egress != EG_ROUTER) return 0;                            // EG_LINK: every completion enters the link
        int q = 0;
        uint32_t b = rbits;
        while (q < rn && rt_target((int)(b & 3u)) != l) { ++q; b >>= 2; }
        return q;                                                     // == rn: none of the known decisions goes to l
    }
    __device__ __forceinline__ int64_t sum_services(int m) const {   // of the next m requests to start, m <= services_known()
        if (HSU(svc_kind != 0, false)) return (int64_t)m * svc_const_ns;
        int64_t sum = 0;
        for (int i = 0; i < m; ++i) sum += ns_i(fl.ring_s[(hs_ + i) & (kNRing - 1)][tid]);
        return sum;
    }
    __device__ __forceinline__ int services_known() const { return HSU(svc_kind != 0, false) ? kNRing : nsv; }
    // the next admission the LP already knows about: its own Source's tick, a message in the bag, an injected Request
    __device__ __forceinline__ int64_t next_admission() const {
        int64_t a = A < bmin ? A : bmin;
        if (has_sched() && SA < a) a = SA;
        if (has_xsrc()) { const int64_t xm = xsrc_min(); if (xm < a) a = xm; }
        return a;
    }

    // The map  ea(H) = min(mA, max(H + mB, mD))  of this LP as the sender on link l (transit floor `lat`), from its CURRENT
    // state: a lower bound on the arrival of every message it has not appended to l yet, given that nothing reaches it from
    // upstream before H.  (min D / next own event / the next free workers' service times: every quantity is already determined.)
    // `kind` / `sdl` (optional) say how mA depends on what the next receive can change, so that the caller can keep the map
    // across the iteration boundary instead of evaluating it twice:  0: not at all;  1: mA = max(next known admission, D_pre)
    // + sdl;  2: mA = next_time() + sdl;  -1: re-evaluate.
    __device__ __forceinline__ void bound_map(int32_t l, int64_t lat, int64_t &mA, int64_t &mB, int64_t &mD, int *kind = nullptr,
                                              int64_t *sdl = nullptr) const {
        auto sat = [](int64_t a, int64_t b) { return (a == kInfNs || b == kInfNs) ? kInfNs : a + b; };
        mA = kInfNs; mB = kInfNs; mD = INT64_MIN;
        if (kind) *kind = -1;
        if constexpr (C == 1 && FAST) {
            const int known = services_known();
            if (HSU(presend, true) && l == fl_link) {
                // Requests with ordinal < early_upto are pre-sent; the first message still to come belongs to request u =
                // early_upto or a later one -- whichever is the first whose (pre-drawn) route decision is l:
                //   u in service          leaves at D[0]; the requests behind it start when it leaves;
                //   u waiting             starts exactly when request u - 1 (pre-sent, departure D_pre) leaves;
                //   u not admitted yet    is admitted at a >= min(next own arrival, H), starts at >= max(a, D_pre);
                // each later candidate adds the service time of the request in front of it.  Truncated sums are conservative.
                const int64_t u = early_upto;
                const int boff = (int)(u - completed);               // its distance from the next completion
                int q = 0;                                           // requests from u on that go elsewhere first
                if (HSU(egress == EG_ROUTER, true)) {
                    uint32_t b = boff < 16 ? rbits >> (2 * boff) : 0u;
                    const int have = rn - boff;
                    while (q < have && q < kLookMax && rt_target((int)(b & 3u)) != l) { ++q; b >>= 2; }
                }
                if (u < started) {                                   // in service
                    const int e = q < known ? q : known;
                    mA = sat(sat(D[0], sum_services(e)), lat);
                    if (kind) *kind = 0;
                    return;
                }
                const int off = (int)(u - started);
                int64_t sd = 0;
                if (HSU(svc_kind != 0, false)) sd = (int64_t)(q + 1) * svc_const_ns;
                else for (int i = 0; i <= q && off + i < known; ++i) sd += ns_i(fl.ring_s[(hs_ + off + i) & (kNRing - 1)][tid]);
                if (u < accepted) { mA = sat(sat(D_pre, sd), lat); if (kind) *kind = 0; return; }
                const int64_t arr_next = next_admission();
                const int64_t own = arr_next > D_pre ? arr_next : D_pre;
                mA = sat(sat(own, sd), lat);
                mB = sd + lat;
                mD = sat(sat(D_pre, sd), lat);
                if (kind) { *kind = 1; *sdl = sd + lat; }
                return;
            }
            if (kLookMax > 1) {
                // one worker: look ahead to the first completion that is routed to l (see skip_to_link)
                int m = skip_to_link(l) + 1;
                m = m < kLookMax ? m : kLookMax;
                if (active >= conc) {
                    const int e = (m - 1) < known ? (m - 1) : known;
                    mA = sat(sat(D[0], sum_services(e)), lat);
                    if (kind) *kind = 0;
                } else {
                    const int e = m < known ? m : known;
                    const int64_t sd = sum_services(e);
                    mA = sat(sat(next_time(), sd), lat);
                    mB = sd + lat;
                    if (kind) { *kind = 2; *sdl = sd + lat; }
                }
                return;
            }
        }
        int64_t dmin = kInfNs;
#pragma unroll
        for (int i = 0; i < C; ++i) dmin = D[i] < dmin ? D[i] : dmin;
        int64_t a = dmin;
        if (active < conc) {
            const int64_t dur = peek_service_ns(conc - active);
            const int64_t own = sat(next_time(), dur);
            a = own < a ? own : a;
            mB = dur + lat;
        }
        mA = sat(a, lat);
    }

    // ---- further Sources: Source.handle_event (load/source.py:142-180) of an entity of its own (hs_station.hpp root_xsrc)
    __device__ __forceinline__ bool has_xsrc() const { return PF && n_xsrc > 0; }
    __device__ __forceinline__ int64_t xsrc_min() const { return xs_min; }
    __device__ __forceinline__ bool xsrc_at(int64_t t) const { return xs_min == t; }
    __device__ __forceinline__ size_t xo(int j) const { return (size_t)j * (size_t)x_n_lp + (size_t)lp; }
    __device__ __forceinline__ void root_xsrc(int j, int64_t t) {
        ev[0]++;
        const size_t o = xo(j);
        xx->x_n[o] += 1;
        const int64_t stop = xp->xsrc_stop[o];
        const bool payload = !(stop >= 0 && t > stop);
        double area = 1.0;
        if (xp->xsrc_kind[o] == 1) {
            Stream st;
            st.init(seed, xsrc_stream_id(x_base, j), xx->x_k[o]);
            area = exp1_from_uniform(st.next_uniform());
            xx->x_k[o] += 1;
        }
        const int64_t a2 = ns_from_seconds(__dadd_rn(seconds_from_ns(xx->x_arr[o]), __ddiv_rn(area, xp->xsrc_rate[o])));
        xx->x_arr[o] = a2;
        if (payload) push_enq(t);
        if (a2 == t) { xx->XA[o] = kInfNs; qpush(Q_TICK | ((uint32_t)(j + 1) << 3)); }
        else if (a2 < t) xx->XA[o] = kInfNs;
        else { xx->XA[o] = a2; xx->seqX[o] = seq++; xx->crtX[o] = t; xx->dpX[o] = (uint8_t)dp_next(1); xx->rcX[o] = cr; }
        int64_t m = kInfNs;
        for (int i = 0; i < n_xsrc; ++i) { const int64_t a = xx->XA[xo(i)]; m = a < m ? a : m; }
        xs_min = m;
    }

    // ---- general path ----------------------------------------------------------------------
    __device__ __forceinline__ void root_tick(int64_t t) {
        const uint32_t r = do_tick(t);
        if (r & 1u) push_enq(t);
        if (r & 2u) qpush(Q_TICK);
    }
    __device__ __forceinline__ void root_cont(int slot, int64_t t) { if (do_cont(slot, t)) qpush(Q_POLL); }
    __device__ __forceinline__ void root_msg(int i, int64_t t) { push_enq(do_msg(i, t)); }

    // pending root at time t with the earliest creation: 0 none, 1 tick, 2+slot departure, 64+i message
    __device__ __forceinline__ int pick_root(int64_t t) const {
        int best = 0; int64_t bc = 0; uint32_t bs = 0; bool bmsg = false; int64_t bl = 0;
        if (has_sched() && SA == t) {
            if (sc_idx == nullptr) return 62;
            best = 62; bc = INT64_MIN; bs = sc_idx[sc_i];           // constructed before run(): its true sort index
        }
        if (A == t && (best == 0 || (int32_t)(seqA - bs) < 0)) { best = 1; bc = crtA; bs = seqA; }
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (D[i] == t && (best == 0 || (int32_t)(seqD[i] - bs) < 0)) { best = 2 + i; bc = crtD[i]; bs = seqD[i]; }
        if (PF && n_xsrc > 0 && xs_min == t)     // further Source j's pending tick: root code 48 + j
            for (int j = 0; j < n_xsrc; ++j) {
                const size_t o = xo(j);
                const uint32_t sq = xx->seqX[o];
                if (xx->XA[o] == t && (best == 0 || (int32_t)(sq - bs) < 0)) { best = 48 + j; bc = xx->crtX[o]; bs = sq; }
            }
#pragma unroll
        for (int j = 0; j < kMaxProbes; ++j)     // probe j's pending tick: root code 56 + j
            if (PF && j < n_probes && PA[j] == t && (best == 0 || (int32_t)(seqP[j] - bs) < 0)) { best = 56 + j; bc = crtP[j]; bs = seqP[j]; }
        for (int i = 0; bmin == t && i < bag_n; ++i) {
            if (bg_t(i) != t) continue;
            const int64_t ts = bg_ts(i);
            const int64_t ln = gid_of(bg_link(i));
            bool better;
            if (best == 0) better = true;
            else if (!bmsg) better = ts < bc;                       // local event first on equal creation time
            else better = (ts < bc) || (ts == bc && ln < bl);
            if (better) { best = 64 + i; bc = ts; bmsg = true; bl = ln; }
        }
        return best;
    }
    // creation time of pending root `w` (pick_root's code); a message's is its send time
    __device__ __forceinline__ int64_t root_crt(int w) const {
        int64_t c = INT64_MIN;                                            // 62: constructed before run()
        if (w == 1) c = crtA;
        else if (w >= 64) c = bg_ts(w - 64);
        else if (PF && w >= 48 && w < 48 + kMaxXSrc) c = xx->crtX[xo(w - 48)];
        else if (PF && w >= 56 && w < 56 + kMaxProbes) {
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j == w - 56) c = crtP[j];
        } else if (!(PF && w == 62)) {
#pragma unroll
            for (int i = 0; i < C; ++i) if (i == w - 2) c = crtD[i];
        }
        return c;
    }
    __device__ __forceinline__ void run_root(int w, int64_t t) {
        cd = 0; cr = root_crt(w);                                         // a root: pending from an earlier nanosecond
        if (w == 1) root_tick(t);
        else if (w >= 64) root_msg(w - 64, t);
        else if (PF && w >= 48 && w < 48 + kMaxXSrc) { if constexpr (PF) root_xsrc(w - 48, t); }
        else if (PF && w >= 56 && w < 56 + kMaxProbes) { if constexpr (PF) root_probe(w - 56, t); }
        else if (PF && w == 62) { if constexpr (PF) root_sched(t); }
        else root_cont(w - 2, t);
    }
    // ---- windows (core/simulation.py:527-541 `_run_window`: `_execute_until` again) ------------------------------------------
    // The one event beyond end_time is the FIRST micro-event of its timestamp group.  The election (hs_net_window) runs it with
    // root_first(), which leaves what that event created in the in-group FIFO exactly as run_root() would; pending_pack() /
    // resume_pending() carry the FIFO over to the next run_until, where the group finishes before anything else happens (it is
    // the network's earliest pending work).  Returns true for a departure: its continuation has run (statistics), the forwarded
    // request's way out (Sink / RandomRouter / NetworkLink -- reference events of their own) has not.
    __device__ __forceinline__ bool root_first(int w, int64_t t) {
        pre_run_hazard(t);
        cd = 0; cr = root_crt(w);
        if (w == 1) root_tick(t);                                         // the SourceEvent; its Request / a same-ns next tick wait in the FIFO
        else if (w >= 64) root_msg(w - 64, t);                            // the link's continuation; its Request@Server waits
        else if (PF && w >= 48 && w < 48 + kMaxXSrc) { if constexpr (PF) root_xsrc(w - 48, t); }
        else if (PF && w >= 56 && w < 56 + kMaxProbes) { if constexpr (PF) root_probe(w - 56, t); }   // its probe_event waits
        else if (PF && w == 62) { if constexpr (PF) root_sched(t); }      // the injected Request@Server; its QUEUE_NOTIFY waits
        else { (void)do_cont_core(w - 2, t); return true; }
        return false;
    }
    // the FIFO behind root_first() as one word: codes 0..2 (8 bits each) | count << 24 | way-out-pending << 26 | worker slot << 27
    __device__ __forceinline__ uint32_t pending_pack(bool egress, int slot, bool &fits) const {
        fits = qn <= 3 && pn <= 1 && qh == 0 && ph == 0 && slot < 32;
        uint32_t p = (uint32_t)(qn & 3) << 24;
        for (int k = 0; k < qn && k < 3; ++k) p |= (uint32_t)qmem[k][tid] << (8 * k);
        if (egress) p |= (1u << 26) | ((uint32_t)slot << 27);
        return p;
    }
    __device__ __forceinline__ int64_t pending_payload() const { return pn > 0 ? enq[(size_t)ph * enq_stride] : 0; }
    // ... and the rest of that group, on a freshly loaded LP (the FIFO's lineage columns qdep / qrc are global memory: slots
    // 0 .. count - 1 still hold what the election's qpush wrote)
    __device__ __forceinline__ void resume_pending(uint32_t p, int64_t t, int64_t payload) {
        const int cnt = (int)((p >> 24) & 3u);
        qh = 0; qn = 0; ph = 0; pn = 0;
        for (int k = 0; k < cnt; ++k) {
            const uint32_t code = (p >> (8 * k)) & 0xffu;
            qmem[k][tid] = (uint8_t)code;
            if ((code & 7u) == Q_ENQ && pn == 0) { enq[0] = payload; pn = 1; }
        }
        qn = cnt;
        if (p & (1u << 26)) {                                             // root_cont() behind its do_cont_core()
            const int slot = (int)(p >> 27);
            int64_t created = 0;
            cd = 0;
#pragma unroll
            for (int i = 0; i < C; ++i) if (i == slot) { cr = crtD[i]; created = crt[i]; }
            do_egress(t, created);
            if (active < conc) qpush(Q_POLL);
        }
        run_group_general(t);             // the roots of this nanosecond that are still pending, then the FIFO
        last_time = t;
    }
    __device__ __forceinline__ void drain(int64_t t) {
        while (qn > 0) {
            const uint32_t code = qpop();
            switch (code & 7u) {
                case Q_ENQ: if (do_enqueue(t, pop_enq_payload())) qpush(Q_NOTIFY); break;
                case Q_NOTIFY: if (do_notify()) qpush(Q_POLL); break;
                case Q_POLL: if (do_poll()) qpush(Q_DELIVER); break;
                case Q_DELIVER: { const uint32_t sm = do_deliver_work(t, false, 0); if (sm) qpush(Q_CONT | ((sm - 1) << 3)); } break;
                case Q_TICK:
                    if (PF && (code >> 3) != 0) { if constexpr (PF) root_xsrc((int)(code >> 3) - 1, t); }
                    else root_tick(t);
                    break;
                case Q_CONT: root_cont((int)(code >> 3), t); break;
                case Q_PSAMPLE: if constexpr (PF) do_probe_sample((int)(code >> 3), t); break;
                default: break;
            }
        }
    }
    __device__ __forceinline__ void run_group_general(int64_t t) {
        for (;;) { const int w = pick_root(t); if (w == 0) break; run_root(w, t); }
        drain(t);
    }

    __device__ __forceinline__ int64_t next_local() const {
        int64_t t = A;
#pragma unroll
        for (int i = 0; i < C; ++i) t = D[i] < t ? D[i] : t;
        if (has_probe()) { const int64_t pm = probe_min(); if (pm < t) t = pm; }
        if (has_sched() && SA < t) t = SA;
        if (has_xsrc()) { const int64_t xm = xsrc_min(); if (xm < t) t = xm; }
        return t;
    }
    __device__ __forceinline__ int64_t next_time() const { const int64_t a = next_local(), b = bag_min(); return a < b ? a : b; }

    // ---- C == 1, FAST: one timestamp group as straight-line predicated code ----------------------------------------
    // The asynchronous engine's group loop is divergent by nature (every lane is at a different event); run_group()'s three
    // root kinds (tick / message / departure) then execute one after the other, each for a handful of lanes.  Here the
    // common case -- exactly one pending root at `t`, nothing that creates a same-timestamp successor -- is ONE instruction
    // stream with selects, the same idea as Station::step_c1: the reference events of the group are decided first from
    // speculative peeks at the pre-drawn values, anything unusual is detected before a single word of state changes and is
    // handed to run_group() (ties, a next tick on / before `t`, a zero-length service, an empty ring, a lossy or second
    // link).  Event counts, statistics, creation stamps and draw consumption are exactly run_group()'s.
    // `act`: this lane takes part (the caller's loop is UNIFORM: every lane of the wavefront walks through the same trips, a
    // lane without a ready group is predicated off -- no divergent loop exits, whose exec-mask bookkeeping and loop-carried
    // copies cost more than the arithmetic).
    __device__ __forceinline__ void step1(int64_t t, bool act, bool force_general) {
        static_assert(C == 1, "step1 is the single-worker specialisation");
#ifdef HS_CYC2
        unsigned long long cy_t = __builtin_readcyclecounter();
#endif
        bool tick = act && (A == t), dep = act && (D[0] == t);
        int cnt = (tick ? 1 : 0) + (dep ? 1 : 0), mi = 0;
        if (act && bmin == t) { ++cnt; if (bag_n > 1 && bg_t(1) == t) ++cnt; }   // the sorted bag: entry 0 is the message (mi = 0)
        bool msg = act && !tick && !dep;                              // (cnt == 1 is checked below)
        // speculative draws: peeks, nothing consumed yet
        const bool poisson = HSU(src_kind == 1, true), svc_exp = HSU(svc_kind == 0, true);
        const double inc = poisson ? fl.ring_a[ha][tid] : inc_const;
        const int64_t a2 = ns_i(__dadd_rn(UNI ? seconds_from_ns_d(arr_d) : seconds_from_ns(arr_time), inc));
        const double s_new = svc_exp ? fl.ring_s[hs_][tid] : svc_const_s;
        const int64_t dur = svc_exp ? ns_i(s_new) : svc_const_ns;
        const bool router = HSU(egress == EG_ROUTER, true);
        const int ridx = (int)(rbits & 3u);
        const int32_t target = HSU(egress == EG_SINK, false) ? -1 : HSU(egress == EG_LINK, false) ? link_of : router ? rt_target(ridx) : -2;
        bool to_sink, to_link, payload, arrv, acc, notify, poll, deliver, pre_done;
        int64_t buf1;
        auto derive = [&]() {                                         // which reference events happen
            to_sink = dep && target == -1; to_link = dep && target >= 0;
            payload = tick && !(HSU(stop_ns >= 0, false) && t > stop_ns);
            arrv = payload || msg;
            acc = arrv && !(HSU(qcap >= 0, false) && buf >= qcap);
            notify = acc && buf == 0;
            poll = (notify && active < conc) || dep;
            buf1 = buf + (acc ? 1 : 0);
            deliver = poll && buf1 > 0;
            pre_done = HSU(presend, true) && dep && completed < early_upto;     // the departing request's message went out ahead of time
        };
        derive();
        if constexpr (PF) {
            if (act && has_probe() && probe_at(t)) cnt += 2;          // the rare roots: always the general path
            if (act && has_sched() && SA == t) cnt += 2;
            if (act && has_xsrc() && xsrc_at(t)) cnt += 2;
            if (tick && prof_kind != kProfConstant) cnt += 2;         // (its next arrival is a numerical inversion)
        }
        const bool slow = act && (force_general || cnt != 1 || (tick && (a2 <= t || (poisson && na == 0))) ||
                          (deliver && (dur == 0 || (svc_exp && nsv == 0) || (dep && started >= win_hi))) || (dep && router && rn == 0) ||
                          (to_link && (target != fl_link || HSU(fl_loss > 0.0, false) || (!pre_done && HSU(fl_jit == 0, true) && nj == 0))));
#ifdef HS_RINGSTAT
        if (slow) stat_slow = 1;
#endif
        if (__builtin_expect(slow, 0)) run_group(t, force_general);
        if (slow) { tick = dep = msg = false; }                       // (handled; the predicated code below does nothing for it)
        derive();
        HS_CY2(0)
        // ---- Source.handle_event
        ev[0] += tick; generated += tick;
        arr_time = tick ? a2 : arr_time;
        if constexpr (UNI) arr_d = tick ? (double)a2 : arr_d;
        A = tick ? a2 : A;
        seqA = tick ? seq : seqA;
        // lineage (hs_station.hpp step_c1): the group's one root is the tick, the message (created at its send time, read from the bag
        // together with its created_at below -- as a read of its own up here it sat on the trip's critical path: +0.36 ms on the
        // headline ring) or the departure
        const int64_t crtA_root = crtA, crtD_root = crtD[0];
        rcA = tick ? crtA : rcA;
        dpA = tick ? 1 : dpA;
        crtA = tick ? t : crtA;
        seq += tick ? 1u : 0u;
        if (tick && poisson) { ha = (ha + 1) & (kNRing - 1); --na; }
        // ---- NetworkLink continuation at the egress side: the message leaves the bag
        int64_t created_in = t, msg_ts = 0;
        if (msg) {
            ev[9]++;
            created_in = bg_cr(mi);
            msg_ts = bg_ts(mi);
            if (UNI || bg_link(mi) == fi_link) fi_packets++; else ns->link_packets[bg_link(mi)]++;
            bag_remove(mi);
        }
        // ---- Queue._handle_enqueue / QueueDriver._handle_notify
        ev[1] += arrv;
        dropped += (arrv && !acc) ? 1 : 0;
        if (acc) {
            if (accepted < cap) adm[accepted * lgs] = created_in; else overflow = 1;
            window_admit(created_in);
        }
        if (acc && HSU(presend, true) && early_upto == accepted) {
            // pre-send at admission: the request's start and departure are already determined (see `early_upto`)
            const int off = (int)(accepted - started);                 // requests that start before it
            if (!svc_exp || off < nsv) {
                const double s_k = svc_exp ? fl.ring_s[(hs_ + off) & (kNRing - 1)][tid] : svc_const_s;
                const int64_t dur_k = svc_exp ? ns_i(s_k) : svc_const_ns;
                const int64_t s_at = t > D_pre ? t : D_pre;
                if (dur_k > 0) (void)pre_send(accepted, (int)(accepted - completed), s_at + dur_k, created_in, s_at);
            }
        }
        accepted += acc;
        ev[2] += notify;
        HS_CY2(1)
        // ---- worker continuation: statistics, then the forwarded request's way out
        int64_t created_out = 0;
        if (dep) {
            ev[6]++; completed++;
            total_service = __dadd_rn(total_service, svc_s[0]);
            created_out = crt[0];
            active = active > 0 ? active - 1 : 0;
            D[0] = kInfNs;
            if (router) { ev[10]++; routed++; rbits >>= 2; --rn; }
        }
        if (to_sink) {
            ev[7]++;
            if (received < cap) { sink_t[received * lgs] = t; sink_created[received * lgs] = created_out; } else overflow = 1;
            received++;
        }
        if (to_link) {                                                // send_link_fast without the loss branch
            ev[8]++; fl_in++; fl_sent++;
            if (!pre_done) {
                double delay = fl_delay0;
                if (HSU(fl_jit == 0, true)) {
                    delay = __dadd_rn(delay, fl.ring_j[hj][tid]);
                    hj = (hj + 1) & (kNRing - 1); --nj;
                }
                if (!(delay > 0.0)) delay = 0.0;
                fl_append(t + ns_i(delay), t, created_out, lin_pack(link_steps(), crtD_root, t));   // (the departure is the root)
            }
        }
        if (dep && early_upto < completed) { early_upto = completed; D_pre = t; }   // (left the ordinary way)
        HS_CY2(2)
        // ---- QUEUE_POLL, then QUEUE_DELIVER + the retargeted payload at the worker
        ev[3] += poll;
        buf = buf1 - (deliver ? 1 : 0);
        if (deliver) {
            ev[4]++; ev[5]++;
            const int64_t k = started++;
            active++;
            int64_t created = created_in;                             // arrival side: the request that found the buffer empty
            if (dep) created = fl.crc[k & (kNRing - 1)][tid];        // in the window (k < win_hi: checked with `slow`)
            win_hi = win_hi <= k ? k + 1 : win_hi;
            if (HSU(presend, true) && k >= early_upto) (void)pre_send(k, (int)(k - completed), t + dur, created, t);   // pre-send at the start
            svc_s[0] = s_new; crt[0] = created;
            D[0] = t + dur; seqD[0] = seq++; crtD[0] = t;
            dpD[0] = dep ? 4 : 6; rcD[0] = dep ? crtD_root : tick ? crtA_root : msg_ts;
            if (svc_exp) { hs_ = (hs_ + 1) & (kNRing - 1); --nsv; }
        }
        last_time = (tick || dep || msg) ? t : last_time;            // (the general path sets it itself)
        HS_CY2(3)
    }

    // Totals::undecided bit 2 (as Station::pick_root, hs_station.hpp): a root constructed BEFORE run() -- a first tick, a
    // Probe's first tick, a scheduled Request -- beside any other root of this nanosecond (a message counts).  Their order
    // is what the prologue's true sort indices decide; an engine that skipped the prologue repeats the run behind it
    // (hs_engine.hip lazy_prologue).  Lone roots cannot tie; every group with two roots comes through run_group() -- or is the
    // group of the one event beyond end_time, which the election enters through root_first() (round 6: the election did not
    // look, tools/gpu_random_sweep.py multi_source_ring_windows_async 132469: a Probe's first tick on the nanosecond of a
    // constant Source's tick, first event beyond a window end).
    __device__ __forceinline__ void pre_run_hazard(int64_t t) {
        if constexpr (PF) {
            int cnt = 0;
            bool pre = false;
            if (A == t) { ++cnt; pre = rcA == INT64_MIN; }
#pragma unroll
            for (int i = 0; i < C; ++i) cnt += (D[i] == t) ? 1 : 0;
            if (bmin == t) for (int i = 0; i < bag_n; ++i) cnt += bg_t(i) == t ? 1 : 0;
#pragma unroll
            for (int j = 0; j < kMaxProbes; ++j) if (j < n_probes && PA[j] == t) { ++cnt; pre = pre || rcP[j] == INT64_MIN; }
            if (has_sched() && SA == t) { ++cnt; pre = true; if (sc_i + 1 < sc_end && sc_t[sc_i + 1] == t) ++cnt; }
            if (has_xsrc() && xsrc_at(t))
                for (int j = 0; j < n_xsrc; ++j) if (xx->XA[xo(j)] == t) { ++cnt; pre = pre || xx->rcX[xo(j)] == INT64_MIN; }
            if (pre && cnt >= 2) undecided |= 4;
        }
    }
    __device__ __forceinline__ void run_group(int64_t t, bool force_general) {
        int n_at = (A == t) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < C; ++i) n_at += (D[i] == t) ? 1 : 0;
        int mi = -1;
        if (bmin == t)                    // (the bag's earliest arrival is in a register: no scan for local-only groups)
            for (int i = 0; i < bag_n; ++i) if (bg_t(i) == t) { ++n_at; mi = i; }
        if (has_probe() && probe_at(t)) n_at += 2;                       // a probe tick: always the general path
        if (has_sched() && SA == t) n_at += 2;                           // so is a scheduled Request
        if (has_xsrc() && xsrc_at(t)) n_at += 2;                         // and a tick of a further Source
        pre_run_hazard(t);
        if (n_at == 1 && !force_general) {
            bool general = false, want_poll = false, have_created = false;
            int64_t created = 0;
            cd = 0;
            if (A == t) {
                cr = crtA;
                const uint32_t r = do_tick(t);
                if (r & 2u) { if (r & 1u) push_enq(t); qpush(Q_TICK); general = true; }
                else if (r & 1u) { want_poll = do_enqueue(t, t) && do_notify(); have_created = true; created = t; cd = 3; }   // tick -> Request -> QUEUE_NOTIFY -> QUEUE_POLL
            } else if (mi >= 0) {
                cr = bg_ts(mi);                                          // the link's continuation was created when the message was sent
                created = do_msg(mi, t);
                want_poll = do_enqueue(t, created) && do_notify();
                have_created = true;
                cd = 3;                                                  // continuation -> Request -> QUEUE_NOTIFY -> QUEUE_POLL
            } else {
                int slot = 0;
#pragma unroll
                for (int i = 0; i < C; ++i) if (D[i] == t) slot = i;
#pragma unroll
                for (int i = 0; i < C; ++i) if (i == slot) cr = crtD[i];
                want_poll = do_cont(slot, t);
                cd = 1;                                                  // continuation -> QUEUE_POLL
            }
            // `have_created` is only valid when the buffer was empty before this chain's enqueue, which is exactly
            // when do_enqueue returned true (was_empty) -- the only way want_poll is set on the arrival branches.
            if (want_poll) general = chain_from_poll(t, have_created, created);
            if (general) drain(t);
        } else {
            run_group_general(t);
        }
        last_time = t;
    }
};

}  // namespace hs
#undef HSU
