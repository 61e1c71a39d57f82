/* hs_engine.h -- C ABI of the MI355X discrete-event engine (libhs_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (adamfilli/happy-simulator, pure Python) has no FFI; the seam these entry
 * points replace is `Simulation._execute_until(end_time_ns)` plus
 * `_build_summary()` (happysimulator/core/simulation.py:449-505, :543-591)
 * for the lowered entity set {Source, Server(QueuedResource), Sink}, and the
 * replica / partition fan-out of happysimulator/parallel (runner.py:82-142,
 * simulation.py:164-223).  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller-owned host buffers (the engine
 * copies; it never frees or retains caller memory), no exceptions or aborts
 * across the ABI.  Every function returns HS_OK (0) or a negative hs_status;
 * hs_last_error() gives the message.  A handle is not thread-safe.  There is
 * NO CPU fallback: if no gfx950 device is present hs_engine_create() fails.
 */
#ifndef HS_ENGINE_H
#define HS_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_ABI_VERSION 16

typedef enum hs_status {
    HS_OK = 0,
    HS_E_INVALID = -1,      /* bad argument / configuration (the reference raises ValueError) */
    HS_E_NO_DEVICE = -2,    /* no HIP device: the product path has no CPU fallback */
    HS_E_HIP = -3,          /* HIP runtime error */
    HS_E_UNSUPPORTED = -4,  /* entity / topology not lowered by this engine (never silent) */
    HS_E_OVERFLOW = -5,     /* a per-LP record log overflowed its capacity; re-create with a larger one */
    HS_E_STATE = -6         /* call order violation */
} hs_status;

/* How `Simulation`s map onto logical processes (LPs).
 * SINGLE:   all LPs belong to ONE Simulation (one heap): exactly one event beyond
 *           end_time is processed in total (core/simulation.py:472), the globally first.
 * REPLICAS: every LP is its own Simulation (ParallelRunner, parallel/runner.py:73-79, or
 *           ParallelSimulation without links, parallel/simulation.py:170-195): each LP
 *           processes its own first event beyond end_time. */
typedef enum hs_mode { HS_MODE_SINGLE = 0, HS_MODE_REPLICAS = 1 } hs_mode;

/* load/source.py:182-268 factories */
typedef enum hs_source_kind { HS_SRC_NONE = 0, HS_SRC_POISSON = 1, HS_SRC_CONSTANT = 2 } hs_source_kind;
/* distributions/{exponential,constant}.py */
typedef enum hs_latency_kind {
    HS_LAT_EXPONENTIAL = 0,
    HS_LAT_CONSTANT = 1,
    HS_LAT_NO_SERVER = 2   /* svc_kind only: the LP has no Server, its Source feeds the Sink/Counter directly */
} hs_latency_kind;
/* Server(downstream=...): nothing, a Sink-like collector, a NetworkLink to another station's Server
 * (components/network/link.py:114), or a RandomRouter over {the station's Sink, NetworkLinks}
 * (components/random_router.py:9).  LINK and ROUTER need hs_engine_set_network. */
typedef enum hs_egress_kind {
    HS_EGRESS_NONE = 0,
    HS_EGRESS_SINK = 1,
    HS_EGRESS_LINK = 2,
    HS_EGRESS_ROUTER = 3,
    HS_EGRESS_SERVER = 4   /* hs_stations.egress only: another Server (hs_stations.downstream_lp), tandem queues */
} hs_egress_kind;

/* reference-equivalent event kinds counted by the engine */
enum {
    HS_EV_SOURCE = 0,       /* SourceEvent @ Source                         */
    HS_EV_ENQUEUE = 1,      /* Request @ Server (QueuedResource.handle_event) */
    HS_EV_NOTIFY = 2,       /* QUEUE_NOTIFY @ driver                        */
    HS_EV_POLL = 3,         /* QUEUE_POLL @ queue                           */
    HS_EV_DELIVER = 4,      /* QUEUE_DELIVER @ driver                       */
    HS_EV_WORK = 5,         /* Request @ worker (handle_queued_event start) */
    HS_EV_CONTINUATION = 6, /* ProcessContinuation @ worker                 */
    HS_EV_SINK = 7,         /* Request @ Sink                               */
    HS_EV_LINK = 8,         /* Request @ NetworkLink (transit starts)       */
    HS_EV_LINK_CONT = 9,    /* ProcessContinuation @ NetworkLink (transit over) */
    HS_EV_ROUTE = 10,       /* Request @ RandomRouter                       */
    HS_EV_LB = 11,          /* Request @ LoadBalancer (forwarded to the selected backend)        */
    HS_EV_LB_RESP = 12,     /* _lb_response @ LoadBalancer (completion hook of the forwarded Request) */
    HS_EV_PROBE_TICK = 13,  /* SourceEvent @ Probe (instrumentation/probe.py:81-164)                         */
    HS_EV_PROBE = 14,       /* probe_event (daemon) @ the measurement callback                                */
    HS_EV_KINDS = 15
};

typedef struct hs_config {
    uint32_t struct_size;   /* sizeof(hs_config), for ABI evolution */
    int32_t device;         /* HIP device ordinal */
    int32_t n_lp;           /* station LPs resident on this engine (this GPU's shard) */
    int32_t mode;           /* hs_mode */
    int64_t start_ns;       /* Simulation start_time (Instant.Epoch = 0) */
    int64_t horizon_ns;     /* latest end_time that will be passed to hs_engine_run_until: sizes the record logs */
    uint64_t seed;          /* Philox key of the run (per-LP override: hs_stations.seed) */
    uint64_t lp_base;       /* global index of LP 0: default stream_base of LP i is lp_base + i */
    int64_t log_capacity;   /* records per LP in the admission / sink logs; 0 = derive from rate * horizon */
} hs_config;

/* One station LP = [optional Source] -> Server(c, FIFO, capacity) -> [Sink].
 * Struct-of-arrays; every pointer is [n_lp] or NULL for the documented default.
 * Field meanings and defaults follow the reference constructors:
 *   Source.poisson/constant(rate, stop_after)            load/source.py:182-268
 *   Server(concurrency=1, service_time=ConstantLatency(0.01), queue_capacity=None, downstream=None)
 *                                                        components/server/server.py:64-122 */
typedef struct hs_stations {
    const uint8_t *src_kind;           /* hs_source_kind; NULL = HS_SRC_POISSON */
    const double *src_rate;            /* events/s; required when any source exists */
    const int64_t *src_stop_after_ns;  /* < 0 = never; NULL = never */
    const int32_t *concurrency;        /* 1..32; NULL = 1 */
    const uint8_t *svc_kind;           /* hs_latency_kind; NULL = HS_LAT_CONSTANT */
    const double *svc_mean_s;          /* NULL = 0.01 */
    const int64_t *queue_cap;          /* < 0 = unbounded; NULL = unbounded */
    const uint8_t *egress;             /* hs_egress_kind; NULL = HS_EGRESS_SINK */
    const uint64_t *seed;              /* per-LP Philox key (replica i: base_seed + i); NULL = cfg.seed */
    const uint64_t *stream_base;       /* per-LP stream id base; NULL = cfg.lp_base + i */
    /* Time-varying arrival rate (Source.with_profile, load/source.py:271-320; load/profile.py:52-113): the next arrival
     * is found by the reference's own numerical procedure -- adaptive Simpson + bracket search + Brent
     * (load/arrival_time_provider.py:84-144) -- restated on the device (csrc/hs_profile.hpp).  src_rate of such a source
     * is its PEAK rate (used to size the record logs).  NULL = every source has a ConstantRateProfile(src_rate). */
    const uint8_t *src_profile_kind;   /* hs_profile_kind */
    const double *src_profile_params;  /* [n_lp][4]: LINEAR_RAMP {duration_s, start_rate, end_rate, -};
                                          SPIKE {baseline_rate, spike_rate, warmup_s, spike_duration_s} */
    /* Probe(target, metric, interval) attached to the LP's Source / Server / Sink (instrumentation/probe.py:81-164):
     * a daemon Source of its own that samples getattr(target, metric) every `interval` seconds (tick times follow
     * ConstantArrivalTimeProvider over _ProbeProfile: the general numerical path, like the reference).  Each tick is two
     * reference events (SourceEvent@Probe, probe_event).  Up to four probes per LP (probe_metric_more).  NULL = no probes. */
    const uint8_t *probe_metric;       /* hs_probe_metric; 255 = none */
    const double *probe_interval_s;    /* > 0 */
    /* Simulation.schedule(Event(time, "Request", target=<the LP's Server>)) before run() (core/simulation.py:195-206):
     * LP i receives the Requests sched_time_ns[sched_off[i] .. sched_off[i + 1]), ascending per LP and, among equal
     * times, in the order the caller built the Events; context["created_at"] = the Event's own time (core/event.py:176).
     * An Event built before run() precedes every run-time event of the same nanosecond on its LP.  NULL = none. */
    const int64_t *sched_off;          /* [n_lp + 1] */
    const int64_t *sched_time_ns;      /* [sched_off[n_lp]], each >= start_ns */
    /* The order in which the reference CONSTRUCTS its pre-run events -- the first SourceEvent of every Source in
     * `sources=[...]` order, the first tick of every Probe in `probes=[...]` order (Simulation.__init__,
     * core/simulation.py:145-160), then the Events handed to schedule() in the order the caller built them -- fixes their
     * `_sort_index` values 0 .. N_init-1 from the process-wide counter (core/event.py:53-77), while run() numbers its own
     * events from 0 again (core/event_heap.py:48).  In HS_MODE_SINGLE, when probes or scheduled Requests exist, the engine
     * runs the first N_init constructions of the run in the reference's exact heap order (csrc/hs_exact.hpp) so that every
     * same-nanosecond meeting of the two counters is decided as the reference decides it.  All three NULL = LP order /
     * array order. */
    const int32_t *source_order;       /* [number of Sources] LP indices in `sources=` order (also the last key of the election of
                                        * the one event beyond end_time: a pending tick ranks by its own Source's position here) */
    const int32_t *probe_order;        /* [number of LPs with a Probe] LP indices in `probes=` order */
    /* More than one Probe on an LP (Probe.on_many, instrumentation/probe.py:119-164): slots 1 .. 3 (slot 0 = probe_metric /
     * probe_interval_s above); slots are filled from 0.  probe_slot_order[k] = slot of the k-th entry of probe_order (an LP
     * with several probes appears several times there); NULL = every entry is slot 0. */
    const uint8_t *probe_metric_more;  /* [3][n_lp] hs_probe_metric; 255 = none */
    const double *probe_interval_more; /* [3][n_lp] */
    const uint8_t *probe_slot_order;   /* [number of probes] */
    const int64_t *sched_rank;         /* [sched_off[n_lp]], indexed like sched_time_ns: the Event's position among ALL the
                                          Events the caller constructed for schedule(), distinct and >= 0 (an Event that was
                                          cancelled before run() keeps its position -- it consumed a sort index -- but is not
                                          passed to the engine); NULL = array order */
    /* Several Sources feeding one Server (`Source.poisson(rate, target=server)` more than once; load/source.py:142-180 runs per
     * Source): slots 1 .. 3 of the LP (slot 0 = src_kind / src_rate / src_stop_after_ns above); slots are filled from 0.  Each
     * is an entity of its own with its own arrival stream -- stream base (1 << 40) | (stream_base[lp] << 2) | (slot - 1), kind
     * ARRIVAL -- constant or Poisson rate (no profile).  An LP with more than one Source runs on the engines' general path, and
     * the run starts with the prologue (csrc/hs_exact.hpp): the first ticks of an LP's Sources carry consecutive pre-run sort
     * indices, which run-time events of the same nanosecond can overtake.  source_slot_order[k] = slot of the k-th entry of
     * source_order (an LP with several Sources appears several times there); NULL = every entry is slot 0.  Also on networked
     * stations (both network engines and shards).  NULL = one Source per LP at most.
     * The election of the one event beyond end_ns: a pending DEPARTURE of a Server with several Sources ranks by its LP's
     * first-listed Source -- a stand-in for the Source its lineage goes back to.  A station engine (HS_MODE_SINGLE) notices when the
     * election comes down to that key and repeats the run on the single-heap loop (exact, one lane: hs_engine_prologue_path() == 2
     * afterwards).  So does a station NETWORK held by one engine (since the round of ABI 14: runs of up to 4 000 000 events; the
     * single-heap machinery is built on demand for models without pre-run events, and hs_engine_reset returns to the parallel
     * engines); longer runs and shards of a partitioned network detect such an election and refuse it by name (HS_E_UNSUPPORTED). */
    const uint8_t *src_more_kind;      /* [3][n_lp] hs_source_kind; HS_SRC_NONE = none */
    const double *src_more_rate;       /* [3][n_lp] */
    const int64_t *src_more_stop_after_ns; /* [3][n_lp] < 0 = never; NULL = never */
    const uint8_t *source_slot_order;  /* [number of Sources] */
    /* Tandem queues: `Server(..., downstream=<another Server>)` (components/server/server.py:64-122,271-272; the forwarded Event
     * keeps its context, core/entity.py:83-105).  egress[i] == HS_EGRESS_SERVER: every completion of LP i arrives at the Server of
     * LP downstream_lp[i] at the same instant, created_at unchanged.  At most 7 Servers in a row, no cycles, HS_MODE_SINGLE, no
     * hs_engine_set_network.  A Server may have several upstream Servers: up to four merge on the passes, more run on the single-heap
     * loop from the start (hs_engine_tandem_path() == 2).  Tandem queues next to Probes / scheduled Requests / several Sources per
     * Server start on the passes and move to the single heap when a pre-run event shares its nanosecond with another event of its LP
     * (hs_engine_prologue_path).  The engine runs the chain in passes, upstream first (csrc/hs_station.hpp "tandem queues"); results
     * equal the reference's single heap event for event, ties inside a nanosecond included.  NULL = no such Server. */
    const int32_t *downstream_lp;      /* [n_lp] read where egress == HS_EGRESS_SERVER */
} hs_stations;
typedef enum hs_probe_metric {
    HS_PROBE_DEPTH = 0,        /* QueuedResource.depth */
    HS_PROBE_ACTIVE = 1,       /* Server.active_requests */
    HS_PROBE_ACCEPTED = 2,     /* stats_accepted */
    HS_PROBE_DROPPED = 3,      /* stats_dropped */
    HS_PROBE_COMPLETED = 4,    /* Server._requests_completed */
    HS_PROBE_RECEIVED = 5,     /* Sink.events_received */
    HS_PROBE_GENERATED = 6,    /* Source.generated_count */
    HS_PROBE_NONE = 255
} hs_probe_metric;
typedef enum hs_profile_kind { HS_PROF_CONSTANT = 0, HS_PROF_LINEAR_RAMP = 1, HS_PROF_SPIKE = 2 } hs_profile_kind;

/* Links between stations (the engine-side form of the reference's partition links, parallel/link.py:18-79,
 * and of `NetworkLink(latency=ConstantLatency(lat_min), jitter=ExponentialLatency(jitter_mean) | None,
 * egress=<Server of station dst>)`).  The smallest constant latency over all links is the lookahead W of the
 * conservative time windows; it must be > 0 (the reference enforces min_latency > 0 the same way). */
typedef struct hs_network {
    const uint8_t *egress_kind;      /* [n_lp] hs_egress_kind; replaces hs_stations.egress */
    const int32_t *router_target0;   /* [n_lp] RandomRouter targets in constructor order: -1 = the station's Sink, */
    const int32_t *router_target1;   /* [n_lp]   >= 0 = link index; used when egress_kind == HS_EGRESS_ROUTER */
    const int32_t *link_of;          /* [n_lp] link index when egress_kind == HS_EGRESS_LINK */
    const uint64_t *router_stream_base; /* [n_lp] NULL = the station's stream base */
    int32_t n_links;
    const int32_t *link_dst;         /* [n_links] destination station (its Server) */
    const double *link_lat_min_s;    /* [n_links] ConstantLatency seconds, > 0 */
    const uint8_t *link_jitter_kind; /* [n_links] HS_LAT_EXPONENTIAL: jitter = ExponentialLatency(mean), one draw of the link's stream per
                                      * packet; HS_LAT_CONSTANT: jitter = ConstantLatency(mean), no draw (link.py:195-200); mean 0 = jitter=None */
    const double *link_jitter_mean_s;/* [n_links] */
    const uint64_t *link_stream_base;/* [n_links] NULL = stream base of the source station */
    const int32_t *link_src;         /* [n_links] source station (for the default stream base and validation) */
    int32_t bag_capacity;            /* in-flight messages per destination station; 0 = 16 */
    /* --- one shard of a network partitioned over several engines (one per GPU); 0 / NULL = the whole network ---
     * The engine owns stations [cfg.lp_base, cfg.lp_base + cfg.n_lp) of n_global_lp; link_src / link_dst are then
     * NETWORK-WIDE station indices, the link table holds every link that starts or ends in the shard, and
     * link_gid gives each link its network-wide id.  This is the engine-side form of the reference's
     * SimulationPartition / PartitionLink split (parallel/partition.py:21-38, parallel/link.py:18-79). */
    int32_t n_global_lp;
    const int64_t *link_gid;         /* [n_links] */
    int64_t n_global_links;
    /* NetworkLink(packet_loss_rate) (components/network/link.py:131-138): every request entering the link is lost
     * with this probability (u of the link's LOSS stream < rate, where the reference asks the process-wide
     * `random.random()`); in [0, 1].  NULL = lossless.  NetworkLink(bandwidth_bps) needs no field: the lowered event
     * providers put no payload_size into the metadata, so the transmission time (link.py:209-214) is 0 for any
     * bandwidth and bytes_transmitted stays 0, exactly as in the reference. */
    const double *link_loss_rate;    /* [n_links] */
    /* RandomRouter(targets=[...]) with other than two targets (components/random_router.py:32-45): the route draw picks
     * targets[int(u * len(targets))].  1..4 targets, at most two of them NetworkLinks (-1 = the station's Sink, which may
     * appear several times).  NULL = every router has exactly two targets (router_target0 / router_target1). */
    const uint8_t *router_n_targets; /* [n_lp] */
    const int32_t *router_target2;   /* [n_lp] */
    const int32_t *router_target3;   /* [n_lp] */
    /* Links whose losses are decided by a TABLE instead of a stream (ABI 14): `PartitionLink(packet_loss=p)`
     * (parallel/link.py:31-39) is applied by the reference's coordinator at the exchange with ONE `random.Random(seed)`
     * for the whole run, drawn in exchange order (parallel/coordinator.py:68,203-205) -- a sequence the caller replays
     * over the cross-partition sends of a run (hs_engine_read_send_log) and hands back as one bit per packet
     * (hs_engine_set_link_drops).  link_drop_capacity[l] > 0: link l has a table of that many packets (all zero = nothing
     * is lost until the caller says so); a packet beyond it is HS_E_OVERFLOW.  Such a link must have link_loss_rate 0.
     * NULL = no link has one. */
    const int64_t *link_drop_capacity; /* [n_links] */
} hs_network;

/* Exchange buffers of a shard: device memory owned by the caller (torch tensors on the host side, so that
 * torch.distributed / RCCL can move them).  A row is {count, then 5 x int64 per message}: arrival ns, send ns,
 * created_at ns, (destination station << 32 | link gid), lineage (csrc/hs_netstation.hpp lin_pack).  Replaces the reference's per-partition Python outbox lists
 * (parallel/simulation.py:107,142-151) and `_exchange_events` (parallel/coordinator.py:182-227). */
typedef struct hs_shard {
    int32_t rank, world;
    const int64_t *shard_lo;         /* host, [world + 1]: rank r owns stations [shard_lo[r], shard_lo[r+1]) */
    int64_t *outbox_dev;             /* device int64 [world][1 + 5 * msg_capacity]: row r = messages for rank r */
    int64_t *inbox_dev;              /* device, same shape: row r = messages from rank r (after the all-to-all) */
    int32_t msg_capacity;
    int32_t reserved;
    int64_t window_ns;               /* lookahead W = min over ALL shards of hs_summary.window_ns (caller all-reduces) */
    int64_t *gvt_dev;                /* device int64[2]: window k accumulates this rank's earliest pending work into
                                        [k & 1]; the caller all-reduces (min) it before window k + 1 reads it */
    int64_t *cand_dev;               /* device int64[8]: {valid, t, t_created, station, steps from its group's root, that root's
                                        creation time, construction rank among this shard's entities, kind}: this rank's first
                                        event beyond end_ns; the ranks' candidates compare by (t, t_created, steps, root time,
                                        construction rank) = (time, _sort_index), core/event.py:337-344.  The rank in [6] only
                                        orders entities of ONE shard; across shards the caller derives the network-wide rank from
                                        (station, kind): kind 0 = a departure / message / injected Request (ranks like the
                                        station's first-listed Source), 2 + s = the tick of the station's Source in slot s,
                                        8 + s = the tick of its Probe in slot s (happy_simulator_amd/sharded.py election_rank) */
} hs_shard;

typedef struct hs_net_stats {
    int64_t *routed;                 /* [n_lp]    RandomRouter.stats_routed            components/random_router.py:36 */
    int64_t *link_entered;           /* [n_links] requests that entered the link (Request@Link events) */
    int64_t *link_packets_sent;      /* [n_links] NetworkLink.packets_sent             components/network/link.py:162 */
    int64_t *link_packets_dropped;   /* [n_links] NetworkLink.packets_dropped          components/network/link.py:132 */
} hs_net_stats;

typedef struct hs_summary {
    /* SimulationSummary fields (instrumentation/summary.py:47-87), engine-wide */
    int64_t events_processed;              /* total_events_processed */
    int64_t events_by_kind[HS_EV_KINDS];
    int64_t events_cancelled;              /* always 0 on this path */
    int64_t final_time_ns;                 /* SINGLE: time of the last processed event; REPLICAS: max over LPs */
    int64_t requests_completed;            /* sum of Server._requests_completed */
    int64_t sink_records;                  /* sum of Sink.events_received */
    /* engine telemetry */
    double last_run_ms;                    /* device time of the last hs_engine_run_until (HIP events) */
    double kernel_ms;                      /* device time of the dominant kernel(s) in that call */
    int64_t launches;                      /* kernel launches in the last run (network engine: windows + 1) */
    int64_t window_ns;                     /* network engine: lookahead W; 0 otherwise */
    int32_t overflow;                      /* 1 if a record log overflowed */
    int32_t reserved;
} hs_summary;

/* Per-LP results, SoA out-buffers [n_lp]; any pointer may be NULL. */
typedef struct hs_lp_stats {
    int64_t *generated;        /* Source._generated_count            load/source.py:159 */
    int64_t *accepted;         /* Queue.stats_accepted               components/queue.py:138 */
    int64_t *dropped;          /* Queue.stats_dropped                components/queue.py:128 */
    int64_t *completed;        /* Server._requests_completed         server/server.py:256 */
    int64_t *rejected;         /* Server._requests_rejected          server/server.py:233 */
    double *total_service_s;   /* Server._total_service_time         server/server.py:257 */
    int64_t *sink_received;    /* Sink.events_received               components/common.py:37 */
    int64_t *queue_depth;      /* QueuedResource.depth */
    int32_t *active;           /* Server.active_requests */
    int64_t *events;           /* events processed by this LP (REPLICAS: that replica's total_events_processed) */
    int64_t *final_time_ns;    /* time of the LP's last processed event */
} hs_lp_stats;

typedef struct hs_engine hs_engine;

int hs_abi_version(void);
/* Build identity: the hash of the sources (csrc/, this header, the compiler flags) the library was built from, compiled into the
 * library itself; the host side rebuilds when it differs from the sources next to it (happy_simulator_amd/_native.py). */
const char *hs_build_sources_hash(void);
/* Number of visible HIP devices (0 when there is no GPU). */
int hs_device_count(void);

/* Tandem queues (hs_stations.downstream_lp): which path the engine is on -- 0 no tandem queues, 1 passes of the station kernel
 * (upstream Servers first), 2 the single-heap loop (csrc/hs_exact.hpp): hs_engine_run_until repeats a run there when the passes
 * met a same-nanosecond order between two Servers' events that their lineage key does not decide (lock-step constant arrivals
 * and services, comparisons around the Sources' first ticks), and a Server with more than four upstream Servers starts there. */
int hs_engine_tandem_path(const hs_engine *h);

/* The prologue (csrc/hs_exact.hpp): the reference numbers the events constructed before run() -- the Sources' and Probes' first
 * ticks, Requests injected with Simulation.schedule() -- first and restarts the count for the run's own events
 * (core/simulation.py:77,145-160), so a run-time event can sort BEFORE a pre-run event of its nanosecond.  Engines with Probes,
 * scheduled Requests or several Sources per Server replay that on a single-heap loop -- one lane for the whole engine.  A station
 * engine and (since ABI 12's round) a network engine driven with hs_engine_run_until skip it at first and repeat the run behind it
 * only if a pre-run event shared its nanosecond with another event of its LP (an arriving message counts), or the run was shorter
 * than the pre-run events are many; shards of a partitioned network (hs_engine_shard_*) keep it.  0: the engine has no prologue,
 * 1: skipped so far, 2: it runs (debug flag 1 << 16 makes it run always). */
int hs_engine_prologue_path(const hs_engine *h);

/* Windows over a station NETWORK (ABI 14): `Simulation.control.run_until / _run_window` (core/simulation.py:527-541) call
 * `_execute_until` again with a later end, and the reference continues from its heap -- O(window).  What the last hs_engine_run_until on a
 * network engine did: 0 the first run since the reset, 1 it CONTINUED from the state the run before it left (every station's rows, the
 * messages in flight, the links' lower bounds, and the timestamp group the election of the one event beyond the earlier end stopped
 * inside -- finished first, csrc/hs_kernels.hpp hs_net_resume), 2 nothing moved (the end is not beyond the event the earlier run
 * already processed: the reference's loop condition `current_time <= end` is false), 3 the run was REPEATED from the start to the
 * new end (an engine whose prologue -- Probes, scheduled Requests, several Sources per Server -- still holds the single-lane heap at
 * the window end, a state the asynchronous kernel cannot take back, debug flag 1 << 24): exact as well, at the cost of the prefix.
 * Diagnostics only; results are identical. */
int hs_engine_window_path(const hs_engine *h);

int hs_engine_create(const hs_config *cfg, hs_engine **out);
int hs_engine_set_stations(hs_engine *h, const hs_stations *st);
/* Optional, after hs_engine_set_stations and before the first run: connect stations with links / routers.
 * Requires HS_MODE_SINGLE.  The engine then advances in conservative windows of W = min link latency; one
 * hs_engine_run_until per hs_engine_reset (windows are internal, the call is not re-entrant). */
int hs_engine_set_network(hs_engine *h, const hs_network *net);
int hs_engine_get_net_stats(hs_engine *h, const hs_net_stats *out);
/* Table-decided link losses (hs_network.link_drop_capacity; replaces the `self._rng.random() < link.packet_loss` of
 * parallel/coordinator.py:203-205).  hs_engine_set_link_drops: packet number e of `link` (a local link index) is lost iff
 * bit e of `bits` is set (n_bits <= the link's capacity; the rest stays 0); takes effect at the next reset / run.
 * hs_engine_read_send_log: every packet that entered a table-decided link since the last reset, as int64 triples
 * {send time ns, network-wide link id, packet number}, in no particular order; returns their number (the first
 * `capacity` are written) or a negative hs_status. */
int hs_engine_set_link_drops(hs_engine *h, int32_t link, const uint32_t *bits, int64_t n_bits);
int64_t hs_engine_read_send_log(hs_engine *h, int64_t *out_triples, int64_t capacity);
/* external != 0: run on the caller's HIP stream (e.g. torch's current stream, so that collectives and engine
 * launches are ordered by the stream; a NULL handle is the device's default stream).  external == 0: back to the
 * engine's own stream. */
int hs_engine_set_stream(hs_engine *h, void *hip_stream, int external);
/* Sharded network = WindowedCoordinator.run (parallel/coordinator.py:75-172) with one partition per GPU and
 * GVT-driven windows.  After create / set_stations / set_network(n_global_lp > 0) / shard_attach, per run:
 *   shard_begin(end);
 *   for k = 0, 1, 2, ...:  shard_window(k)            EXECUTE: window end = min(end, max(prev+1, GVT) + W - 1)
 *                          all-to-all outbox -> inbox   EXCHANGE (caller: RCCL)
 *                          shard_inject(k)
 *                          all-reduce(min) gvt_dev[k&1] GVT     (caller: RCCL)
 *      until shard_progress(k) reports a window end == end;
 *   shard_final(k+1); all-gather cand_dev; shard_overshoot(station - lp_base) on the rank that owns the minimum.
 * All calls except shard_progress only enqueue work on the engine's stream. */
int hs_engine_shard_attach(hs_engine *h, const hs_shard *sh);
/* LIVE exchange (ABI 14): the EXCHANGE step of parallel/coordinator.py:182-227 without a launch boundary.  Every rank runs ONE
 * launch of the asynchronous engine for the whole run; a station whose link leaves the shard appends to the link's queue in the
 * DESTINATION rank's memory and publishes the link's lower bound there (system-scope stores over hipIpcOpenMemHandle mappings:
 * peer-to-peer over xGMI between GPUs, plain device memory when ranks share a device), the receiver polls its own memory.  A shard
 * engine's link queues are uncached, exportable device memory for this.  Per process:
 *   live_export -> all-gather the 3 handles per rank -> live_attach(all handles, peer_link)        (once)
 *   per run: shard_begin(end); <barrier of the caller's: every rank has reset>; live_run(); live_wait();
 *            shard_final(0); all-gather cand_dev; shard_overshoot on the winner           (as for the other exchange paths)
 * peer_link[l] (local link index l): the link's index in the link table of the rank that owns its destination station (-1 when
 * that is this rank).  The launches wait for one another: together they must be resident on the device(s) -- no more workgroups
 * of 256 stations than CUs per device; every wait is bounded (HS_E_HIP instead of a hang).  One shard per PROCESS: launches of one
 * process on several streams may share a hardware queue and then run one after the other. */
int hs_engine_shard_live_export(hs_engine *h, void *handles_out);               /* 3 x HS_IPC_HANDLE_BYTES: records, words, positions */
int hs_engine_shard_live_attach(hs_engine *h, const void *all_handles, const int32_t *peer_link);
int hs_engine_shard_live_run(hs_engine *h);
int hs_engine_shard_live_wait(hs_engine *h);
int hs_engine_shard_begin(hs_engine *h, int64_t end_ns);
int hs_engine_shard_window(hs_engine *h, int64_t k);
int hs_engine_shard_inject(hs_engine *h, int64_t k);
int hs_engine_shard_progress(hs_engine *h, int64_t k_last, int64_t *window_end_out);
int hs_engine_shard_final(hs_engine *h, int64_t k);
int hs_engine_shard_overshoot(hs_engine *h, int32_t lp);
/* Asynchronous exchange ROUNDS instead of windows (same attach / begin / final / overshoot): every shard runs the
 * asynchronous engine (per-link lower bounds, hs_net_async) for `max_iters` iterations, then the caller moves the outbox
 * rows (all-to-all) and all-reduces (MAX) `bounds_dev` -- device int64[n_cross + 1]: the lower bound of every cross-shard
 * link (network-wide ids in cross_gid, the same list on every rank) and, last, an "I still have work" flag -- and calls
 * inject_async.  Rounds follow the boundary stations' lookahead (tens of ms of simulated time) instead of the smallest
 * link latency.  async_done synchronises and reports the all-reduced flag of the last round.
 *   begin; do { round; <all-to-all>; <all-reduce MAX bounds_dev>; inject_async; } while (any_not_done); final(0); ... */
int hs_engine_shard_async_setup(hs_engine *h, int32_t n_cross, const int64_t *cross_gid, int64_t *bounds_dev, int32_t max_iters);
int hs_engine_shard_round(hs_engine *h);
int hs_engine_shard_inject_async(hs_engine *h);
int hs_engine_shard_async_done(hs_engine *h, int32_t *any_not_done);
/* DEVICE-SIDE exchange between the rounds, replacing the all-to-all and the all-reduce of the bounds (the reference's exchange
 * step: parallel/coordinator.py:182-227 drains Python outboxes on the main thread).  Every rank exports the handles of its two
 * exchange buffers (hipIpcGetMemHandle; HS_IPC_HANDLE_BYTES each: inbox, bounds), the ranks all-gather them (2 x world handles in
 * rank order) and attach (hipIpcOpenMemHandle: the peers' buffers mapped into this process -- xGMI peer-to-peer across GPUs, plain
 * device memory when ranks share a GPU).  After a round, `push` writes this rank's outbox rows and link bounds straight into the
 * peers' buffers (system-scope stores, only the messages that exist); then ONE barrier -- the stream-ordered all-reduce of the
 * "still working" word, the only collective left on the path -- and `inject_ipc` takes what the peers pushed (system-scope loads,
 * element-wise max of the bounds) and injects it as inject_async does.  The buffers are double-buffered by round parity, which is
 * what makes one barrier per round enough.
 *   export; <all-gather handles>; attach; begin; do { round; push; <all-reduce MAX of 1 word>; inject_ipc; } while (any_not_done); final(0) */
#define HS_IPC_HANDLE_BYTES 64
int hs_engine_shard_ipc_export(hs_engine *h, void *handles_out /* 2 x HS_IPC_HANDLE_BYTES */);
int hs_engine_shard_ipc_attach(hs_engine *h, const void *all_handles /* world x 2 x HS_IPC_HANDLE_BYTES, rank order */);
/* ... ranks that live in ONE process (virtual shards on one device: tests, tools) hand each other the buffers' addresses instead:
 * ipc_buffers returns this engine's two buffers (after ipc_export), peers_local takes the [world] addresses of every rank's. */
int hs_engine_shard_ipc_buffers(hs_engine *h, int64_t **inbox_out, int64_t **bounds_out);
int hs_engine_shard_peers_local(hs_engine *h, int64_t *const *inbox_ptrs, int64_t *const *bounds_ptrs);
int hs_engine_shard_push(hs_engine *h);
int hs_engine_shard_inject_ipc(hs_engine *h);
/* Simulation.__init__ bootstrap (core/simulation.py:145-154): clock to start_ns, every Source draws its
 * first arrival.  Called implicitly by the first run; call again to rewind the engine for another run.
 * (On a uniform grid small enough for one wavefront per LP -- csrc/hs_kernels_wave.hpp -- the bootstrap is performed by the next
 * run's kernel itself instead of a launch of its own; any getter or other run path in between performs it first: no observable
 * difference, one launch and one round trip of the state through HBM less.) */
int hs_engine_reset(hs_engine *h);
/* == Simulation._execute_until(end_ns): process events until the last processed event's time exceeds
 * end_ns (one-event overshoot included).  Re-entrant (windows, core/simulation.py:527-541 `_run_window`): station engines continue
 * from the state the call before left; NETWORK engines (hs_engine_set_network) hold no mid-run state between launches and REPEAT
 * the run from start_ns to the new end_ns -- the reference processes events in one global order whatever the window ends are, so
 * the state after windows e_1 <= ... <= e_k is the state of one run to e_k (tests/test_gpu_ring.py) -- at the cost of the whole
 * prefix per window.  An end_ns at or before the previous one moves nothing (the reference's loop condition is already false).
 * Blocks until the device work is complete. */
int hs_engine_run_until(hs_engine *h, int64_t end_ns);
/* Same, but only enqueues the work on the engine's stream.  The results are FINAL behind hs_engine_synchronize: that is where a run
 * that skipped the prologue (hs_engine_prologue_path) or tandem passes that met an undecided tie (hs_engine_tandem_path) are
 * repeated on the single heap.  Every getter below (get_summary, get_lp_stats, get_net_stats, read_sink(s), read_probe(_slot),
 * read_source_generated) finalises a pending asynchronous run first, so results read without an explicit synchronize are final too. */
int hs_engine_run_until_async(hs_engine *h, int64_t end_ns);
int hs_engine_synchronize(hs_engine *h);
/* hs_engine_reset + hs_engine_run_until_async, `repeats` times back to back on the engine stream, timing
 * each run kernel with HIP events recorded on that stream; ms_out[repeats] receives per-run kernel times.
 * Used by bench.py (the whole call is also wall-clocked by the caller). */
int hs_engine_bench_runs(hs_engine *h, int64_t end_ns, int32_t repeats, float *kernel_ms_out, float *total_ms_out);

int hs_engine_get_summary(hs_engine *h, hs_summary *out);
int hs_engine_get_lp_stats(hs_engine *h, const hs_lp_stats *out);
/* Sink records of one LP in processing order: completion time and context["created_at"]
 * (components/common.py:36-44).  Returns the number of records copied (<= cap) or a negative hs_status. */
int64_t hs_engine_read_sink(hs_engine *h, int32_t lp, int64_t *t_ns, int64_t *created_ns, int64_t cap);
/* All LPs at once: counts[n_lp] receives per-LP record counts; t_ns/created_ns receive the records
 * concatenated in LP order (caller sizes them from a previous get_lp_stats / sink_records). */
int64_t hs_engine_read_sinks(hs_engine *h, int64_t *counts, int64_t *t_ns, int64_t *created_ns, int64_t cap_total);

/* =====================================================================================================
 * Load-balancer topologies (BASELINE configs[4]; reference: components/load_balancer/load_balancer.py:347-433,
 * strategies.py:336-433, wiring of examples/visual/chash_example.py:118-140):
 *
 *     S x Source  ->  LoadBalancer(strategy=ConsistentHash(virtual_nodes))  ->  B x Server  ->  Sink(s)
 *
 * All hops between a Source's tick and the backend's queue take zero simulated time, so there is no lookahead for
 * conservative windows; but the graph is feed-forward, so one `_execute_until(end)` is a PIPELINE of device passes:
 *   1. every Source LP (one lane each) generates its ticks up to end_ns, draws the client id of each Request
 *      (metadata["client_id"] = str(int(u * n_clients)), u from the source's KEY stream -- the Philox-plugged form of
 *      chash_example.py:69-88) and looks up the backend ConsistentHash.select picks for it;
 *   2. the Requests are radix-sorted by (backend, arrival ns) -- this is the part of the reference's global heap
 *      order (core/event_heap.py:54-108) that matters to a backend;
 *   3. every backend LP (one lane each) runs the Queue/Driver/Worker protocol over its arrival list;
 *   4. completions of a Sink shared by all backends are radix-sorted by completion ns (the Sink's processing order);
 *   5. the first event beyond end_ns is elected among all LPs (one-event overshoot, core/simulation.py:472).
 * The md5 ring is built on the host once, at creation (ConsistentHash.add_backend runs in LoadBalancer.__init__).
 * One run per hs_lb_run call (the call is not re-entrant: it restarts from start_ns). */
typedef struct hs_lb_config {
    uint32_t struct_size;
    int32_t device;
    int32_t n_sources;
    int32_t n_backends;
    int64_t start_ns;
    int64_t horizon_ns;        /* latest end_ns that will be passed to hs_lb_run: sizes every buffer */
    uint64_t seed;             /* Philox key of the run */
    int32_t virtual_nodes;     /* ConsistentHash(virtual_nodes=...), >= 1 */
    int32_t shared_sink;       /* 1: every backend's downstream is ONE Sink; 0: one Sink per backend (or none, see egress) */
    int64_t tick_capacity;     /* ticks per source in the arrival log; 0 = derive from rate * horizon */
    int32_t strategy;          /* hs_lb_strategy: how LoadBalancer._forward_request picks the backend (load_balancer.py:368) */
    int32_t reserved;
} hs_lb_config;
/* LoadBalancingStrategy.select (components/load_balancer/strategies.py), all backends healthy:
 *   CONSISTENT_HASH  strategies.py:336-433: the md5 ring over `virtual_nodes` points per backend, key = metadata["client_id"];
 *   ROUND_ROBIN      strategies.py:50-73, the LoadBalancer's DEFAULT: backends[_index % len(backends)], _index += 1 per select --
 *                    i.e. the k-th Request the LoadBalancer processes (global (time, _sort_index) order over all Sources) goes
 *                    to backend k mod B.  The engine ranks all Requests by arrival first (one more device sort), then proceeds
 *                    as for any other assignment;  hs_lb_sources.n_clients and virtual_nodes are ignored;
 *   RANDOM           strategies.py:137-150: random.choice(backends) with the choice plugged like every draw of the seed-matched
 *                    definition: backends[int(u * len(backends))], u = the Request's draw from its Source's
 *                    KEY stream (the stream the client ids come from otherwise);  n_clients and virtual_nodes are ignored. */
typedef enum hs_lb_strategy { HS_LB_CONSISTENT_HASH = 0, HS_LB_ROUND_ROBIN = 1, HS_LB_RANDOM = 2 } hs_lb_strategy;

typedef struct hs_lb_sources {     /* [n_sources] each; NULL = documented default */
    const uint8_t *src_kind;           /* hs_source_kind (POISSON / CONSTANT); NULL = POISSON */
    const double *src_rate;            /* required */
    const int64_t *src_stop_after_ns;  /* < 0 = never; the provider returns no Request when time > stop_after */
    const int64_t *n_clients;          /* client ids are drawn uniformly from [0, n_clients); required, >= 1 */
    const uint64_t *stream_base;       /* NULL = i */
    /* Source.with_profile(LinearRampProfile | SpikeProfile) in front of the LoadBalancer: as hs_stations.src_profile_kind /
     * src_profile_params (src_rate = the PEAK rate: it sizes the tick log).  NULL = constant rates. */
    const uint8_t *src_profile_kind;   /* hs_profile_kind */
    const double *src_profile_params;  /* [n_sources][4] */
} hs_lb_sources;

typedef struct hs_lb_backends {    /* [n_backends] each */
    const int32_t *concurrency;        /* 1..32; NULL = 1 */
    const uint8_t *svc_kind;           /* hs_latency_kind; NULL = HS_LAT_CONSTANT */
    const double *svc_mean_s;          /* NULL = 0.01 */
    const int64_t *queue_cap;          /* < 0 = unbounded */
    const uint8_t *egress;             /* HS_EGRESS_NONE / HS_EGRESS_SINK; NULL = SINK */
    const uint64_t *stream_base;       /* NULL = n_sources + j */
    const char *names;                 /* concatenated backend names (the ring hashes "<name>:<i>"); required */
    const int32_t *name_off;           /* [n_backends + 1] offsets into names */
} hs_lb_backends;

typedef struct hs_lb_stats {       /* any pointer may be NULL */
    int64_t *generated;        /* [n_sources]  Source._generated_count */
    int64_t *lb;               /* [5] LoadBalancer.stats: requests_received, requests_forwarded, requests_failed,
                                      no_backend_available, len(_in_flight) */
    int64_t *total_requests;   /* [n_backends] BackendInfo.total_requests (load_balancer.py:385-386) */
    int64_t *accepted, *dropped, *completed, *rejected;   /* [n_backends] as hs_lp_stats */
    double *total_service_s;   /* [n_backends] */
    int64_t *queue_depth;      /* [n_backends] */
    int32_t *active;           /* [n_backends] */
    int64_t *sink_received;    /* [n_backends] Request@Sink events caused by this backend's completions */
} hs_lb_stats;

typedef struct hs_lb hs_lb;

int hs_lb_create(const hs_lb_config *cfg, const hs_lb_sources *src, const hs_lb_backends *be, hs_lb **out);
/* Simulation.__init__ + run() to end_ns.  Blocks until the device work is complete.  Every call runs from start_ns: windows
 * e_1 <= ... <= e_k over a load-balancer graph (`_run_window`, core/simulation.py:527-541) are k calls, and the state after the
 * last one is the state of one run to e_k -- which is what the reference's windows leave (one global event order, every call stops
 * behind the first event beyond its end): tests/test_gpu_lb.py::test_lb_windows_equal_one_run. */
int hs_lb_run(hs_lb *h, int64_t end_ns);
/* `repeats` complete runs back to back on the engine's stream; per-run device time of the whole pipeline and of the
 * sort passes alone (HIP events on that stream). */
int hs_lb_bench_runs(hs_lb *h, int64_t end_ns, int32_t repeats, float *run_ms_out, float *sort_ms_out);
int hs_lb_get_summary(hs_lb *h, hs_summary *out);
int hs_lb_get_stats(hs_lb *h, const hs_lb_stats *out);
/* Sink records in the Sink's processing order.  shared_sink: sink must be 0 (all completions, global order);
 * otherwise sink = backend index.  Returns the number of records copied or a negative hs_status. */
int64_t hs_lb_read_sink(hs_lb *h, int32_t sink, int64_t *t_ns, int64_t *created_ns, int64_t cap);
/* Probe.on(<backend Server> | <Sink>, metric, interval) on a load-balancer graph (instrumentation/probe.py:81-164), set once
 * after hs_lb_create and before the first run: target_kind 0 = backend Server `target_index` (depth, active_requests,
 * stats_accepted, stats_dropped, requests_completed), 1 = Sink `target_index` (events_received; the shared Sink is index 0),
 * 2 = Source `target_index` (generated_count; not for a Source with stop_after).
 * Tick times follow the reference's ConstantArrivalTimeProvider over _ProbeProfile; each tick at or before end_ns is two
 * reference events (kinds 13, 14) and the pending tick takes part in the election of the one event beyond end_ns.  The samples
 * are read off the run's logs; a sample on the very nanosecond of an event of its target makes hs_lb_run fail with
 * HS_E_UNSUPPORTED (that order is the reference's sort-index ledger, not lowered for this graph) -- never a guess. */
int hs_lb_set_probes(hs_lb *h, int32_t n_probes, const int32_t *target_kind, const int32_t *target_index,
                     const uint8_t *metric, const double *interval_s);
/* samples of probe `probe` in the last run: times and values -> returns their number (<= cap) */
int64_t hs_lb_read_probe(hs_lb *h, int32_t probe, int64_t *t_ns, int64_t *values, int64_t cap);
/* Sink.latency_stats() of the shared Sink, computed on the device (components/common.py:59-76 with
 * instrumentation/data.py:197-210): out = {count, avg, min, max, p50, p99} in seconds.  avg = sum(sorted latencies) / n with
 * the sum taken in index order in binary64 the way hs_set_float_sum_mode says. */
int hs_lb_latency_stats(hs_lb *h, double out[6]);
/* What `sum(list_of_floats)` means on the interpreter that runs the reference (CPython Python/bltinmodule.c builtin_sum):
 * 0 = plain left-to-right additions (CPython < 3.12; the library's default), 1 = Neumaier's compensated sum (CPython >= 3.12 --
 * the reference requires >= 3.13, pyproject.toml:11).  The host side sets it from sys.version_info, so that the device
 * statistics (hs_lb_latency_stats, hs_sink_latency_stats) equal what the same interpreter computes from the list
 * (components/common.py:59-76 `sum(sorted_vals) / n`).  Process-wide. */
int hs_set_float_sum_mode(int compensated);
/* The sorted ring's backend index per point, [n_backends * virtual_nodes] (strategies.py:381-391). */
int hs_lb_ring(hs_lb *h, int32_t *ring_backend);
/* ConsistentHash.select for a key string (strategies.py:412-433): backend index. */
int32_t hs_lb_select(hs_lb *h, const char *key);
const char *hs_lb_last_error(const hs_lb *h);
void hs_lb_destroy(hs_lb *h);
/* md5 digest of a byte string (the ring's hash function), exported so tests can check it against RFC 1321. */
void hs_md5(const char *msg, int64_t len, uint8_t out[16]);

/* Debug: bit 0 routes every timestamp group of every backend through the general in-group FIFO path; bit 1 disables the
 * request-order loop of single-worker unbounded backends (event-order loop with its single-event fast path instead);
 * bit 2 keeps the backend streams in dense per-backend segments instead of the wave-coalesced [k][backend] layout;
 * 8: the Sources draw their own stream values; 16: look-back radix passes; 32: 32 LPs per wavefront; 64: round 3's backend pipeline
 * instead of the segmented scan; 128: the int64 instantiation of the scan; 256: no speculated whole-ns arrival steps;
 * 512: hs_lbk_sources instead of hs_lbk_sources_lean; 1024: the lean Source kernel gives up, so that the run repeats with
 * hs_lbk_sources.  Every combination gives bit-identical results (tests/test_gpu_lb.py). */
int hs_debug_lb_flags(hs_lb *h, int flags);

/* Debug / tests: stable LSD radix sort of n (key, value) pairs on `device` over key bits [0, key_bits) --
 * the sort the load-balancer engine uses (csrc/hs_radix.hpp).  Host arrays in, host arrays out. */
int hs_debug_radix_sort(int32_t device, int64_t n, int32_t key_bits, const uint64_t *keys_in, const uint64_t *vals_in,
                        uint64_t *keys_out, uint64_t *vals_out, float *device_ms);
/* Merge of per-station Sink logs into ONE Sink shared by several stations (components/common.py:36-44 appends in global
 * processing order): stable device sort of n (completion ns, created_at ns) records by completion time, in place on host
 * arrays.  Feed it the stations' records concatenated in station order. */
int hs_merge_sink_records(int32_t device, int64_t n, int64_t *t_ns, int64_t *created_ns);

/* Sink.latency_stats() (components/common.py:59-76; percentiles as instrumentation/data.py:197-210) of a Sink's records on
 * the device: latency = t - created_at in ns -> seconds, radix sort, binary64 sum of the SORTED values in index order (what
 * `sum(sorted_vals)` does: hs_set_float_sum_mode), interpolated percentiles.  out = {count, avg, min, max, p50, p99}.  Host buffers. */
int hs_sink_latency_stats(int32_t device, int64_t n, const int64_t *t_ns, const int64_t *created_ns, double out[6]);

/* Samples of the LP's Probe in sampling order: (sample time ns, value) -- what the reference appends to the probe's
 * Data container (instrumentation/probe.py:63).  Returns the number copied or a negative hs_status. */
int64_t hs_engine_read_probe(hs_engine *h, int32_t lp, int64_t *t_ns, int64_t *values, int64_t cap);
/* ... of the probe in `slot` (0 .. 3) of the LP */
int64_t hs_engine_read_probe_slot(hs_engine *h, int32_t lp, int32_t slot, int64_t *t_ns, int64_t *values, int64_t cap);
/* Source.generated_count of the Sources in `slot` (1 .. 3: hs_stations.src_more_*; slot 0 = hs_lp_stats.generated) of
 * every LP -> out[n_lp]. */
int hs_engine_read_source_generated(hs_engine *h, int32_t slot, int64_t *out);

const char *hs_last_error(const hs_engine *h);
const char *hs_last_global_error(void);
void hs_engine_destroy(hs_engine *h);

/* Device-side scalar semantics, exported so tests can compare them bit-for-bit with the oracle:
 * fills u[i] = uniform(seed, sid, k0+i), e[i] = -hs_log(1-u[i]), ns[i] = trunc((e[i]/rate)*1e9). */
/* Debug: bit 0 routes every timestamp group through the general in-group FIFO path (tests compare it with
 * the fast path); bit 4 (16) runs a station network on the windowed engine (one launch per window) even when the
 * asynchronous whole-run engine is available; bit 6 (64) disables the asynchronous engine's in-wavefront scan of the
 * per-link bounds; bit 7 (128) makes its idle wavefronts back off (s_sleep) between polls. */
int hs_debug_set_flags(hs_engine *h, int flags);
/* Debug: telemetry of the last asynchronous network run: {sum over wavefronts of loop iterations, max over wavefronts,
 * timestamp groups run (sum over LPs), wavefronts}. */
int hs_debug_async_counters(hs_engine *h, unsigned long long out[4]);

int hs_debug_draws(int32_t device, uint64_t seed, uint64_t sid, uint64_t k0, int64_t n, double rate,
                   double *u, double *e, int64_t *ns);

/* Evaluation budget of the tick-table kernel (csrc/hs_tables.hpp): adaptive-Simpson intervals ONE lane may visit for ONE
 * arrival of a time-varying Source (64 lanes share an integral).  Replaces nothing in the reference -- its integrator
 * (numerics/integration.py:11-90) has no limit and needs minutes for such arrivals; this is the device's time guard, a
 * RUN-TIME argument (round 2: a build-time constant).  Default 2^24; takes effect at the next reset / run. */
int hs_engine_set_profile_budget(hs_engine *h, int64_t intervals_per_lane);
int hs_lb_set_profile_budget(hs_lb *h, int64_t intervals_per_lane);

/* Debug / test hook: the tick table of one stream -- next_arrival_time's general path (load/arrival_time_provider.py:84-144)
 * iterated from start_ns until two ticks lie beyond horizon_ns.  profile = {kind (1 ramp, 2 spike, 3 constant rate through the
 * general path), p0..p3}; lone = 0: the cooperative kernel, 1: one lane with the sequential integrator.  Returns the number of
 * entries (<= cap) or a negative error; out_status[0] != 0: over the evaluation budget, out_status[1] != 0: cap too small. */
int64_t hs_debug_tick_table(int32_t device, const double *profile, int32_t poisson, uint64_t seed, uint64_t sid,
                            int64_t start_ns, int64_t horizon_ns, int64_t cap, int64_t lane_budget, int32_t lone,
                            int64_t *out_times, uint64_t *out_status);

/* Debug: the engine divides by per-LP constants (1e9, rate, lambda) with a multiply + FMA sequence that must be
 * bit-identical to the IEEE quotient.  q_fast[i] = that sequence for a[i] / b, q_ieee[i] = the hardware
 * division, q_ns[i] = seconds_from_ns((int64)a[i]) (compare with a[i] / 1e9). */
int hs_debug_const_div(int32_t device, double b, int64_t n, const double *a, double *q_fast, double *q_ieee,
                       double *q_ns);

/* ------------------------------------------------------------------------------------------------------------------
 * General entity graphs (ABI 15; LoadBalancer nodes: ABI 16).
 *
 * The station engines take graphs of the shape [Sources] -> Server -> {Sink | NetworkLink | RandomRouter} with one sender
 * per link, <= 4 Sources per Server, <= 4 router targets ... (hs_engine_set_stations / hs_engine_set_network).  Everything
 * else the same entity classes can be wired into -- a NetworkLink with several senders (components/network/link.py:114-189),
 * a RandomRouter with any number of targets, among them Servers and other routers, and with several upstreams
 * (components/random_router.py:32-45), Server(downstream=<Server>) next to links (components/server/server.py:64-122,271-272),
 * any number of Sources per Server (load/source.py:93-180), any concurrency (server/concurrency.py:67-141) -- runs here:
 * ONE heap ordered by (time, _sort_index), popped event by event like `Simulation._execute_until`
 * (core/simulation.py:449-505), every reference Event materialised with the sort index the reference gives it (the two
 * counters of core/event.py:53-77 and core/event_heap.py:48), push / pop in CPython's heapq sift order.  One lane of one
 * wavefront walks the loop; the first 4 096 heap entries live in LDS (128 KB of the CU's 160), the rest and all node state
 * in HBM.  An exactness path for small models (~2.4 us per event and heap; heaps side by side: hs_graph_run_many / _parts), not a throughput path: the station engines stay the
 * product's hot path, and the host (happy_simulator_amd/graph_engine.py) only comes here with a graph they refuse.
 * Replaces: Simulation.__init__'s bootstrap (core/simulation.py:145-160), Simulation.schedule (:195-206), _execute_until
 * (:449-505) and the handlers of load/source.py:142-180, components/queue.py:122-166, components/queue_driver.py:66-99,
 * components/server/server.py:202-273, components/common.py:36-44, components/random_router.py:32-45,
 * components/network/link.py:114-216 for such a graph.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct hs_graph hs_graph;

typedef enum hs_node_kind { HS_NODE_SOURCE = 0, HS_NODE_SERVER = 1, HS_NODE_SINK = 2, HS_NODE_LINK = 3, HS_NODE_ROUTER = 4, HS_NODE_PROBE = 5, HS_NODE_LB = 6 } hs_node_kind;

typedef struct hs_graph_config {
    uint32_t struct_size;            /* sizeof(hs_graph_config) */
    int32_t device;
    int64_t start_ns;                /* Simulation(start_time=) */
    uint64_t seed;                   /* Philox key of every entity stream */
    int64_t heap_capacity;           /* pending events; 0 = sized from the graph.  All three capacities GROW on demand: */
    int64_t request_capacity;        /* Requests alive (queued, in service, in transit); 0 = sized from the graph */
    int64_t record_capacity;         /* Sink records of the whole run; 0 = 65 536 */
    int64_t max_events;              /* a run that would process more is refused (HS_E_UNSUPPORTED); 0 = no limit */
    int64_t profile_budget;          /* adaptive-Simpson intervals one lane may visit for one tick of a time-varying Source or a
                                      * Probe (hs_engine_set_profile_budget); 0 = the default 2^24 */
} hs_graph_config;

/* Nodes in the caller's order.  SOURCE nodes must come in `sources=[...]` order (their first SourceEvents take the pre-run
 * sort indices 0, 1, ... in that order, core/simulation.py:145-154).  Entity streams: sid = stream_base << 3 | kind
 * (DESIGN.md section 3): a Source draws ARRIVAL, a Server SERVICE, a link LINK (jitter) and LOSS, a router ROUTE. */
typedef struct hs_graph_nodes {
    int32_t n_nodes;
    const uint8_t *kind;             /* [n] hs_node_kind */
    const int32_t *target;           /* [n] Source: the node its Requests are aimed at; Server: downstream; link: egress; -1 = none */
    const uint64_t *stream_base;     /* [n] */
    const uint8_t *src_kind;         /* [n] Sources: HS_SRC_POISSON / HS_SRC_CONSTANT (constant-rate profile) */
    const double *src_rate;          /* [n] Sources: events / s, > 0 */
    const int64_t *src_stop_after_ns;/* [n] Sources: SimpleEventProvider(stop_after); < 0 = never */
    const int32_t *concurrency;      /* [n] Servers: FixedConcurrency(max_concurrent) >= 1 */
    const uint8_t *lat_kind;         /* [n] Servers: service distribution; links: jitter (HS_LAT_EXPONENTIAL / HS_LAT_CONSTANT) */
    const double *lat_mean_s;        /* [n] ... its mean (a link with HS_LAT_CONSTANT and mean 0: jitter=None) */
    const double *link_lat_min_s;    /* [n] links: ConstantLatency base latency, >= 0 */
    const double *link_loss_rate;    /* [n] links: packet_loss_rate in [0, 1]; NULL = lossless */
    const int64_t *queue_cap;        /* [n] Servers: FIFOQueue capacity; < 0 = unbounded */
    const int32_t *rt_off;           /* [n] routers: targets rt_targets[rt_off .. rt_off + rt_cnt) in constructor order */
    const int32_t *rt_cnt;           /* [n] */
    const int32_t *rt_targets;       /* [n_rt] node ids (Sink / link / Server / router) */
    int32_t n_rt;
    /* Sources with a time-varying profile (load/profile.py:52-113; Source.with_profile): hs_profile_kind and its four parameters as
     * in hs_stations.src_profile_*; src_rate is then the peak rate.  Their ticks -- next_arrival_time's general path,
     * load/arrival_time_provider.py:84-144 -- come from the tick-table kernel (csrc/hs_tables.hpp), like the station engines'.  NULL = none. */
    const uint8_t *src_profile_kind; /* [n] */
    const double *src_profile_params;/* [n][4] */
    /* HS_NODE_PROBE: Probe(target, metric, interval) (instrumentation/probe.py:81-164) -- a Source whose provider is a constant
     * tick chain over _ProbeProfile(interval) (the general path too) and whose payload is the daemon probe_event that samples
     * getattr(target, metric).  `target` = the sampled node; PROBE nodes come behind the SOURCE nodes in `probes=[...]` order (their
     * first ticks take the pre-run indices behind the Sources', core/simulation.py:156-160).  Samples: hs_graph_read_records
     * (node = the probe, t = sample time, created = the value). */
    const uint8_t *probe_metric;     /* [n] hs_probe_metric */
    const double *probe_interval_s;  /* [n] > 0 */
    /* HS_NODE_LB: LoadBalancer(backends=[...], strategy) (components/load_balancer/load_balancer.py:347-473) -- any number of them,
     * anywhere a Request can go.  Backends = rt_targets[rt_off .. rt_off + rt_cnt) in add_backend order, SERVER nodes, all healthy.
     * ConsistentHash (strategies.py:336-433): lb_vnodes ring points md5("<backend name>:<i>") per backend (names / name_off), the key
     * of a Request is str(client_id); RoundRobin (strategies.py:50-73); Random (strategies.py:137-150) with random.choice plugged by
     * the Request's key draw: backends[client_id].  A Source with src_n_clients > 0 builds its Requests like a
     * ClientKeyEventProvider (examples/visual/chash_example.py:69-88): client_id = int(u * n_clients), u from its KEY stream.  A
     * Request WITHOUT a client_id (a scheduled one, one of a plain Source) at a ConsistentHash LoadBalancer takes the strategy's own
     * fallback RoundRobin, which only such Requests advance (strategies.py:362,420-421); it must not reach a Random LoadBalancer
     * (the reference draws from the process-wide generator): the caller refuses such graphs; the device sends it to backend 0.  Every forwarded Request carries the `_lb_response` completion hook
     * (load_balancer.py:413-431): one more event behind the backend's enqueue.  NULL = no LoadBalancer. */
    const uint8_t *lb_strategy;      /* [n] hs_lb_strategy */
    const int32_t *lb_vnodes;        /* [n] ConsistentHash(virtual_nodes) */
    const char *names;               /* concatenated entity names (only an LB's backends need one) */
    const int32_t *name_off;         /* [n + 1] */
    const int64_t *src_n_clients;    /* [n] Sources; 0 = a plain SimpleEventProvider */
} hs_graph_nodes;

typedef struct hs_graph_stats {      /* host arrays [n_nodes] (rt_taken: [n_rt]); any pointer may be NULL */
    int64_t *generated;              /* Source._generated_count                         load/source.py:159 */
    int64_t *payloads;               /* Requests the Source's event provider built      load/source.py:67-86 */
    int64_t *accepted, *dropped;     /* Queue.stats_accepted / stats_dropped            components/queue.py:127-139 */
    int64_t *completed, *rejected;   /* Server._requests_completed / _rejected          server/server.py:112-114 */
    double *total_service_s;         /* Server._total_service_time */
    int64_t *queue_depth, *active;   /* QueuedResource.depth, Server.active_requests */
    int64_t *received;               /* Sink.events_received */
    int64_t *entered;                /* Requests that entered the link */
    int64_t *packets_sent;           /* NetworkLink.packets_sent                        components/network/link.py:162 */
    int64_t *packets_dropped;        /* NetworkLink.packets_dropped                     components/network/link.py:132 */
    int64_t *routed;                 /* RandomRouter.stats_routed                       components/random_router.py:36 */
    int64_t *rt_taken;               /* [n_rt] how often each target slot was drawn     (target_counts, random_router.py:37);
                                      * a LoadBalancer's slots: BackendInfo.total_requests       load_balancer.py:385-386 */
    int64_t *lb;                     /* [n][6] LoadBalancer: requests_received, requests_forwarded, requests_failed,
                                      * no_backend_available, len(_in_flight)                    load_balancer.py:349-388
                                      * and the selections of the strategy's RoundRobin: RoundRobin._index, or the index of
                                      * ConsistentHash's key-less fallback                         strategies.py:66-67,362 */
} hs_graph_stats;

int hs_graph_create(const hs_graph_config *cfg, const hs_graph_nodes *nodes, hs_graph **out);
/* Simulation.schedule(Event(time, "Request", target=<node>)) (core/simulation.py:195-206): the Event was constructed outside
 * the run, so it takes the next index of the process-wide counter behind the Sources' first ticks; context["created_at"] =
 * its own time.  Calls in the order the caller constructed the Events.  `node`: a Server, Sink, link, router or
 * LoadBalancer. */
int hs_graph_schedule(hs_graph *g, int32_t node, int64_t time_ns);
/* `_execute_until(end)`: pops while the PREVIOUS event's time <= end_ns (so exactly one event beyond the end is processed,
 * core/simulation.py:472); may be called again with a later end (windows, :527-541). */
int hs_graph_run_until(hs_graph *g, int64_t end_ns);
/* ParallelRunner.run_sweep / run_replicas (parallel/runner.py:82-142: one worker process per independent Simulation) for graphs of
 * this path: `n` handles on one device run to `end_ns` SIDE BY SIDE -- one workgroup and one heap each, one launch for all of them
 * (up to 1 024 resident at once: four per CU with a 32 KB window of the heap in LDS), relaunched for the ones that have to grow a buffer.  Every handle ends in
 * exactly the state hs_graph_run_until(handle, end_ns) would leave; the device time a handle reports (hs_summary.last_run_ms) is the batch's.  On
 * an error the first handle's hs_graph_last_error names the graph. */
int hs_graph_run_many(hs_graph *const *graphs, int32_t n, int64_t end_ns);
/* ONE Simulation whose entity graph falls into parts that no Request can cross (connected components, or unions of them), every
 * handle one part with its nodes in the Simulation's order: the parts run side by side like hs_graph_run_many's replicas and
 * together leave what ONE heap over all of them leaves (core/simulation.py:449-505).  Why that is exact: an event's place among
 * its part's events never depends on the other parts -- except (i) inside a timestamp group that holds both an event numbered
 * BEFORE the run (a Source's / Probe's first tick, a schedule()d Request: the process-wide counter, core/event.py:53-67) and one
 * numbered BY the run (the heap's own counter, which counts every part's events), and (ii) for the single event beyond `end_ns`,
 * which is the earliest over all parts (the loop tests the previous event's time, core/simulation.py:472).  The parts stop in front
 * of their first event beyond `end_ns`; the one part that holds the earliest then processes it.  Returns HS_OK; or 1 = UNDECIDED
 * -- case (i) occurred, or two parts hold the earliest time: the caller repeats the run on one heap (the handles' state is then
 * meaningless); or a negative hs_status.  Call once per handle set (no later end). */
int hs_graph_run_parts(hs_graph *const *parts, int32_t n, int64_t end_ns);
int hs_graph_get_summary(hs_graph *g, hs_summary *out);
int hs_graph_get_stats(hs_graph *g, hs_graph_stats *out);
/* Every Sink record of the run in processing order: (Sink node, completion ns, created_at ns).  Returns the number of
 * records (copies min(that, cap)) or a negative hs_status. */
int64_t hs_graph_read_records(hs_graph *g, int32_t *node, int64_t *t_ns, int64_t *created_ns, int64_t cap);
const char *hs_graph_last_error(const hs_graph *g);
void hs_graph_destroy(hs_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* HS_ENGINE_H */
