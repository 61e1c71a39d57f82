/* hs_oracle.c -- event-level CPU restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see hs_oracle.h).  Plain C11, no dependencies.
 *
 * Each handler cites the reference code it restates (paths relative to
 * /root/reference/happysimulator).  Nothing here is shared with the product.
 */
#include "hs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hs_rng_ref.h"

/* ---------------------------------------------------------------- events */
typedef struct {
    int64_t time;   /* Event.time (ns)                      core/event.py:149 */
    uint64_t idx;   /* Event._sort_index                    core/event.py:165 */
    int32_t kind;   /* HSO_EV_*                                               */
    int32_t node;   /* target node                                            */
    int32_t req;    /* request slot (payload / context), -1 if none           */
    int32_t aux;    /* per-kind scratch (service slot ...)                    */
#ifdef HSO_LINEAGE  /* analysis build only (tools/election_rules.py): when was it created, and its creator, and that one's */
    int64_t crt, crt2, crt3;
    /* ... and its place in the FIFO that the heap is among events of one nanosecond: how many steps after the root of the group
     * it was created in (cdepth), when that root was created and how deep in ITS group (rcrt, rcdepth), and once more */
    int64_t cdepth, rcrt, rcdepth, r2crt, r2cdepth;
    int64_t rsrc;      /* the Source / Probe node whose tick is the most recent one in the event's ancestry (its own node for a tick); -1 none */
#endif
} hso_event;

/* request = the payload Event's identity + its context dict
 * (context["created_at"], load/source.py:76-79; forwarded unchanged by
 * Entity.forward, core/entity.py:100-105). */
typedef struct {
    int64_t created_ns;
    uint64_t idx;       /* sort index of the payload Event object (kept on retarget, queue_driver.py:86-90) */
    double service_s;   /* service_time_s captured by the generator frame (server/server.py:246-247) */
    int32_t hops;
    int32_t next_free;
    int64_t client_id;  /* context["metadata"]["client_id"] (as an int; the key string is its decimal form), -1 = none */
    int32_t lb_hook;    /* LB node whose `on_complete` hook rides on this Event (load_balancer.py:413-431), -1 = none */
} hso_request;

/* ConsistentHash ring point (strategies.py:336-433): md5 digest as a big-endian 128-bit integer + backend node */
typedef struct { uint64_t hi, lo; int32_t backend; int32_t seq; } hso_ring_pt;

typedef struct {
    int32_t *buf; int64_t head, len, cap;  /* FIFOQueue deque, components/queue_policy.py:75-114 */
} hso_fifo;

typedef struct {
    /* Source */
    int64_t arr_time_ns;      /* ArrivalTimeProvider.current_time */
    uint64_t arr_draws;
    int64_t generated;        /* Source._generated_count, load/source.py:159 */
    /* Server / QueuedResource */
    hso_fifo fifo;
    int64_t accepted, dropped;            /* Queue.stats_*, components/queue.py:127-139 */
    int32_t active;                       /* FixedConcurrency._active, server/concurrency.py:67-141 */
    int64_t completed, rejected;          /* Server._requests_*, server/server.py:112-114 */
    double total_service_s;               /* Server._total_service_time */
    uint64_t svc_draws;
    /* NetworkLink / RandomRouter */
    int64_t packets_sent;                 /* NetworkLink.packets_sent, components/network/link.py:162 */
    uint64_t link_draws;
    uint64_t loss_draws;                  /* packets that entered a lossy link */
    int64_t routed;                       /* RandomRouter.stats_routed, components/random_router.py:36 */
    uint64_t route_draws;
    /* LoadBalancer */
    int64_t lb_received, lb_forwarded, lb_failed, lb_no_backend, lb_in_flight, lb_next_id, lb_fallback_idx;
    int64_t *lb_total_requests;           /* [rt_cnt] BackendInfo.total_requests */
    hso_ring_pt *ring; int64_t ring_len;
    uint64_t key_draws;
    /* Sink */
    int64_t received;
    int64_t *sink_t, *sink_created; int64_t sink_cap;
} hso_node;

struct hso_sim {
    hso_graph g;   /* deep-copied arrays */
    hso_params p;
    hso_node *nodes;
    hso_event *heap; int64_t heap_len, heap_cap, heap_peak;
    hso_request *reqs; int32_t req_len, req_cap, req_free;
    uint64_t counter;          /* active sort counter */
    uint64_t global_counter;   /* the process-wide counter after Simulation.__init__ (core/event.py:53): Events a caller
                                  builds for Simulation.schedule() before run() take their index from it */
    int64_t current_ns;
    int64_t processed, by_kind[HSO_EV_KINDS];
    hsr_mt19937 mt_py, mt_np;
    hsr_mt19937 mt_coord;                 /* the coordinator's random.Random(seed) (hso_graph.ploss) */
    /* trace */
    int64_t *tr_t; int32_t *tr_kind; int32_t *tr_node; int64_t *tr_idx; int64_t tr_len;
#ifdef HSO_LINEAGE
    int cur_valid; int64_t cur_crt, cur_crt2;      /* the event being processed (creator of what is pushed now) */
    int64_t cur_rsrc;                              /* ... and the tick its children descend from (hso_event::rsrc) */
    int64_t g_depth, g_rcrt, g_rcdepth, g_r2crt, g_r2cdepth;   /* ... its depth in the current group and the group's root */
    hso_event *dump; int64_t dump_len;             /* everything pending when the one event beyond end_ns was popped, it first */
#endif
};

/* ------------------------------------------------------------------ heap
 * heapq with Event.__lt__ = (time, _sort_index), core/event.py:337-344,
 * core/event_heap.py:54-108.  push/pop follow CPython heapq's sift procedures
 * step for step (see heap_pop). */
static int ev_lt(const hso_event *a, const hso_event *b) {
    if (a->time != b->time) return a->time < b->time;
    return a->idx < b->idx;
}
static void heap_push(hso_sim *s, hso_event e) {
#ifdef HSO_LINEAGE
    e.crt = s->cur_valid ? s->current_ns : INT64_MIN;          /* INT64_MIN: constructed before run() */
    e.crt2 = s->cur_valid ? s->cur_crt : INT64_MIN;
    e.crt3 = s->cur_valid ? s->cur_crt2 : INT64_MIN;
    e.cdepth = s->cur_valid ? s->g_depth + 1 : 0;
    e.rcrt = s->cur_valid ? s->g_rcrt : INT64_MIN; e.rcdepth = s->cur_valid ? s->g_rcdepth : 0;
    e.r2crt = s->cur_valid ? s->g_r2crt : INT64_MIN; e.r2cdepth = s->cur_valid ? s->g_r2cdepth : 0;
    e.rsrc = (e.kind == HSO_EV_SOURCE || e.kind == HSO_EV_PROBE_TICK) ? (int64_t)e.node : s->cur_valid ? s->cur_rsrc : -1;
#endif
    if (s->heap_len == s->heap_cap) {
        s->heap_cap = s->heap_cap ? s->heap_cap * 2 : 1024;
        s->heap = (hso_event *)realloc(s->heap, (size_t)s->heap_cap * sizeof(hso_event));
    }
    int64_t i = s->heap_len++;
    while (i > 0) {
        int64_t par = (i - 1) >> 1;
        if (!ev_lt(&e, &s->heap[par])) break;
        s->heap[i] = s->heap[par];
        i = par;
    }
    s->heap[i] = e;
    if (s->heap_len > s->heap_peak) s->heap_peak = s->heap_len;
}
static hso_event heap_pop(hso_sim *s) {
    /* heapq.heappop: move the last leaf to the root, sift the hole DOWN to a leaf
     * always following the smaller child (right child on a tie: `not left < right`),
     * then sift the item back UP (_siftup then _siftdown in CPython's heapq).  The
     * exact procedure is mirrored so that equal (time, idx) keys -- possible between
     * init-time and run-time sort counters, SURVEY.md A1 -- pop in heapq's order. */
    hso_event top = s->heap[0];
    hso_event last = s->heap[--s->heap_len];
    int64_t n = s->heap_len;
    if (n == 0) return top;
    int64_t pos = 0, child = 1;
    while (child < n) {
        int64_t right = child + 1;
        if (right < n && !ev_lt(&s->heap[child], &s->heap[right])) child = right;
        s->heap[pos] = s->heap[child];
        pos = child;
        child = 2 * pos + 1;
    }
    while (pos > 0) {
        int64_t par = (pos - 1) >> 1;
        if (!ev_lt(&last, &s->heap[par])) break;
        s->heap[pos] = s->heap[par];
        pos = par;
    }
    s->heap[pos] = last;
    return top;
}

/* ----------------------------------------------------------------- utils */
static uint64_t next_index(hso_sim *s) { return s->counter++; } /* _next_sort_index, core/event.py:62-67 */

static int32_t req_alloc(hso_sim *s) {
    int32_t r;
    if (s->req_free >= 0) { r = s->req_free; s->req_free = s->reqs[r].next_free; return r; }
    if (s->req_len == s->req_cap) {
        s->req_cap = s->req_cap ? s->req_cap * 2 : 1024;
        s->reqs = (hso_request *)realloc(s->reqs, (size_t)s->req_cap * sizeof(hso_request));
    }
    return s->req_len++;
}
static void req_release(hso_sim *s, int32_t r) { s->reqs[r].next_free = s->req_free; s->req_free = r; }

static void fifo_push(hso_fifo *f, int32_t v) {
    if (f->len == f->cap) {
        int64_t ncap = f->cap ? f->cap * 2 : 16;
        int32_t *nb = (int32_t *)malloc((size_t)ncap * sizeof(int32_t));
        for (int64_t i = 0; i < f->len; ++i) nb[i] = f->buf[(f->head + i) % f->cap];
        free(f->buf);
        f->buf = nb; f->head = 0; f->cap = ncap;
    }
    f->buf[(f->head + f->len) % f->cap] = v;
    f->len++;
}
static int32_t fifo_pop(hso_fifo *f) {
    int32_t v = f->buf[f->head];
    f->head = (f->head + 1) % f->cap;
    f->len--;
    return v;
}

static double draw_uniform(hso_sim *s, int32_t node, int stream_kind, uint64_t *draws) {
    if (s->p.rng_mode == HSO_RNG_MT19937) {
        /* arrivals: np.random.random() (global); service: random.random() (global) -- SURVEY A5 */
        (*draws)++;
        return stream_kind == HS_STREAM_ARRIVAL ? hsr_mt_res53(&s->mt_np) : hsr_mt_res53(&s->mt_py);
    }
    uint64_t sid = (s->g.stream_base[node] << 3) | (uint64_t)stream_kind;
    return hsr_uniform(s->p.seed, sid, (*draws)++);
}

static double exp1(hso_sim *s, double u) {
    /* stock streams use libm log (math.log); the Philox-plugged streams use hs_log */
    if (s->p.rng_mode == HSO_RNG_MT19937) return -log(1.0 - u);
    return hsr_exp1(u);
}

/* ---- time-varying profiles: the general path of ArrivalTimeProvider.next_arrival_time ------------------------------
 * (load/arrival_time_provider.py:84-144) with numerics/integration.py:11-90 and numerics/root_finding.py:27-152. */
typedef struct { int32_t kind; double p[4]; } hso_profile;

/* rate_fn(t) = profile.get_rate(Instant.from_seconds(t))   (arrival_time_provider.py:85-86, load/profile.py:52-113) */
static double prof_rate(const hso_profile *pf, double t_seconds) {
    double t = hsr_seconds_from_ns(hsr_ns_from_seconds(t_seconds));
    if (pf->kind == HSO_PROF_LINEAR_RAMP) {
        double duration = pf->p[0], start = pf->p[1], end = pf->p[2];
        if (t <= 0) return start;
        if (t >= duration) return end;
        double fraction = t / duration;
        return start + fraction * (end - start);
    }
    if (pf->kind == HSO_PROF_SPIKE) {
        double baseline = pf->p[0], spike = pf->p[1], warmup = pf->p[2], dur = pf->p[3];
        if (t < warmup) return baseline;
        if (t < warmup + dur) return spike;
        return baseline;
    }
    return pf->p[0];                                                 /* _ProbeProfile.get_rate: self.rate */
}
static double simpson3(double fa, double fm, double fb, double h) { return h / 3.0 * (fa + 4.0 * fm + fb); }
static double simpson_adaptive(const hso_profile *pf, double a, double b, double fa, double fb, double s_whole, int depth,
                               double tol) {
    double m = (a + b) / 2.0;
    double h = (b - a) / 2.0;
    double fm = prof_rate(pf, m);
    double lm = (a + m) / 2.0;
    double rm = (m + b) / 2.0;
    double flm = prof_rate(pf, lm);
    double frm = prof_rate(pf, rm);
    double s_left = simpson3(fa, flm, fm, h / 2.0);
    double s_right = simpson3(fm, frm, fb, h / 2.0);
    double s_combined = s_left + s_right;
    double error_estimate = (s_combined - s_whole) / 15.0;
    if (depth >= 50 || fabs(error_estimate) < tol) return s_combined + error_estimate;   /* Richardson extrapolation */
    double left = simpson_adaptive(pf, a, m, fa, fm, s_left, depth + 1, tol / 2.0);
    double right = simpson_adaptive(pf, m, b, fm, fb, s_right, depth + 1, tol / 2.0);
    return left + right;
}
static double integrate_simpson(const hso_profile *pf, double a, double b, double tol) {
    if (a == b) return 0.0;
    if (a > b) return -integrate_simpson(pf, b, a, tol);
    double fa = prof_rate(pf, a), fb = prof_rate(pf, b);
    double m = (a + b) / 2.0;
    double fm = prof_rate(pf, m);
    double h = (b - a) / 2.0;
    double s_whole = simpson3(fa, fm, fb, h);
    return simpson_adaptive(pf, a, b, fa, fb, s_whole, 0, tol);
}
typedef struct { const hso_profile *pf; double t_start, target; } hso_objective;
static double objective(const hso_objective *o, double t) { return integrate_simpson(o->pf, o->t_start, t, 1e-10) - o->target; }
static double dmin2(double a, double b) { return b < a ? b : a; }   /* Python min(a, b) */
static double dmax2(double a, double b) { return b > a ? b : a; }   /* Python max(a, b) */

/* brentq(f, a, b): numerics/root_finding.py:27-152.  Returns 1 when converged. */
static int brentq(const hso_objective *o, double a, double b, double *root) {
    const double xtol = 1e-12, rtol = 4 * 2.220446049250313e-16;
    double fa = objective(o, a), fb = objective(o, b);
    if (fa * fb > 0) return 0;                                       /* ValueError in the reference */
    if (fabs(fa) < fabs(fb)) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
    double c = a, fc = fa, d = b - a, e = d;
    for (int iteration = 0; iteration < 100; ++iteration) {
        double tol = 2.0 * rtol * fabs(b) + xtol;
        double m = (c - b) / 2.0;
        if (fabs(m) <= tol || fb == 0) { *root = b; return 1; }
        if (fabs(e) >= tol && fabs(fa) > fabs(fb)) {
            double sr = fb / fa, p, q;
            if (a == c) { p = 2.0 * m * sr; q = 1.0 - sr; }
            else {
                q = fa / fc;
                double r = fb / fc;
                p = sr * (2.0 * m * q * (q - r) - (b - a) * (r - 1.0));
                q = (q - 1.0) * (r - 1.0) * (sr - 1.0);
            }
            if (p > 0) q = -q; else p = -p;
            if (2.0 * p < dmin2(3.0 * m * q - fabs(tol * q), fabs(e * q))) { e = d; d = p / q; }
            else { d = m; e = m; }
        } else { d = m; e = m; }
        a = b; fa = fb;
        if (fabs(d) > tol) b = b + d;
        else if (m > 0) b = b + tol;
        else b = b - tol;
        fb = objective(o, b);
        if (fb * fc > 0) { c = a; fc = fa; d = b - a; e = d; }
        else if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    }
    *root = b;
    return 0;
}
int64_t hso_profile_next_arrival(int32_t prof_kind, const double p[4], int64_t t_start_ns, double target_area) {
    hso_profile pf;
    pf.kind = prof_kind;
    memcpy(pf.p, p, sizeof pf.p);
    double t_start_sec = hsr_seconds_from_ns(t_start_ns);
    hso_objective o = {&pf, t_start_sec, target_area};
    double current_rate = prof_rate(&pf, t_start_sec), t_high;
    if (current_rate > 0) {
        double estimated_delay = (target_area / current_rate) * 2.0;          /* optimistic linear prediction */
        estimated_delay = dmax2(1e-9, dmin2(estimated_delay, 3600.0));
        t_high = t_start_sec + estimated_delay;
    } else t_high = t_start_sec + 0.1;
    double t_low = t_start_sec;
    int found = 0;
    for (int i = 0; i < 50; ++i) {                                             /* bracket search, geometric expansion */
        if (objective(&o, t_high) > 0) { found = 1; break; }
        double step = dmax2(1e-6, t_high - t_low);
        t_high += step * 2.0;
    }
    if (!found) return -1;                                                     /* RuntimeError in the reference */
    double root;
    if (!brentq(&o, t_low, t_high, &root)) return -1;
    return hsr_ns_from_seconds(root);                                          /* Instant.from_seconds(result.root) */
}

/* ArrivalTimeProvider.next_arrival_time (load/arrival_time_provider.py:57-144): constant-rate fast path
 * t' = from_seconds(to_seconds(t) + E/rate), otherwise the general path above. */
static int64_t next_arrival(hso_sim *s, int32_t n) {
    hso_node *nd = &s->nodes[n];
    double target_area;
    if (s->g.arr_kind[n] == HSO_ARR_POISSON)
        target_area = exp1(s, draw_uniform(s, n, HS_STREAM_ARRIVAL, &nd->arr_draws)); /* poisson_arrival.py:31 */
    else
        target_area = 1.0;                                                             /* constant_arrival.py:23 */
    if (s->g.prof_kind && s->g.prof_kind[n] != HSO_PROF_CONSTANT) {
        int64_t t = hso_profile_next_arrival(s->g.prof_kind[n], s->g.prof_p + 4 * (size_t)n, nd->arr_time_ns, target_area);
        nd->arr_time_ns = t < 0 ? INT64_MAX : t;                    /* the reference raises: the run would abort */
        return nd->arr_time_ns;
    }
    double t_start = hsr_seconds_from_ns(nd->arr_time_ns);
    double t_next = t_start + target_area / s->g.rate[n];
    nd->arr_time_ns = hsr_ns_from_seconds(t_next);
    return nd->arr_time_ns;
}

/* LatencyDistribution.get_latency(...).to_seconds() as used at server/server.py:246-247:
 * Duration.from_seconds(sample) (first truncation) then float(ns)/1e9. */
static double sample_latency_s(hso_sim *s, int32_t n, int stream_kind, uint64_t *draws) {
    double sample;
    if (s->g.lat_kind[n] == HSO_LAT_EXP) {
        double lambda = 1.0 / s->g.lat_mean[n];                      /* exponential.py:36 */
        sample = exp1(s, draw_uniform(s, n, stream_kind, draws)) / lambda; /* random.expovariate */
    } else {
        sample = s->g.lat_mean[n];                                   /* constant.py:33-35 */
    }
    return hsr_seconds_from_ns(hsr_ns_from_seconds(sample));
}

static void trace(hso_sim *s, const hso_event *e) {
    if (s->tr_len < s->p.trace_cap) {
        int64_t i = s->tr_len++;
        s->tr_t[i] = e->time; s->tr_kind[i] = e->kind; s->tr_node[i] = e->node; s->tr_idx[i] = (int64_t)e->idx;
    }
}

/* a Request-carrying Event aimed at `node`: which handler it lands in */
static int32_t arrival_kind_for(const hso_sim *s, int32_t node) {
    switch (s->g.kind[node]) {
        case HSO_SERVER: return HSO_EV_ENQUEUE;
        case HSO_SINK: return HSO_EV_SINK;
        case HSO_LINK: return HSO_EV_LINK;
        case HSO_ROUTER: return HSO_EV_ROUTE;
        case HSO_LB: return HSO_EV_LB;
        default: return -1;
    }
}

/* -------------------------------------------------------------- handlers */

/* Source.handle_event, load/source.py:142-180 + SimpleEventProvider.get_events :67-86 */
static void on_source(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    int64_t stop = s->g.stop_after_ns[n];
    int has_payload = !(stop >= 0 && e->time > stop);               /* :68 */
    hso_event payload;
    if (has_payload) {
        int32_t r = req_alloc(s);
        s->reqs[r].created_ns = e->time;                            /* context["created_at"] = time */
        s->reqs[r].hops = 0;
        s->reqs[r].service_s = 0.0;
        s->reqs[r].client_id = -1;
        s->reqs[r].lb_hook = -1;
        if (s->g.n_clients && s->g.n_clients[n] > 0) {              /* chash_example.py:83: one id per Request */
            double u = draw_uniform(s, n, HS_STREAM_KEY, &nd->key_draws);
            s->reqs[r].client_id = (int64_t)(u * (double)s->g.n_clients[n]);
        }
        payload.time = e->time;
        payload.idx = next_index(s);                                /* payload constructed first (:158) */
        s->reqs[r].idx = payload.idx;
        payload.node = s->g.target[n];
        payload.kind = arrival_kind_for(s, payload.node);
        payload.req = r; payload.aux = 0;
    }
    nd->generated++;                                                /* :159 */
    hso_event tick;
    tick.time = next_arrival(s, n);                                 /* :170 */
    tick.idx = next_index(s);                                       /* SourceEvent constructed second (:171) */
    tick.kind = HSO_EV_SOURCE; tick.node = n; tick.req = -1; tick.aux = 0;
    if (has_payload) heap_push(s, payload);                         /* return [*payload_events, next_tick] (:174) */
    heap_push(s, tick);
}

/* Probe tick: Source.handle_event with _ProbeEventProvider (instrumentation/probe.py:69-78): one daemon `probe_event`
 * aimed at a CallbackEntity, then the next tick. */
static void on_probe_tick(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    hso_event pe = {e->time, next_index(s), HSO_EV_PROBE, n, -1, 0};
    nd->generated++;
    hso_event tick;
    tick.time = next_arrival(s, n);
    tick.idx = next_index(s);
    tick.kind = HSO_EV_PROBE_TICK; tick.node = n; tick.req = -1; tick.aux = 0;
    heap_push(s, pe);
    heap_push(s, tick);
}
/* measure_callback (instrumentation/probe.py:51-66): data.add_stat(getattr(target, metric), event.time) */
static void on_probe_event(hso_sim *s, const hso_event *e) {
    hso_node *nd = &s->nodes[e->node];
    const hso_node *tg = &s->nodes[s->g.target[e->node]];
    int64_t v = 0;
    switch (s->g.probe_metric[e->node]) {
        case HSO_M_DEPTH: v = tg->fifo.len; break;                   /* QueuedResource.depth */
        case HSO_M_ACTIVE: v = tg->active; break;                    /* Server.active_requests */
        case HSO_M_ACCEPTED: v = tg->accepted; break;
        case HSO_M_DROPPED: v = tg->dropped; break;
        case HSO_M_COMPLETED: v = tg->completed; break;
        case HSO_M_RECEIVED: v = tg->received; break;                /* Sink.events_received */
        case HSO_M_GENERATED: v = tg->generated; break;              /* Source.generated_count */
        default: break;
    }
    if (nd->received == nd->sink_cap) {
        nd->sink_cap = nd->sink_cap ? nd->sink_cap * 2 : 256;
        nd->sink_t = (int64_t *)realloc(nd->sink_t, (size_t)nd->sink_cap * sizeof(int64_t));
        nd->sink_created = (int64_t *)realloc(nd->sink_created, (size_t)nd->sink_cap * sizeof(int64_t));
    }
    nd->sink_t[nd->received] = e->time;
    nd->sink_created[nd->received] = v;
    nd->received++;
}

/* QueuedResource.handle_event -> Queue._handle_enqueue, queued_resource.py:139-143, queue.py:122-147 */
static void on_enqueue(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    int was_empty = nd->fifo.len == 0;                              /* queue.py:124 */
    int64_t cap = s->g.queue_cap[n];
    int32_t hook = s->reqs[e->req].lb_hook;                         /* Event.on_complete of this Event */
    s->reqs[e->req].lb_hook = -1;                                   /* hooks are one-shot (core/event.py:290-311) */
    if (cap >= 0 && nd->fifo.len >= cap) {                          /* FIFOQueue.push, queue_policy.py:94-98 */
        nd->dropped++;                                              /* queue.py:128 */
        req_release(s, e->req);
    } else {
        s->reqs[e->req].idx = e->idx;   /* the queued payload IS this Event object (a forwarded request is a new Event) */
        fifo_push(&nd->fifo, e->req);
        nd->accepted++;                                             /* queue.py:138 */
        if (was_empty) {                                            /* queue.py:144-146 */
            hso_event nf = {e->time, next_index(s), HSO_EV_NOTIFY, n, -1, 0};
            heap_push(s, nf);
        }
    }
    /* Event.invoke (core/event.py:277-283): the handler returned a plain list, so the completion hooks run NOW --
     * after the handler's own events were constructed -- and their events are appended: the LB's `_lb_response`. */
    if (hook >= 0) {
        hso_event rs = {e->time, next_index(s), HSO_EV_LB_RESP, hook, -1, 0};
        heap_push(s, rs);
    }
}

/* QueueDriver._handle_notify, queue_driver.py:92-99 */
static void on_notify(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    if (!(s->nodes[n].active < s->g.concurrency[n])) return;        /* has_capacity, concurrency.py:122-131 */
    hso_event pl = {e->time, next_index(s), HSO_EV_POLL, n, -1, 0};
    heap_push(s, pl);
}

/* Queue._handle_poll, queue.py:149-166 */
static void on_poll(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    if (nd->fifo.len == 0) return;                                  /* :151-154 */
    int32_t r = fifo_pop(&nd->fifo);
    hso_event dv = {e->time, next_index(s), HSO_EV_DELIVER, n, r, 0};
    heap_push(s, dv);
}

/* QueueDriver._handle_delivery/_handle_work_payload, queue_driver.py:66-90:
 * the SAME payload object is re-timed, re-targeted and re-pushed with its
 * original sort index; the schedule_poll completion hook is attached. */
static void on_deliver(hso_sim *s, const hso_event *e) {
    hso_event wk = {e->time, s->reqs[e->req].idx, HSO_EV_WORK, e->node, e->req, 0};
    heap_push(s, wk);
}

/* Server.handle_queued_event up to the yield, server/server.py:202-250, via
 * Event._start_process (core/event.py:313-325) and the first
 * ProcessContinuation.invoke (core/event.py:465-508). */
static void on_work(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    (void)next_index(s);              /* the continuation built by _start_process: invoked at once, never pushed */
    if (!(nd->active < s->g.concurrency[n])) {                      /* acquire failed, server.py:223-234 */
        nd->rejected++;
        req_release(s, e->req);
        /* generator returns immediately -> StopIteration; hook schedule_poll sees no capacity -> nothing */
        return;
    }
    nd->active++;
    double service_s = sample_latency_s(s, n, HS_STREAM_SERVICE, &nd->svc_draws); /* :246-247 */
    s->reqs[e->req].service_s = service_s;
    /* `yield service_time_s` -> resume_time = self.time + delay (event.py:499) = ns + int(delay*1e9) (temporal.py:222) */
    hso_event ct = {e->time + hsr_ns_from_seconds(service_s), next_index(s), HSO_EV_CONTINUATION, n, e->req, 0};
    heap_push(s, ct);
}

/* generator resumes after the yield, server/server.py:252-273; StopIteration
 * path of ProcessContinuation.invoke (core/event.py:522-533) then the
 * schedule_poll completion hook (queue_driver.py:79-84). */
static void on_continuation(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    nd->active = nd->active > 0 ? nd->active - 1 : 0;               /* release, concurrency.py:112-119 */
    nd->completed++;
    nd->total_service_s += s->reqs[e->req].service_s;               /* :256-257 */
    int32_t dn = s->g.target[n];
    if (dn >= 0) {                                                  /* forward(event, downstream), :271-272 */
        hso_event fw = {e->time, next_index(s), arrival_kind_for(s, dn), dn, e->req, 0};
        heap_push(s, fw);
    } else {
        req_release(s, e->req);
    }
    if (nd->active < s->g.concurrency[n]) {                         /* schedule_poll hook */
        hso_event pl = {e->time, next_index(s), HSO_EV_POLL, n, -1, 0};
        heap_push(s, pl);
    }
}

/* Sink.handle_event, components/common.py:36-44 */
static void on_sink(hso_sim *s, const hso_event *e) {
    hso_node *nd = &s->nodes[e->node];
    if (nd->received == nd->sink_cap) {
        nd->sink_cap = nd->sink_cap ? nd->sink_cap * 2 : 256;
        nd->sink_t = (int64_t *)realloc(nd->sink_t, (size_t)nd->sink_cap * sizeof(int64_t));
        nd->sink_created = (int64_t *)realloc(nd->sink_created, (size_t)nd->sink_cap * sizeof(int64_t));
    }
    nd->sink_t[nd->received] = e->time;
    nd->sink_created[nd->received] = s->reqs[e->req].created_ns;
    nd->received++;
    int32_t hook = s->reqs[e->req].lb_hook;                         /* a Sink used directly as an LB backend */
    s->reqs[e->req].lb_hook = -1;
    req_release(s, e->req);
    if (hook >= 0) {
        hso_event rs = {e->time, next_index(s), HSO_EV_LB_RESP, hook, -1, 0};
        heap_push(s, rs);
    }
}

/* RandomRouter.handle_event, components/random_router.py:32-45.  The stock router calls
 * random.randint (global MT19937); the Philox-plugged router of tests/golden/make_golden.py picks
 * idx = int(u * len(targets)) from the node's ROUTE stream -- same handler otherwise. */
static void on_route(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    nd->routed++;
    int32_t cnt = s->g.rt_cnt[n];
    double u = draw_uniform(s, n, HS_STREAM_ROUTE, &nd->route_draws);
    int32_t idx = (int32_t)(u * (double)cnt);
    int32_t tgt = s->g.rt_targets[s->g.rt_off[n] + idx];
    hso_event ev = {e->time, next_index(s), arrival_kind_for(s, tgt), tgt, e->req, 0};
    heap_push(s, ev);
}

/* NetworkLink.handle_event up to its yield, components/network/link.py:114-154.  Packet loss (:131-138) is decided
 * first -- the generator returns before its first yield, so the event costs the continuation's sort index and nothing
 * else; `dropped` of the link node = packets_dropped.  Bandwidth: the lowered providers put no payload_size in the
 * metadata, so the transmission time (:209-214) is 0 * 8 / bandwidth = 0.0 whatever the bandwidth.
 * delay = latency.get_latency(now).to_seconds() [+ jitter.get_latency(now).to_seconds()], max(0, .)
 * (_calculate_delay :190-216).  Modelled link: latency = ConstantLatency(lat_min),
 * jitter = ExponentialLatency(lat_mean) when lat_kind == EXP, ConstantLatency(lat_mean) when lat_kind == CONST and
 * lat_mean > 0 (the reference's datacenter_network preset, components/network/conditions.py:60-63), none otherwise. */
static void on_link(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    (void)next_index(s);                                            /* continuation built by _start_process */
    if (s->g.loss[n] > 0.0 && draw_uniform(s, n, HS_STREAM_LOSS, &nd->loss_draws) < s->g.loss[n]) {
        nd->dropped++;                                              /* packets_dropped, link.py:132 */
        req_release(s, e->req);
        return;
    }
    /* PartitionLink.packet_loss: the coordinator drops the cross-partition event at the exchange (parallel/coordinator.py:203-205:
     * `if link.packet_loss > 0 and self._rng.random() < link.packet_loss: continue`) -- it never reaches the destination */
    if (s->g.ploss[n] > 0.0 && hsr_mt_res53(&s->mt_coord) < s->g.ploss[n]) {
        nd->dropped++;
        req_release(s, e->req);
        return;
    }
    double delay = hsr_seconds_from_ns(hsr_ns_from_seconds(s->g.lat_min[n]));      /* ConstantLatency */
    if (s->g.lat_kind[n] == HSO_LAT_EXP) {
        double lambda = 1.0 / s->g.lat_mean[n];
        double sample = exp1(s, draw_uniform(s, n, HS_STREAM_LINK, &nd->link_draws)) / lambda;
        delay = delay + hsr_seconds_from_ns(hsr_ns_from_seconds(sample));           /* jitter */
    } else if (s->g.lat_mean[n] > 0.0) {
        delay = delay + hsr_seconds_from_ns(hsr_ns_from_seconds(s->g.lat_mean[n])); /* jitter = ConstantLatency(lat_mean): no draw */
    }
    if (!(delay > 0.0)) delay = 0.0;                                /* max(0.0, delay) */
    hso_event ct = {e->time + hsr_ns_from_seconds(delay), next_index(s), HSO_EV_LINK_CONT, n, e->req, 0};
    heap_push(s, ct);
}

/* transit over: link.py:156-189 -- a NEW Event for the egress with a copy of the context */
static void on_link_cont(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    nd->packets_sent++;
    int32_t eg = s->g.target[n];
    if (eg < 0) { req_release(s, e->req); return; }
    hso_event fw = {e->time, next_index(s), arrival_kind_for(s, eg), eg, e->req, 0};
    heap_push(s, fw);
}

/* ------------------------------------------------------------------- md5 (RFC 1321)
 * ConsistentHash._hash = int(hashlib.md5(key.encode()).hexdigest(), 16), strategies.py:377-379. */
void hso_md5(const char *msg, int64_t len, uint8_t out[16]) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
        0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
        0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20,
                              5, 9, 14, 20, 5, 9, 14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t h0 = 0x67452301, h1 = 0xefcdab89, h2 = 0x98badcfe, h3 = 0x10325476;
    int64_t padded = ((len + 8) / 64 + 1) * 64;
    uint8_t *buf = (uint8_t *)calloc((size_t)padded, 1);
    memcpy(buf, msg, (size_t)len);
    buf[len] = 0x80;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) buf[padded - 8 + i] = (uint8_t)(bits >> (8 * i));
    for (int64_t off = 0; off < padded; off += 64) {
        uint32_t M[16];
        for (int i = 0; i < 16; ++i)
            M[i] = (uint32_t)buf[off + 4 * i] | ((uint32_t)buf[off + 4 * i + 1] << 8) |
                   ((uint32_t)buf[off + 4 * i + 2] << 16) | ((uint32_t)buf[off + 4 * i + 3] << 24);
        uint32_t a = h0, b = h1, c = h2, d = h3;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            uint32_t x = a + f + K[i] + M[g];
            a = d; d = c; c = b;
            b = b + ((x << R[i]) | (x >> (32 - R[i])));
        }
        h0 += a; h1 += b; h2 += c; h3 += d;
    }
    free(buf);
    uint32_t hs[4] = {h0, h1, h2, h3};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(hs[i] >> (8 * j));
}
static void md5_u128(const char *msg, int64_t len, uint64_t *hi, uint64_t *lo) {
    uint8_t d[16];
    hso_md5(msg, len, d);
    uint64_t h = 0, l = 0;
    for (int i = 0; i < 8; ++i) { h = (h << 8) | d[i]; l = (l << 8) | d[8 + i]; }   /* int(hexdigest, 16): big-endian */
    *hi = h; *lo = l;
}
static int ring_cmp(const void *pa, const void *pb) {
    const hso_ring_pt *a = (const hso_ring_pt *)pa, *b = (const hso_ring_pt *)pb;
    if (a->hi != b->hi) return a->hi < b->hi ? -1 : 1;
    if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
    return a->seq < b->seq ? -1 : (a->seq > b->seq);        /* list.sort is stable: insertion order on equal hashes */
}
/* ConsistentHash.add_backend for every backend in LoadBalancer.__init__ order (strategies.py:381-391) */
static void lb_build_ring(hso_sim *s, int32_t n) {
    hso_node *nd = &s->nodes[n];
    int32_t cnt = s->g.rt_cnt[n], v = s->g.vnodes[n] > 0 ? s->g.vnodes[n] : 0;    /* (<= 0: RoundRobin / Random -- no ring) */
    nd->ring_len = (int64_t)cnt * v;
    nd->ring = (hso_ring_pt *)malloc((size_t)(nd->ring_len > 0 ? nd->ring_len : 1) * sizeof(hso_ring_pt));
    nd->lb_total_requests = (int64_t *)calloc((size_t)(cnt > 0 ? cnt : 1), sizeof(int64_t));
    char key[512];
    int64_t k = 0;
    for (int32_t b = 0; b < cnt; ++b) {
        int32_t be = s->g.rt_targets[s->g.rt_off[n] + b];
        int32_t nl = s->g.name_off[be + 1] - s->g.name_off[be];
        for (int32_t i = 0; i < v; ++i) {
            memcpy(key, s->g.names + s->g.name_off[be], (size_t)nl);
            int m = snprintf(key + nl, sizeof(key) - (size_t)nl, ":%d", i);     /* f"{backend.name}:{i}" */
            md5_u128(key, nl + m, &nd->ring[k].hi, &nd->ring[k].lo);
            nd->ring[k].backend = be; nd->ring[k].seq = (int32_t)k;
            ++k;
        }
    }
    qsort(nd->ring, (size_t)nd->ring_len, sizeof(hso_ring_pt), ring_cmp);
}
/* ConsistentHash.select (strategies.py:412-433): first ring point with hash >= md5(key), else the first point */
static int32_t lb_select_key(const hso_sim *s, int32_t n, const char *key, int64_t len) {
    const hso_node *nd = &s->nodes[n];
    uint64_t hi, lo;
    md5_u128(key, len, &hi, &lo);
    int64_t a = 0, b = nd->ring_len;                                /* lower bound == the reference's linear scan */
    while (a < b) {
        int64_t m = (a + b) >> 1;
        const hso_ring_pt *p = &nd->ring[m];
        int ge = (p->hi > hi) || (p->hi == hi && p->lo >= lo);
        if (ge) b = m; else a = m + 1;
    }
    if (a == nd->ring_len) a = 0;
    return nd->ring[a].backend;
}
int32_t hso_lb_select(const hso_sim *s, int32_t node, const char *key) { return lb_select_key(s, node, key, (int64_t)strlen(key)); }

/* LoadBalancer._forward_request, load_balancer.py:347-433 (all backends healthy).  Strategy by `vnodes`: > 0 ConsistentHash
 * (strategies.py:336-433); 0 RoundRobin (strategies.py:50-73: backends[_index % len], _index += 1 per select -- the LB's processing
 * order); -1 Random (strategies.py:137-150: random.choice(backends)) with the choice plugged like every other draw of the
 * seed-matched oracle: backends[int(u * len)], u the Request's draw from its Source's KEY stream (client_id holds the index). */
static void on_lb(hso_sim *s, const hso_event *e) {
    int32_t n = e->node;
    hso_node *nd = &s->nodes[n];
    nd->lb_received++;                                              /* :349 */
    if (s->g.rt_cnt[n] == 0) {                                      /* no healthy backends, :352-366 */
        nd->lb_no_backend++; nd->lb_failed++;
        req_release(s, e->req);
        return;
    }
    int32_t be;
    if (s->g.vnodes[n] > 0 && s->reqs[e->req].client_id < 0) {
        /* no metadata, so _default_get_key returns None: `return self._fallback.select(backends, request)` (strategies.py:362,420-421),
         * a RoundRobin of the strategy's own that only the key-less Requests advance */
        be = s->g.rt_targets[s->g.rt_off[n] + (int32_t)(nd->lb_fallback_idx % s->g.rt_cnt[n])];
        nd->lb_fallback_idx++;
    } else if (s->g.vnodes[n] > 0) {
        char key[32];
        int len = snprintf(key, sizeof key, "%lld", (long long)s->reqs[e->req].client_id);   /* str(metadata["client_id"]) */
        be = lb_select_key(s, n, key, len);
    } else if (s->g.vnodes[n] == 0) {
        be = s->g.rt_targets[s->g.rt_off[n] + (int32_t)(nd->lb_next_id % s->g.rt_cnt[n])];  /* RoundRobin._index == selects so far */
    } else {
        int64_t c = s->reqs[e->req].client_id;
        if (c < 0 || c >= s->g.rt_cnt[n]) c = 0;
        be = s->g.rt_targets[s->g.rt_off[n] + (int32_t)c];
    }
    nd->lb_next_id++;                                               /* :375-376 */
    nd->lb_in_flight++;                                             /* :378-382 */
    for (int32_t b = 0; b < s->g.rt_cnt[n]; ++b)
        if (s->g.rt_targets[s->g.rt_off[n] + b] == be) { nd->lb_total_requests[b]++; break; }   /* :385-386 */
    nd->lb_forwarded++;                                             /* :388 */
    /* a NEW Event for the backend, same context (created_at survives), + the response hook (:398-431) */
    s->reqs[e->req].lb_hook = n;
    hso_event fw = {e->time, next_index(s), arrival_kind_for(s, be), be, e->req, 0};
    heap_push(s, fw);
}
/* LoadBalancer._handle_response, load_balancer.py:435-473: bookkeeping only */
static void on_lb_resp(hso_sim *s, const hso_event *e) {
    hso_node *nd = &s->nodes[e->node];
    if (nd->lb_in_flight > 0) nd->lb_in_flight--;
}

void hso_get_lb_stats(const hso_sim *s, int32_t node, int64_t out[6], int64_t *total_requests, int32_t *ring_backend) {
    const hso_node *nd = &s->nodes[node];
    out[0] = nd->lb_received; out[1] = nd->lb_forwarded; out[2] = nd->lb_failed; out[3] = nd->lb_no_backend;
    out[4] = nd->lb_in_flight;
    out[5] = s->g.vnodes[node] > 0 ? nd->lb_fallback_idx : s->g.vnodes[node] == 0 ? nd->lb_next_id : -1;   /* the strategy's RoundRobin._index */
    if (total_requests) memcpy(total_requests, nd->lb_total_requests, (size_t)s->g.rt_cnt[node] * 8);
    if (ring_backend) for (int64_t i = 0; i < nd->ring_len; ++i) ring_backend[i] = nd->ring[i].backend;
}

/* ------------------------------------------------------------------- API */
#define DUP(field, type)                                                         \
    do {                                                                         \
        type *c_ = (type *)malloc((size_t)n * sizeof(type));                     \
        if (g->field) memcpy(c_, g->field, (size_t)n * sizeof(type));            \
        else memset(c_, 0, (size_t)n * sizeof(type));                            \
        s->g.field = c_;                                                         \
    } while (0)

hso_sim *hso_create(const hso_graph *g, const hso_params *p) {
    hso_sim *s = (hso_sim *)calloc(1, sizeof(hso_sim));
    int32_t n = g->n_nodes;
    s->g.n_nodes = n;
    DUP(kind, int32_t); DUP(target, int32_t); DUP(stream_base, uint64_t);
    DUP(arr_kind, int32_t); DUP(rate, double); DUP(stop_after_ns, int64_t);
    DUP(concurrency, int32_t); DUP(lat_kind, int32_t); DUP(lat_mean, double); DUP(lat_min, double);
    DUP(queue_cap, int64_t); DUP(rt_off, int32_t); DUP(rt_cnt, int32_t);
    DUP(n_clients, int64_t); DUP(vnodes, int32_t); DUP(prof_kind, int32_t); DUP(probe_metric, int32_t); DUP(loss, double); DUP(ploss, double);
    {
        double *pp = (double *)calloc((size_t)n * 4 + 1, sizeof(double));
        if (g->prof_p) memcpy(pp, g->prof_p, (size_t)n * 4 * sizeof(double));
        s->g.prof_p = pp;
    }
    {
        int32_t *no = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
        if (g->name_off) memcpy(no, g->name_off, ((size_t)n + 1) * sizeof(int32_t));
        s->g.name_off = no;
        int32_t nb = no[n] > 0 ? no[n] : 1;
        char *nm = (char *)calloc((size_t)nb, 1);
        if (g->names && no[n] > 0) memcpy(nm, g->names, (size_t)no[n]);
        s->g.names = nm;
    }
    {
        int32_t m = g->n_rt > 0 ? g->n_rt : 1;
        int32_t *c_ = (int32_t *)calloc((size_t)m, sizeof(int32_t));
        if (g->rt_targets && g->n_rt > 0) memcpy(c_, g->rt_targets, (size_t)g->n_rt * sizeof(int32_t));
        s->g.rt_targets = c_; s->g.n_rt = g->n_rt;
    }
    s->p = *p;
    s->nodes = (hso_node *)calloc((size_t)n, sizeof(hso_node));
    s->req_free = -1;
    s->current_ns = p->start_ns;
    if (p->trace_cap > 0) {
        s->tr_t = (int64_t *)malloc((size_t)p->trace_cap * 8);
        s->tr_kind = (int32_t *)malloc((size_t)p->trace_cap * 4);
        s->tr_node = (int32_t *)malloc((size_t)p->trace_cap * 4);
        s->tr_idx = (int64_t *)malloc((size_t)p->trace_cap * 8);
    }
    { uint32_t ck = p->coord_seed; hsr_mt_init_by_array(&s->mt_coord, &ck, 1); }   /* random.Random(seed), parallel/coordinator.py:68 */
    if (p->rng_mode == HSO_RNG_MT19937) {
        uint32_t key = p->mt_seed_py;
        hsr_mt_init_by_array(&s->mt_py, &key, 1);   /* random.seed(int) */
        hsr_mt_init_genrand(&s->mt_np, p->mt_seed_np); /* np.random.seed(int) */
    }
    /* Simulation.__init__ bootstrap, core/simulation.py:145-154 + Source.start, load/source.py:120-140:
     * sources in list order; their first SourceEvents take indices 0..S-1 from the GLOBAL counter. */
    for (int32_t i = 0; i < n; ++i) if (s->g.kind[i] == HSO_LB) lb_build_ring(s, i);
    s->counter = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (s->g.kind[i] != HSO_SOURCE) continue;
        s->nodes[i].arr_time_ns = p->start_ns;                      /* provider.current_time = start_time */
        hso_event tick;
        tick.time = next_arrival(s, i);
        tick.idx = next_index(s);
        tick.kind = HSO_EV_SOURCE; tick.node = i; tick.req = -1; tick.aux = 0;
        heap_push(s, tick);
    }
    /* then the probes, in list order (core/simulation.py:156-160) */
    for (int32_t i = 0; i < n; ++i) {
        if (s->g.kind[i] != HSO_PROBE) continue;
        s->nodes[i].arr_time_ns = p->start_ns;
        hso_event tick;
        tick.time = next_arrival(s, i);
        if (tick.time == INT64_MAX) continue;                       /* "Rate is zero indefinitely. Source will not start." */
        tick.idx = next_index(s);
        tick.kind = HSO_EV_PROBE_TICK; tick.node = i; tick.req = -1; tick.aux = 0;
        heap_push(s, tick);
    }
    /* run(): _active_sim_context switches Event construction to the per-heap
     * counter, which starts again at 0 (core/event_heap.py:48, core/sim_future.py:64-73). */
    s->global_counter = s->counter;
    s->counter = 0;
    return s;
}

/* Simulation.schedule(Event(time, "Request", target=<node>)) before run() (core/simulation.py:195-206): the Event was
 * constructed outside the run, so its sort index comes from the process-wide counter, which Simulation.__init__ reset
 * and the bootstrap SourceEvents advanced (core/event.py:53-67,165); context["created_at"] = its own time (:176).
 * Calls must come in the order the caller constructed the Events.  Returns 0, or -1 for a node that takes no Request. */
int hso_schedule(hso_sim *s, int32_t node, int64_t time_ns) {
    int32_t k = arrival_kind_for(s, node);
    /* (a scheduled Request carries no metadata: a ConsistentHash LoadBalancer falls back to a RoundRobin of its own (on_lb), a Random
     * one would ask the process-wide generator -- not modelled) */
    if (k < 0 || (k == HSO_EV_LB && s->g.vnodes[node] < 0)) return -1;
    int32_t r = req_alloc(s);
    s->reqs[r].created_ns = time_ns;
    s->reqs[r].hops = 0;
    s->reqs[r].service_s = 0.0;
    s->reqs[r].client_id = -1;
    s->reqs[r].lb_hook = -1;
    hso_event ev = {time_ns, s->global_counter++, k, node, r, 0};
    s->reqs[r].idx = ev.idx;
    heap_push(s, ev);
    return 0;
}

/* Simulation._execute_until, core/simulation.py:449-505 (no cancellation on this path) */
int hso_run_until(hso_sim *s, int64_t end_ns) {
    while (s->heap_len > 0 && s->current_ns <= end_ns) {            /* :472 tests the PREVIOUS event's time */
        hso_event e = heap_pop(s);
        if (e.time < s->current_ns) continue;                       /* time travel drop, :480-489 */
#ifdef HSO_LINEAGE
        if (e.time > end_ns && s->dump == NULL) {
            s->dump = (hso_event *)malloc((size_t)(s->heap_len + 1) * sizeof(hso_event));
            s->dump[0] = e;
            memcpy(s->dump + 1, s->heap, (size_t)s->heap_len * sizeof(hso_event));
            s->dump_len = s->heap_len + 1;
        }
        s->cur_valid = 1; s->cur_crt = e.crt; s->cur_crt2 = e.crt2; s->cur_rsrc = e.rsrc;
        if (e.crt < e.time) {          /* created earlier: the root of a chain of this nanosecond's group */
            s->g_depth = 0; s->g_rcrt = e.crt; s->g_rcdepth = e.cdepth; s->g_r2crt = e.rcrt; s->g_r2cdepth = e.rcdepth;
        } else {                       /* created in this very nanosecond: it carries its group's context */
            s->g_depth = e.cdepth; s->g_rcrt = e.rcrt; s->g_rcdepth = e.rcdepth; s->g_r2crt = e.r2crt; s->g_r2cdepth = e.r2cdepth;
        }
#endif
        s->current_ns = e.time;
        s->processed++;
        s->by_kind[e.kind]++;
        trace(s, &e);
        switch (e.kind) {
            case HSO_EV_SOURCE: on_source(s, &e); break;
            case HSO_EV_ENQUEUE: on_enqueue(s, &e); break;
            case HSO_EV_NOTIFY: on_notify(s, &e); break;
            case HSO_EV_POLL: on_poll(s, &e); break;
            case HSO_EV_DELIVER: on_deliver(s, &e); break;
            case HSO_EV_WORK: on_work(s, &e); break;
            case HSO_EV_CONTINUATION: on_continuation(s, &e); break;
            case HSO_EV_SINK: on_sink(s, &e); break;
            case HSO_EV_LINK: on_link(s, &e); break;
            case HSO_EV_LINK_CONT: on_link_cont(s, &e); break;
            case HSO_EV_ROUTE: on_route(s, &e); break;
            case HSO_EV_LB: on_lb(s, &e); break;
            case HSO_EV_LB_RESP: on_lb_resp(s, &e); break;
            case HSO_EV_PROBE_TICK: on_probe_tick(s, &e); break;
            case HSO_EV_PROBE: on_probe_event(s, &e); break;
            default: return -1;
        }
    }
    return 0;
}

#ifdef HSO_LINEAGE
/* rows of (time, sort index, kind, node, created, creator created, its creator created, cdepth, rcrt, rcdepth, r2crt,
 * r2cdepth, rsrc); row 0 = the event the reference processed beyond end_ns */
int64_t hso_read_dump(const hso_sim *s, int64_t *rows, int64_t cap) {
    const int64_t n = s->dump_len < cap ? s->dump_len : cap;
    for (int64_t i = 0; i < n; ++i) {
        const hso_event *e = &s->dump[i];
        int64_t *r = rows + 13 * i;
        r[0] = e->time; r[1] = (int64_t)e->idx; r[2] = e->kind; r[3] = e->node; r[4] = e->crt; r[5] = e->crt2; r[6] = e->crt3;
        r[7] = e->cdepth; r[8] = e->rcrt; r[9] = e->rcdepth; r[10] = e->r2crt; r[11] = e->r2cdepth; r[12] = e->rsrc;
    }
    return s->dump_len;
}
#endif

void hso_get_summary(const hso_sim *s, hso_summary *out) {
    out->events_processed = s->processed;
    memcpy(out->events_by_kind, s->by_kind, sizeof(s->by_kind));
    out->final_time_ns = s->current_ns;
    out->heap_peak = s->heap_peak;
    out->sort_index_next = (int64_t)s->counter;
}

void hso_get_node_stats(const hso_sim *s, int64_t *generated, int64_t *accepted, int64_t *dropped,
                        int64_t *completed, int64_t *rejected, double *total_service_s,
                        int64_t *received, int64_t *depth, int64_t *active) {
    for (int32_t i = 0; i < s->g.n_nodes; ++i) {
        const hso_node *nd = &s->nodes[i];
        if (generated) generated[i] = nd->generated;
        if (accepted) accepted[i] = nd->accepted;
        if (dropped) dropped[i] = nd->dropped;
        if (completed) completed[i] = nd->completed;
        if (rejected) rejected[i] = nd->rejected;
        if (total_service_s) total_service_s[i] = nd->total_service_s;
        if (received) received[i] = nd->received;
        if (depth) depth[i] = nd->fifo.len;
        if (active) active[i] = nd->active;
    }
}

void hso_get_net_stats(const hso_sim *s, int64_t *packets_sent, int64_t *routed) {
    for (int32_t i = 0; i < s->g.n_nodes; ++i) {
        if (packets_sent) packets_sent[i] = s->nodes[i].packets_sent;
        if (routed) routed[i] = s->nodes[i].routed;
    }
}

int64_t hso_sink_count(const hso_sim *s, int32_t node) { return s->nodes[node].received; }

int64_t hso_read_sink(const hso_sim *s, int32_t node, int64_t *t_ns, int64_t *created_ns, int64_t cap) {
    const hso_node *nd = &s->nodes[node];
    int64_t n = nd->received < cap ? nd->received : cap;
    if (n > 0) {
        memcpy(t_ns, nd->sink_t, (size_t)n * 8);
        memcpy(created_ns, nd->sink_created, (size_t)n * 8);
    }
    return n;
}

int64_t hso_read_trace(const hso_sim *s, int64_t *t_ns, int32_t *kind, int32_t *node, int64_t *sort_index,
                       int64_t cap) {
    int64_t n = s->tr_len < cap ? s->tr_len : cap;
    if (n > 0) {
        memcpy(t_ns, s->tr_t, (size_t)n * 8);
        memcpy(kind, s->tr_kind, (size_t)n * 4);
        memcpy(node, s->tr_node, (size_t)n * 4);
        memcpy(sort_index, s->tr_idx, (size_t)n * 8);
    }
    return n;
}

void hso_destroy(hso_sim *s) {
    if (!s) return;
#ifdef HSO_LINEAGE
    free(s->dump);
#endif
    for (int32_t i = 0; i < s->g.n_nodes; ++i) {
        free(s->nodes[i].fifo.buf); free(s->nodes[i].sink_t); free(s->nodes[i].sink_created);
        free(s->nodes[i].ring); free(s->nodes[i].lb_total_requests);
    }
    free((void *)s->g.kind); free((void *)s->g.target); free((void *)s->g.stream_base);
    free((void *)s->g.arr_kind); free((void *)s->g.rate); free((void *)s->g.stop_after_ns);
    free((void *)s->g.concurrency); free((void *)s->g.lat_kind); free((void *)s->g.lat_mean);
    free((void *)s->g.prof_kind); free((void *)s->g.prof_p); free((void *)s->g.probe_metric); free((void *)s->g.loss); free((void *)s->g.ploss);
    free((void *)s->g.n_clients); free((void *)s->g.vnodes); free((void *)s->g.names); free((void *)s->g.name_off);
    free((void *)s->g.lat_min); free((void *)s->g.queue_cap); free((void *)s->g.rt_off); free((void *)s->g.rt_cnt); free((void *)s->g.rt_targets);
    free(s->nodes); free(s->heap); free(s->reqs);
    free(s->tr_t); free(s->tr_kind); free(s->tr_node); free(s->tr_idx);
    free(s);
}

/* ------------------------------------------------------ scalar exports */
double hso_uniform(uint64_t seed, uint64_t sid, uint64_t k) { return hsr_uniform(seed, sid, k); }
double hso_log(double x) { return hs_log_ref(x); }
void hso_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { hsr_philox4x32_10(ctr, key, out); }
double hso_mt_py_random(uint32_t seed, int64_t n_skip) {
    hsr_mt19937 g; uint32_t key = seed; hsr_mt_init_by_array(&g, &key, 1);
    double v = 0; for (int64_t i = 0; i <= n_skip; ++i) v = hsr_mt_res53(&g);
    return v;
}
double hso_mt_np_random(uint32_t seed, int64_t n_skip) {
    hsr_mt19937 g; hsr_mt_init_genrand(&g, seed);
    double v = 0; for (int64_t i = 0; i <= n_skip; ++i) v = hsr_mt_res53(&g);
    return v;
}
