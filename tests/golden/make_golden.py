#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the LIVE upstream reference.

Run by hand in the build container (needs /root/reference):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Two families:
  * Oracle-A: the stock reference, `random.seed(s); np.random.seed(s)` (MT19937
    global streams) -- the README quick-start and the constant-rate cases.
  * Oracle-B: the same reference engine with per-entity counter-based streams
    plugged in through its own extension points (Seam 3): a
    `LatencyDistribution` subclass and an `ArrivalTimeProvider` subclass that
    draw from tests/golden/hs_streams_py.py.  The reference's event loop, queue
    protocol, server generator, truncation rules and sort-index ledger all run
    unmodified.

Every case records: the spec (so tests can rebuild the same graph for the C
oracle and the HIP engine), summary scalars, per-chain statistics, every Sink
record (completion ns + latency_s), and -- for small cases -- the full
processed-event trace (time ns, kind, node, sort index) captured by wrapping
the simulation's `EventHeap.pop`.

The GPU box has no /root/reference; tests only read the .npz files.
"""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import refshim  # noqa: E402

refshim.install()

import hs_streams_py as hs  # noqa: E402
from happysimulator import (  # noqa: E402
    ConstantLatency, ExponentialLatency, Instant, Server, Simulation, Sink, Source,
)
from happysimulator.components.queued_resource import _QueuedResourceWorkerAdapter  # noqa: E402
from happysimulator.core.event import ProcessContinuation  # noqa: E402
from happysimulator.core.temporal import Duration  # noqa: E402
from happysimulator.distributions.latency_distribution import LatencyDistribution  # noqa: E402
from happysimulator.load.arrival_time_provider import ArrivalTimeProvider  # noqa: E402
from happysimulator.load.profile import ConstantRateProfile, LinearRampProfile, SpikeProfile  # noqa: E402
from happysimulator.load.providers.constant_arrival import ConstantArrivalTimeProvider  # noqa: E402
from happysimulator.load.source import SimpleEventProvider  # noqa: E402

from happysimulator.components.network.link import NetworkLink  # noqa: E402
from happysimulator.components.random_router import RandomRouter  # noqa: E402
from happysimulator.core.event import Event  # noqa: E402

from happysimulator.components.load_balancer.load_balancer import LoadBalancer  # noqa: E402
from happysimulator.components.load_balancer.strategies import ConsistentHash  # noqa: E402
from happysimulator.load.event_provider import EventProvider  # noqa: E402

from happysimulator.instrumentation.probe import Probe  # noqa: E402

EV = {"source": 0, "enqueue": 1, "notify": 2, "poll": 3, "deliver": 4, "work": 5, "continuation": 6, "sink": 7,
      "link": 8, "link_cont": 9, "route": 10, "lb": 11, "lb_resp": 12, "probe_tick": 13, "probe": 14}
# metric name -> (which entity of the chain carries it, the reference attribute read by getattr)
PROBE_METRICS = {"depth": ("server", "depth"), "active_requests": ("server", "active_requests"),
                 "stats_accepted": ("server", "stats_accepted"), "stats_dropped": ("server", "stats_dropped"),
                 "requests_completed": ("server", "_requests_completed"), "events_received": ("sink", "events_received"),
                 "generated_count": ("source", "generated_count"),
                 # functions of active_requests and the concurrency (live tests only: the engine samples active_requests)
                 "available_capacity": ("server", "available_capacity"), "has_capacity": ("server", "has_capacity")}


# ---- Seam-3 plug-ins (ours; they only choose the random numbers) ------------------------
class PhiloxExponentialLatency(LatencyDistribution):
    """Same arithmetic as distributions/exponential.py:29-45 with u from a Philox stream and
    hs_log in place of math.log:  sample = -log(1-u) / lambda,  lambda = 1/mean."""

    def __init__(self, mean_latency, stream: hs.Stream):
        super().__init__(mean_latency)
        self._lambda = 1 / self._mean_latency
        self._stream = stream

    def get_latency(self, current_time):
        sample = hs.exp1(self._stream.next_uniform()) / self._lambda
        return Duration.from_seconds(sample)


class PhiloxPoissonArrival(ArrivalTimeProvider):
    """load/providers/poisson_arrival.py:29-31 with u from a Philox stream and hs_log."""

    def __init__(self, profile, start_time, stream: hs.Stream):
        super().__init__(profile, start_time)
        self._stream = stream

    def _get_target_integral_value(self) -> float:
        return hs.exp1(self._stream.next_uniform())


class PhiloxRandomRouter(RandomRouter):
    """components/random_router.py:32-45 with the target index drawn from a Philox stream
    (idx = int(u * len(targets))) instead of the global `random.randint`; the handler body is otherwise the
    reference's: count, build a new Event for the chosen target with the same context."""

    def __init__(self, name, *, targets, stream: hs.Stream):
        super().__init__(name, targets=targets)
        self._stream = stream

    def handle_event(self, event):
        self.stats_routed += 1
        idx = int(self._stream.next_uniform() * len(self.targets))
        self.target_counts[self.targets[idx].name] += 1
        return [Event(time=self.now, event_type=event.event_type, target=self.targets[idx], context=event.context)]


class _PerLinkRandom:
    """Stands in for the `random` module inside components/network/link.py so that the reference's own loss test
    (`random.random() < self.packet_loss_rate`, link.py:131) reads the calling link's Philox LOSS stream instead of the
    process-wide MT19937; the handler body that runs is the reference's, untouched."""

    @staticmethod
    def random():
        link = sys._getframe(1).f_locals["self"]
        return link._loss_stream.next_uniform()


class _PerRequestChoice:
    """Stands in for the `random` module inside components/load_balancer/strategies.py so that the reference's own
    `Random.select` (`random.choice(backends)`, strategies.py:146-150) picks backends[client_id] -- the index the Request's Source
    drew from its KEY stream, int(u * len(backends)) (PhiloxClientProvider with n_clients = number of backends) -- instead of asking
    the process-wide MT19937; the strategy's body that runs is the reference's, untouched."""

    @staticmethod
    def choice(seq):
        request = sys._getframe(1).f_locals["request"]
        return seq[int(request.context["metadata"]["client_id"])]


class PhiloxClientProvider(EventProvider):
    """The request factory of examples/visual/chash_example.py:69-88 (`ClientRequestProvider`): one Request per
    tick whose metadata carries a random client_id for the ConsistentHash strategy -- with the id drawn from the
    source's KEY stream (client_id = int(u * n_clients)) instead of a private `random.Random`."""

    def __init__(self, target, n_clients: int, stream: hs.Stream, stop_after=None):
        self._target = target
        self._n_clients = n_clients
        self._stream = stream
        self._stop_after = stop_after

    def get_events(self, time):
        if self._stop_after is not None and time > self._stop_after:
            return []
        client_id = int(self._stream.next_uniform() * self._n_clients)
        return [Event(time=time, event_type="Request", target=self._target,
                      context={"metadata": {"client_id": str(client_id)}})]


def _per_chain(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def source_plan(spec, chain_ids):
    """`sources=[...]` of a case: the list order as (chain, slot) pairs and, per chain, its Sources in slot order
    [(arrival kind, rate, is the chain's `rate` / `arr` / `profile` Source)].  A chain's slot = the position among ITS Sources
    in the list.  sources_order "chain" (default): every chain's first Source followed by its `more_sources`; "extras_first":
    the `more_sources` (last chain first) before all first Sources."""
    n = spec["n_chains"]
    rate, arr = _per_chain(spec["rate"], n), _per_chain(spec["arr"], n)
    profiles = spec.get("profile") or [None] * n
    more = spec.get("more_sources") or [None] * n
    has_first = {c: not (rate[c] == 0 and profiles[c] is None) for c in chain_ids}
    firsts = [(c, (arr[c], rate[c], True)) for c in chain_ids if has_first[c]]
    if spec.get("sources_order") == "extras_first":
        listed = [(c, (xa, xr, False)) for c in reversed(chain_ids) for xa, xr in (more[c] or [])] + firsts
    else:
        listed = []
        for c in chain_ids:
            listed += [(c, (arr[c], rate[c], True))] * has_first[c] + [(c, (xa, xr, False)) for xa, xr in (more[c] or [])]
    slot_plan = {c: [] for c in chain_ids}
    order = []
    for c, what in listed:
        order.append((c, len(slot_plan[c])))
        slot_plan[c].append(what)
    return order, slot_plan


def ring_source_plan(spec):
    """source_plan for a ring case: station i's first Source is Poisson(ext_rate[i]) (or its profile), `more_sources[i]` lists
    [arrival kind, rate] of further Sources feeding the same Server."""
    n = spec["n"]
    rate = spec["ext_rate"] if isinstance(spec["ext_rate"], list) else [spec["ext_rate"]] * n
    more = spec.get("more_sources") or [None] * n
    firsts = [(i, ("poisson", rate[i], True)) for i in range(n) if rate[i] > 0]
    if spec.get("sources_order") == "extras_first":
        listed = [(i, (xa, xr, False)) for i in reversed(range(n)) for xa, xr in (more[i] or [])] + firsts
    else:
        listed = []
        for i in range(n):
            listed += [(i, ("poisson", rate[i], True))] * (rate[i] > 0) + [(i, (xa, xr, False)) for xa, xr in (more[i] or [])]
    slot_plan = {i: [] for i in range(n)}
    order = []
    for i, what in listed:
        order.append((i, len(slot_plan[i])))
        slot_plan[i].append(what)
    return order, slot_plan


def build_chains(spec, chain_ids, seed):
    """Build reference entities for the given chains; returns (sources, entities, handles)."""
    n = spec["n_chains"]
    rate = _per_chain(spec["rate"], n)
    mean = _per_chain(spec["mean"], n)
    conc = _per_chain(spec.get("concurrency", 1), n)
    qcap = _per_chain(spec.get("queue_cap"), n)
    arr = _per_chain(spec["arr"], n)
    svc = _per_chain(spec["svc"], n)
    stop = spec.get("stop_after_s")
    profiles = spec.get("profile") or [None] * n          # per chain: None | ["ramp", d, s, e] | ["spike", b, s, w, d]

    def make_profile(i):
        pr = profiles[i]
        if pr is None:
            return ConstantRateProfile(rate=rate[i])
        if pr[0] == "ramp":
            return LinearRampProfile(duration_s=pr[1], start_rate=pr[2], end_rate=pr[3])
        return SpikeProfile(baseline_rate=pr[1], spike_rate=pr[2], warmup_s=pr[3], spike_duration_s=pr[4])

    sources, entities, handles, extras, by_slot = [], [], [], {}, {}
    order, slot_plan = source_plan(spec, list(chain_ids))
    for local, i in enumerate(chain_ids):
        base = 0 if spec["mode"] == "replicas" else i     # "single" / "partitions": entities numbered across the whole model
        if spec.get("shared_sink"):      # one collector behind every server: its lists are in global processing order
            shared = getattr(build_chains, "_shared", None)
            if local == 0 or shared is None:
                shared = build_chains._shared = Sink("sink")
            sink = shared
        else:
            sink = Sink(f"sink{i}") if spec.get("downstream", True) else None
        if svc[i] == "exp":
            if spec["rng"] == "philox":
                st = PhiloxExponentialLatency(mean[i], hs.Stream(seed, base, hs.STREAM_SERVICE))
            else:
                st = ExponentialLatency(mean[i])
        else:
            st = ConstantLatency(mean[i])
        server = Server(f"srv{i}", concurrency=conc[i], service_time=st, queue_capacity=qcap[i], downstream=sink)
        # The chain's Sources in slot order (slot = position among the chain's Sources in `sources=[...]`): slot 0 draws from
        # the chain's ARRIVAL stream, slot j >= 1 from stream base (1 << 40) | (base << 2) | (j - 1)
        # (include/hs_engine.h `src_more_kind`).  Several Sources feeding one Server are entities of their own.
        made = []
        for slot, (sa, sr, is_first) in enumerate(slot_plan[i]):
            stop_instant = None if stop is None else Instant.from_seconds(stop)
            if spec["rng"] != "philox":
                factory = Source.poisson if sa == "poisson" else Source.constant
                made.append(factory(rate=sr, target=server, name=f"src{i}" + ("" if is_first else f"_{slot}"), stop_after=stop))
                continue
            prof = make_profile(i) if is_first else ConstantRateProfile(rate=sr)
            if sa == "poisson":
                sbase = base if slot == 0 else (1 << 40) | (base << 2) | (slot - 1)
                prov = PhiloxPoissonArrival(prof, Instant.Epoch, hs.Stream(seed, sbase, hs.STREAM_ARRIVAL))
            else:
                prov = ConstantArrivalTimeProvider(prof, start_time=Instant.Epoch)
            made.append(Source(f"src{i}" + ("" if is_first else f"_{slot}"), SimpleEventProvider(server, "Request", stop_instant), prov))
        by_slot[i] = made
        source = made[0] if made else None
        extras[i] = made[1:]
        entities.append(server)
        if sink is not None and not any(e is sink for e in entities):
            entities.append(sink)
        handles.append((source, server, sink))
    sources = [by_slot[c][slot] for c, slot in order]
    build_chains._extras = extras
    return sources, entities, handles


def classify(ev, node_of):
    et = ev.event_type
    tgt = ev.target
    if et == "source_event":
        return (EV["probe_tick"] if isinstance(tgt, Probe) else EV["source"]), node_of[id(tgt)]
    if et == "probe_event":
        return EV["probe"], node_of.get(id(tgt), -1)
    if et == "QUEUE_NOTIFY":
        return EV["notify"], node_of[id(tgt)]
    if et == "QUEUE_POLL":
        return EV["poll"], node_of[id(tgt)]
    if et == "QUEUE_DELIVER":
        return EV["deliver"], node_of[id(tgt)]
    if isinstance(tgt, _QueuedResourceWorkerAdapter):
        return (EV["continuation"] if isinstance(ev, ProcessContinuation) else EV["work"]), node_of[id(tgt)]
    if isinstance(tgt, Server):
        return EV["enqueue"], node_of[id(tgt)]
    if isinstance(tgt, Sink):
        return EV["sink"], node_of[id(tgt)]
    if isinstance(tgt, RandomRouter):
        return EV["route"], node_of[id(tgt)]
    if isinstance(tgt, LoadBalancer):
        return (EV["lb_resp"] if et == "_lb_response" else EV["lb"]), node_of[id(tgt)]
    if isinstance(tgt, NetworkLink):
        return (EV["link_cont"] if isinstance(ev, ProcessContinuation) else EV["link"]), node_of[id(tgt)]
    raise RuntimeError(f"unclassified event {ev!r}")



def run_sim_windows(sim, spec):
    """`sim.run()`, or -- spec["windows"] = [t_1, ..., t_k] seconds -- the reference's own windowed execution (core/simulation.py:527-541
    `_run_window`, what ParallelSimulation's coordinator drives) to every t_i and then to end_s; the summary as run() builds it."""
    w = spec.get("windows")
    if not w:
        return sim.run()
    for t in list(w) + [spec["end_s"]]:
        sim._run_window(Instant.from_seconds(t))
    sim._is_running = False
    return sim._build_summary()

def run_sim(spec, chain_ids, seed, want_trace):
    if spec["rng"] == "mt":
        random.seed(seed)
        np.random.seed(seed)
    sources, entities, handles = build_chains(spec, chain_ids, seed)
    probes, probe_data = [], {}
    for local, c in enumerate(chain_ids):
        prs = (spec.get("probes") or [None] * spec["n_chains"])[c]
        if prs is None:
            continue
        if not isinstance(prs[0], (list, tuple)):
            prs = [prs]                # one probe; a list of [metric, interval] pairs = several probes on one chain
        for j, pr in enumerate(prs):
            who, attr = PROBE_METRICS[pr[0]]
            target = {"source": handles[local][0], "server": handles[local][1], "sink": handles[local][2]}[who]
            if spec.get("probe_start_s") is not None:      # Probe(start_time=...): Source.start() overwrites it (load/source.py:127)
                from happysimulator.instrumentation.data import Data as _Data
                data = _Data()
                probe = Probe(target, attr, data, interval=pr[1], start_time=Instant.from_seconds(spec["probe_start_s"]))
            else:
                probe, data = Probe.on(target, attr, interval=pr[1])
            data._ns = []              # Data keeps seconds; keep the exact nanoseconds beside it

            def add_stat(value, time, _orig=data.add_stat, _d=data):
                _d._ns.append((time.nanoseconds, value))
                _orig(value, time)

            data.add_stat = add_stat
            probes.append((c, probe))
            probe_data[(c, j)] = data
    sim = Simulation(start_time=Instant.from_seconds(spec.get("start_s", 0)) if spec.get("start_s") else None,
                     end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=entities,
                     probes=[p for _, p in probes])
    sim._probe_data = probe_data
    # chain-local node numbering: chain c -> (source=c, server=c, sink=c); kinds disambiguate
    node_of = {}
    for local, (src, srv, snk) in enumerate(handles):
        c = chain_ids[local]
        if src is not None:
            node_of[id(src)] = c
        for x in build_chains._extras.get(c, []):
            node_of[id(x)] = c
        node_of[id(srv)] = c
        node_of[id(srv._queue)] = c
        node_of[id(srv._driver)] = c
        node_of[id(srv._worker)] = c
        if snk is not None:
            node_of[id(snk)] = c
    for c, probe in probes:
        node_of[id(probe)] = c
    trace = []
    if want_trace:
        heap = sim._event_heap
        orig_pop = heap.pop

        cb_chain = {}                      # CallbackEntity of a probe_event -> chain (resolved through its closure)
        for c, probe in probes:
            cb_chain[id(probe._event_provider.data_sink)] = c

        def pop():
            e = orig_pop()
            k, nd = classify(e, node_of)
            if k == EV["probe"]:
                fn = e.target._fn if hasattr(e.target, "_fn") else e.target.fn
                cells = {id(cell.cell_contents) for cell in (fn.__closure__ or ())}
                nd = next(c for key, c in cb_chain.items() if key in cells)
            trace.append((e.time.nanoseconds, k, nd, e._sort_index))
            return e

        heap.pop = pop
    # Simulation.schedule(): one-off Requests built by the caller AFTER the Simulation (core/simulation.py:195-206)
    for c, t_s in spec.get("schedule") or []:
        if c in chain_ids:
            sim.schedule(Event(time=Instant.from_seconds(t_s), event_type="Request",
                               target=handles[chain_ids.index(c)][1]))
    summary = run_sim_windows(sim, spec)
    return sim, summary, handles, trace


def run_case(spec):
    n = spec["n_chains"]
    want_trace = spec.get("trace", False)
    out = {}
    stats = {k: np.zeros(n, np.int64) for k in
             ("generated", "accepted", "dropped", "completed", "rejected", "received", "depth", "active")}
    total_service = np.zeros(n, np.float64)
    gen_more = np.zeros((3, n), np.int64)
    sink_t, sink_lat, sink_off = [], [], [0]
    traces = []
    totals, finals, durations = [], [], []
    if spec["mode"] == "single":
        groups = [(list(range(n)), spec["seed"])]
    else:  # replicas: one Simulation per chain, seed = base_seed + i (parallel/runner.py:115-142)
        groups = [([i], spec["seed"] + i) for i in range(n)]
    probe_all = {}
    for chain_ids, seed in groups:
        sim, summary, handles, trace = run_sim(spec, chain_ids, seed, want_trace)
        probe_all.update(sim._probe_data)
        totals.append(summary.total_events_processed)
        finals.append(sim._current_time.nanoseconds)
        durations.append(summary.duration_s)
        traces.extend(trace)
        for local, (src, srv, snk) in enumerate(handles):
            i = chain_ids[local]
            stats["generated"][i] = src.generated_count if src is not None else 0
            for j, x in enumerate(build_chains._extras.get(i, [])):
                gen_more[j, i] = x.generated_count
            stats["accepted"][i] = srv.stats_accepted
            stats["dropped"][i] = srv.stats_dropped
            stats["completed"][i] = srv._requests_completed
            stats["rejected"][i] = srv._requests_rejected
            stats["depth"][i] = srv.depth
            stats["active"][i] = srv.active_requests
            total_service[i] = srv._total_service_time
            if snk is not None and not (spec.get("shared_sink") and local > 0):
                stats["received"][i] = snk.events_received
    # sink records in chain order (groups are in chain order too)
        for local, (src, srv, snk) in enumerate(handles):
            if snk is not None and not (spec.get("shared_sink") and local > 0):   # a shared Sink: all records under chain 0
                sink_t.extend(t.nanoseconds for t in snk.completion_times)
                sink_lat.extend(snk.latencies_s)
            sink_off.append(len(sink_t))
    meta = dict(spec=spec, total_events=totals, final_ns=finals, duration_s=durations)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    if spec.get("probes"):
        pt, pv, poff = [], [], [0]
        slots = max([len(pr) if pr is not None and isinstance(pr[0], (list, tuple)) else 1 for pr in spec["probes"]])
        for i in range(n):             # offsets indexed chain * slots + slot (slots == 1: one entry per chain)
            for j in range(slots):
                d = probe_all.get((i, j))
                if d is not None:
                    pt.extend(t for t, _ in d._ns)
                    pv.extend(int(v) for _, v in d._ns)
                poff.append(len(pt))
        out["probe_slots"] = np.asarray([slots], np.int64)
        out["probe_t_ns"] = np.asarray(pt, np.int64)
        out["probe_v"] = np.asarray(pv, np.int64)
        out["probe_off"] = np.asarray(poff, np.int64)
    for k, v in stats.items():
        out[k] = v
    out["total_service_s"] = total_service
    if spec.get("more_sources"):
        out["generated_more"] = gen_more
    out["sink_t_ns"] = np.asarray(sink_t, np.int64)
    out["sink_latency_s"] = np.asarray(sink_lat, np.float64)
    out["sink_off"] = np.asarray(sink_off, np.int64)
    if want_trace:
        tr = np.asarray(traces, np.int64).reshape(-1, 4)
        out["trace"] = tr
    return out, meta


def run_tandem_case(spec):
    """Tandem queues with reference components only: per chain `Source -> Server -> Server -> ... [-> Sink]`, every Server built
    with `downstream=<the next Server>` (components/server/server.py:64-122,271-272).  Spec format, station numbering and stream
    bases: tests/tandem_specs.py (one station per Server, chain-major; station i's entities draw from stream base i; the chain's
    Source belongs to its first station).  Trace nodes: the station index (a Source: its chain's first station, a Sink: its
    chain's last)."""
    for d in (os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))):
        if d not in sys.path:
            sys.path.insert(0, d)
    import tandem_specs as TS

    order, first = TS.station_index(spec)
    seed = spec["seed"]
    sources, entities, servers, sinks, node_of = [], [], {}, [], {}
    for c, ch in enumerate(spec["chains"]):
        n_st = len(ch["stages"])
        sink = Sink(f"sink{c}") if ch["sink"] else None
        nxt = sink
        for st in reversed(range(n_st)):
            sg = ch["stages"][st]
            lat = (PhiloxExponentialLatency(sg["mean"], hs.Stream(seed, first[c] + st, hs.STREAM_SERVICE)) if sg["svc"] == "exp"
                   else ConstantLatency(sg["mean"]))
            nxt = servers[(c, st)] = Server(f"srv{c}_{st}", concurrency=sg["conc"], service_time=lat, queue_capacity=sg["qcap"],
                                            downstream=nxt)
        prof = ConstantRateProfile(rate=ch["rate"])
        prov = (PhiloxPoissonArrival(prof, Instant.Epoch, hs.Stream(seed, first[c], hs.STREAM_ARRIVAL)) if ch["arr"] == "poisson"
                else ConstantArrivalTimeProvider(prof, start_time=Instant.Epoch))
        stop = None if ch["stop_after_s"] is None else Instant.from_seconds(ch["stop_after_s"])
        src = Source(f"src{c}", SimpleEventProvider(servers[(c, 0)], "Request", stop), prov)
        sources.append(src)
        node_of[id(src)] = first[c]
        for st in range(n_st):
            sv = servers[(c, st)]
            entities.append(sv)
            for x in (sv, sv._queue, sv._driver, sv._worker):
                node_of[id(x)] = first[c] + st
        sinks.append(sink)
        if sink is not None:
            entities.append(sink)
            node_of[id(sink)] = first[c] + n_st - 1
    # spec["probes"] = [((chain, stage), metric, interval_s)]: Probes on Servers of the chains, in `probes=[...]` order;
    # spec["sched"] = [((chain, stage), t_s)]: Simulation.schedule(Event(t, "Request", target=that Server)) calls, in call order
    # (tests/tandem_specs.py tandem_probe_case)
    probes, probe_data = [], []
    for cs, metric, iv in spec.get("probes") or []:
        _who, attr = PROBE_METRICS[metric]
        probe, data = Probe.on(servers[tuple(cs)], attr, interval=iv)
        data._ns = []

        def add_stat(value, time, _orig=data.add_stat, _d=data):
            _d._ns.append((time.nanoseconds, value))
            _orig(value, time)

        data.add_stat = add_stat
        probes.append(probe)
        probe_data.append(data)
        node_of[id(probe)] = first[cs[0]] + cs[1]
    sim = Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=entities, probes=probes)
    cb_station = {id(p._event_provider.data_sink): node_of[id(p)] for p in probes}
    trace = []
    heap = sim._event_heap
    orig_pop = heap.pop

    def pop():
        e = orig_pop()
        k, nd = classify(e, node_of)
        if k == EV["probe"]:
            fn = e.target._fn if hasattr(e.target, "_fn") else e.target.fn
            cells = {id(cell.cell_contents) for cell in (fn.__closure__ or ())}
            nd = next(c for key, c in cb_station.items() if key in cells)
        trace.append((e.time.nanoseconds, k, nd, e._sort_index))
        return e

    heap.pop = pop
    for cs, t_s in spec.get("sched") or []:
        sim.schedule(Event(time=Instant.from_seconds(t_s), event_type="Request", target=servers[tuple(cs)]))
    summary = sim.run()
    n = len(order)
    out = {k: np.zeros(n, np.int64) for k in ("accepted", "dropped", "completed", "rejected", "depth", "active")}
    out["total_service_s"] = np.zeros(n, np.float64)
    for i, (c, st) in enumerate(order):
        sv = servers[(c, st)]
        out["accepted"][i], out["dropped"][i] = sv.stats_accepted, sv.stats_dropped
        out["completed"][i], out["rejected"][i] = sv._requests_completed, sv._requests_rejected
        out["depth"][i], out["active"][i] = sv.depth, sv.active_requests
        out["total_service_s"][i] = sv._total_service_time
    out["generated"] = np.asarray([s.generated_count for s in sources], np.int64)
    sink_t, sink_lat, sink_off = [], [], [0]
    for sk in sinks:
        if sk is not None:
            sink_t.extend(t.nanoseconds for t in sk.completion_times)
            sink_lat.extend(sk.latencies_s)
        sink_off.append(len(sink_t))
    out["sink_t_ns"] = np.asarray(sink_t, np.int64)
    out["sink_latency_s"] = np.asarray(sink_lat, np.float64)
    out["sink_off"] = np.asarray(sink_off, np.int64)
    out["trace"] = np.asarray(trace, np.int64).reshape(-1, 4)
    if probes:
        pt, pv, poff = [], [], [0]
        for d in probe_data:
            pt.extend(t for t, _ in d._ns)
            pv.extend(int(v) for _, v in d._ns)
            poff.append(len(pt))
        out["probe_t_ns"] = np.asarray(pt, np.int64)
        out["probe_v"] = np.asarray(pv, np.int64)
        out["probe_off"] = np.asarray(poff, np.int64)
    meta = dict(spec=spec, total_events=[summary.total_events_processed], final_ns=[sim._current_time.nanoseconds],
                duration_s=[summary.duration_s])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out, meta


def run_fan_in_case(case):
    """A forest of Servers with reference components only (tests/tandem_specs.py fan_in_case): Server i forwards to Server
    case["down"][i] (`downstream=`; several Servers may forward to one), Sources on some of them, a Sink behind some roots.
    Station i = Server i; its entities draw from stream base i.  Returns per-Server statistics, the Sinks' records and the full
    processed-event trace (node = the Server's index; a Source / Sink: its Server's)."""
    sv, down, seed = case["servers"], case["down"], case["seed"]
    n = len(sv)
    servers, sinks, node_of = [None] * n, {}, {}
    for i in reversed(range(n)):                     # forwards go to later Servers: build from the back
        s = sv[i]
        if down[i] >= 0:
            nxt = servers[down[i]]
        elif s["sink"]:
            nxt = sinks[i] = Sink(f"sink{i}")
        else:
            nxt = None
        lat = (PhiloxExponentialLatency(s["mean"], hs.Stream(seed, i, hs.STREAM_SERVICE)) if s["svc"] == "exp" else ConstantLatency(s["mean"]))
        servers[i] = Server(f"srv{i}", concurrency=s["conc"], service_time=lat, queue_capacity=s["qcap"], downstream=nxt)
    sources, src_of = [], {}
    for i, s in enumerate(sv):
        if s["src"] is None:
            continue
        prof = ConstantRateProfile(rate=s["src"][1])
        prov = (PhiloxPoissonArrival(prof, Instant.Epoch, hs.Stream(seed, i, hs.STREAM_ARRIVAL)) if s["src"][0] == "poisson"
                else ConstantArrivalTimeProvider(prof, start_time=Instant.Epoch))
        src = Source(f"src{i}", SimpleEventProvider(servers[i], "Request", None), prov)
        sources.append(src)
        src_of[i] = src
        node_of[id(src)] = i
    entities = []
    for i in range(n):
        entities.append(servers[i])
        for x in (servers[i], servers[i]._queue, servers[i]._driver, servers[i]._worker):
            node_of[id(x)] = i
        if i in sinks:
            entities.append(sinks[i])
            node_of[id(sinks[i])] = i
    sim = Simulation(end_time=Instant.from_seconds(case["end_s"]), sources=sources, entities=entities)
    trace = []
    heap = sim._event_heap
    orig_pop = heap.pop

    def pop():
        e = orig_pop()
        k, nd = classify(e, node_of)
        trace.append((e.time.nanoseconds, k, nd, e._sort_index))
        return e

    heap.pop = pop
    summary = sim.run()
    out = {k: np.zeros(n, np.int64) for k in ("accepted", "dropped", "completed", "rejected", "depth", "active", "generated")}
    out["total_service_s"] = np.zeros(n, np.float64)
    for i in range(n):
        x = servers[i]
        out["accepted"][i], out["dropped"][i] = x.stats_accepted, x.stats_dropped
        out["completed"][i], out["rejected"][i] = x._requests_completed, x._requests_rejected
        out["depth"][i], out["active"][i] = x.depth, x.active_requests
        out["total_service_s"][i] = x._total_service_time
        out["generated"][i] = src_of[i].generated_count if i in src_of else 0
    out["sinks"] = {i: (np.asarray([t.nanoseconds for t in sk.completion_times], np.int64), np.asarray(sk.latencies_s, np.float64))
                    for i, sk in sinks.items()}
    out["trace"] = np.asarray(trace, np.int64).reshape(-1, 4)
    out["total_events"] = summary.total_events_processed
    out["final_ns"] = sim._current_time.nanoseconds
    return out


def run_ring_case(spec):
    """N stations on a ring, reference components only:
    Source.poisson(ext_rate) -> Server_i(Exp mean) -> RandomRouter_i([Sink_i, Link_i]);
    Link_i = NetworkLink(latency=ConstantLatency(lat_min), jitter=Exp(jitter_mean), egress=Server_{i+1})."""
    n, seed = spec["n"], spec["seed"]
    import happysimulator.components.network.link as link_mod

    link_mod.random = _PerLinkRandom       # the loss test of every link reads that link's own Philox stream
    sinks = [Sink(f"sink{i}") for i in range(n)]
    servers = [Server(f"srv{i}", concurrency=spec.get("concurrency", 1),
                      service_time=PhiloxExponentialLatency(spec["mean"], hs.Stream(seed, i, hs.STREAM_SERVICE)),
                      queue_capacity=spec.get("queue_cap")) for i in range(n)]
    links, routers, sources, first_profile = [], [], [], {}
    for i in range(n):
        jit = None
        jk = spec.get("jitter_kind", "exp")
        jk = jk[i] if isinstance(jk, list) else jk
        jm = spec.get("jitter_mean")
        jm = jm[i] if isinstance(jm, list) else jm
        if jm is not None and jk == "exp":
            jit = PhiloxExponentialLatency(jm, hs.Stream(seed, i, hs.STREAM_LINK))
        elif jm is not None and jk == "const":       # the reference's own class: no random numbers (conditions.py:60-63 uses it)
            jit = ConstantLatency(jm)
        loss = spec.get("loss", 0.0)
        links.append(NetworkLink(f"link{i}", latency=ConstantLatency(spec["lat_min"]), jitter=jit,
                                 bandwidth_bps=spec.get("bandwidth_bps"),
                                 packet_loss_rate=loss[i] if isinstance(loss, list) else loss,
                                 egress=servers[(i + 1) % n]))
        links[i]._loss_stream = hs.Stream(seed, i, hs.STREAM_LOSS)
        # spec["rt_pattern"][i]: the router's target list, 's' = the station's Sink, 'l' = its link (default "sl")
        pat = (spec.get("rt_pattern") or ["sl"] * n)[i]
        routers.append(PhiloxRandomRouter(f"router{i}", targets=[sinks[i] if ch == "s" else links[i] for ch in pat],
                                          stream=hs.Stream(seed, i, hs.STREAM_ROUTE)))
        servers[i].downstream = routers[i]
        rate = spec["ext_rate"][i] if isinstance(spec["ext_rate"], list) else spec["ext_rate"]
        pr = (spec.get("profile") or [None] * n)[i]
        if rate > 0:
            if pr is None:
                profile = ConstantRateProfile(rate=rate)
            elif pr[0] == "ramp":
                profile = LinearRampProfile(duration_s=pr[1], start_rate=pr[2], end_rate=pr[3])
            else:
                profile = SpikeProfile(baseline_rate=pr[1], spike_rate=pr[2], warmup_s=pr[3], spike_duration_s=pr[4])
            first_profile[i] = profile
        sources.append(None)
    # the stations' Sources in `sources=[...]` order; slot = position among the station's Sources (slot 0: the station's ARRIVAL
    # stream, slot j >= 1: stream base (1 << 40) | (i << 2) | (j - 1)); `more_sources` = further Sources feeding the same Server
    order, slot_plan = ring_source_plan(spec)
    by_slot = {i: [] for i in range(n)}
    for i in range(n):
        for slot, (sa, sr, is_first) in enumerate(slot_plan[i]):
            prof = first_profile[i] if is_first else ConstantRateProfile(rate=sr)
            if sa == "poisson":
                sbase = i if slot == 0 else (1 << 40) | (i << 2) | (slot - 1)
                prov = PhiloxPoissonArrival(prof, Instant.Epoch, hs.Stream(seed, sbase, hs.STREAM_ARRIVAL))
            else:
                prov = ConstantArrivalTimeProvider(prof, start_time=Instant.Epoch)
            by_slot[i].append(Source(f"src{i}_{slot}", SimpleEventProvider(servers[i], "Request", None), prov))
        sources[i] = by_slot[i][0] if by_slot[i] else None
    listed = [by_slot[i][slot] for i, slot in order]
    probes, probe_data = [], {}
    for i, prs in enumerate(spec.get("probes") or []):
        if prs is None:
            continue
        if not isinstance(prs[0], (list, tuple)):
            prs = [prs]                # one probe; a list of [metric, interval] pairs = several probes on one station
        for j, pr in enumerate(prs):
            who, attr = PROBE_METRICS[pr[0]]
            target = {"source": sources[i], "server": servers[i], "sink": sinks[i]}[who]
            probe, data = Probe.on(target, attr, interval=pr[1])
            data._ns = []

            def add_stat(value, time, _orig=data.add_stat, _d=data):
                _d._ns.append((time.nanoseconds, value))
                _orig(value, time)

            data.add_stat = add_stat
            probes.append((i, probe))
            probe_data[(i, j)] = data
    sim = Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=listed,
                     entities=servers + routers + links + sinks, probes=[p for _, p in probes])
    node_of = {}
    for i in range(n):
        for obj in (*by_slot[i], servers[i], servers[i]._queue, servers[i]._driver, servers[i]._worker, routers[i],
                    links[i], sinks[i]):
            if obj is not None:
                node_of[id(obj)] = i
    for i, probe in probes:
        node_of[id(probe)] = i
    trace = []
    if spec.get("trace"):
        heap = sim._event_heap
        orig_pop = heap.pop
        cb_station = {id(probe._event_provider.data_sink): i for i, probe in probes}

        def pop():
            e = orig_pop()
            k, nd = classify(e, node_of)
            if k == EV["probe"]:
                fn = e.target._fn if hasattr(e.target, "_fn") else e.target.fn
                cells = {id(cell.cell_contents) for cell in (fn.__closure__ or ())}
                nd = next(c for key, c in cb_station.items() if key in cells)
            trace.append((e.time.nanoseconds, k, nd, e._sort_index))
            return e

        heap.pop = pop
    for i, t_s in spec.get("schedule") or []:          # Simulation.schedule(): Requests for station i's Server, before run()
        sim.schedule(Event(time=Instant.from_seconds(t_s), event_type="Request", target=servers[i]))
    summary = run_sim_windows(sim, spec)
    out = {}
    meta = dict(spec=spec, total_events=[summary.total_events_processed], final_ns=[sim._current_time.nanoseconds],
                duration_s=[summary.duration_s])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    out["generated"] = np.array([s.generated_count if s is not None else 0 for s in sources], np.int64)
    if spec.get("more_sources"):
        out["generated_more"] = np.array([[by_slot[i][j + 1].generated_count if len(by_slot[i]) > j + 1 else 0 for i in range(n)]
                                          for j in range(3)], np.int64)
    out["accepted"] = np.array([s.stats_accepted for s in servers], np.int64)
    out["dropped"] = np.array([s.stats_dropped for s in servers], np.int64)
    out["completed"] = np.array([s._requests_completed for s in servers], np.int64)
    out["rejected"] = np.array([s._requests_rejected for s in servers], np.int64)
    out["depth"] = np.array([s.depth for s in servers], np.int64)
    out["active"] = np.array([s.active_requests for s in servers], np.int64)
    out["total_service_s"] = np.array([s._total_service_time for s in servers], np.float64)
    out["received"] = np.array([k.events_received for k in sinks], np.int64)
    out["routed"] = np.array([r.stats_routed for r in routers], np.int64)
    out["packets_sent"] = np.array([l.packets_sent for l in links], np.int64)
    if probes:
        pt, pv, poff = [], [], [0]
        slots = max([len(pr) if pr is not None and isinstance(pr[0], (list, tuple)) else 1 for pr in spec["probes"]])
        for i in range(n):             # offsets indexed station * slots + slot
            for j in range(slots):
                d = probe_data.get((i, j))
                if d is not None:
                    pt.extend(t for t, _ in d._ns)
                    pv.extend(int(v) for _, v in d._ns)
                poff.append(len(pt))
        out["probe_slots"] = np.asarray([slots], np.int64)
        out["probe_t_ns"] = np.asarray(pt, np.int64)
        out["probe_v"] = np.asarray(pv, np.int64)
        out["probe_off"] = np.asarray(poff, np.int64)
    out["packets_dropped"] = np.array([l.packets_dropped for l in links], np.int64)
    out["bytes_transmitted"] = np.array([l.bytes_transmitted for l in links], np.int64)
    sink_t, sink_lat, off = [], [], [0]
    for k in sinks:
        sink_t.extend(t.nanoseconds for t in k.completion_times)
        sink_lat.extend(k.latencies_s)
        off.append(len(sink_t))
    out["sink_t_ns"] = np.asarray(sink_t, np.int64)
    out["sink_latency_s"] = np.asarray(sink_lat, np.float64)
    out["sink_off"] = np.asarray(off, np.int64)
    if spec.get("trace"):
        out["trace"] = np.asarray(trace, np.int64).reshape(-1, 4)
    return out, meta


def run_graph_case(spec):
    """An arbitrary graph of the lowered entity set, reference components only (round 4: the oracle's ground for what the engines
    still refuse -- routers with more than four targets or several upstreams, a NetworkLink with several senders, a Server behind a
    Server inside a link network, more than four Sources per Server).  spec:
        n_sinks; servers = [{mean, c, cap, out}], links = [{lat, jk, jm, loss, to}], routers = [{targets}], sources = [{kind, rate, to}]
    with `out` / a router target = ["sink", j] | ["link", l] | ["router", r] | ["server", s] (`out` may be None), `to` a Server index.
    Streams: Source k ARRIVAL base k, Server s SERVICE base s, link l LINK / LOSS base l, router r ROUTE base r."""
    seed = spec["seed"]
    import happysimulator.components.network.link as link_mod

    link_mod.random = _PerLinkRandom
    sinks = [Sink(f"sink{j}") for j in range(spec["n_sinks"])]
    servers = [Server(f"srv{i}", concurrency=sv.get("c", 1),
                      service_time=PhiloxExponentialLatency(sv["mean"], hs.Stream(seed, i, hs.STREAM_SERVICE)),
                      queue_capacity=sv.get("cap")) for i, sv in enumerate(spec["servers"])]
    links = []
    for l, lk in enumerate(spec["links"]):
        jit = None
        if lk.get("jm") is not None and lk.get("jk") == "exp":
            jit = PhiloxExponentialLatency(lk["jm"], hs.Stream(seed, l, hs.STREAM_LINK))
        elif lk.get("jm") is not None and lk.get("jk") == "const":
            jit = ConstantLatency(lk["jm"])
        links.append(NetworkLink(f"link{l}", latency=ConstantLatency(lk["lat"]), jitter=jit, packet_loss_rate=lk.get("loss", 0.0),
                                 egress=servers[lk["to"]]))
        links[l]._loss_stream = hs.Stream(seed, l, hs.STREAM_LOSS)
    routers = [None] * len(spec["routers"])
    # LoadBalancers anywhere a Request can go (round 6): [{strategy: "chash" | "round_robin" | "random", vnodes, backends: [server ...]}]
    lbs = []
    for j, lb in enumerate(spec.get("lbs") or []):
        backends = [servers[b] for b in lb["backends"]]
        if lb["strategy"] == "round_robin":
            from happysimulator.components.load_balancer.strategies import RoundRobin
            strat = RoundRobin()
        elif lb["strategy"] == "random":
            import happysimulator.components.load_balancer.strategies as strat_mod
            strat_mod.random = _PerRequestChoice
            strat = strat_mod.Random()
        else:
            strat = ConsistentHash(virtual_nodes=lb["vnodes"])
        lbs.append(LoadBalancer(f"lb{j}", backends=backends, strategy=strat))

    def entity(ref):
        kind, idx = ref
        return {"sink": sinks, "link": links, "router": routers, "server": servers, "lb": lbs}[kind][idx]

    # routers may target routers: build them in an order in which every router's router-targets exist (the generator keeps them acyclic)
    pending = list(range(len(routers)))
    while pending:
        progressed = False
        for r in list(pending):
            tg = spec["routers"][r]["targets"]
            if all(k != "router" or routers[i] is not None for k, i in tg):
                routers[r] = PhiloxRandomRouter(f"router{r}", targets=[entity(t) for t in tg], stream=hs.Stream(seed, r, hs.STREAM_ROUTE))
                pending.remove(r)
                progressed = True
        assert progressed, "router targets form a cycle of routers"
    for i, sv in enumerate(spec["servers"]):
        if sv.get("out") is not None:
            servers[i].downstream = entity(sv["out"])
    sources = []
    for k, sc in enumerate(spec["sources"]):
        prof = ConstantRateProfile(rate=sc["rate"])
        prov = (PhiloxPoissonArrival(prof, Instant.Epoch, hs.Stream(seed, k, hs.STREAM_ARRIVAL)) if sc["kind"] == "poisson"
                else ConstantArrivalTimeProvider(prof, start_time=Instant.Epoch))
        to = servers[sc["to"]] if isinstance(sc["to"], int) else entity(sc["to"])
        if sc.get("n_clients"):                     # a Request with a client id (ConsistentHash's key / the Random strategy's draw)
            ep = PhiloxClientProvider(to, sc["n_clients"], hs.Stream(seed, k, hs.STREAM_KEY))
        else:
            ep = SimpleEventProvider(to, "Request", None)
        sources.append(Source(f"src{k}", ep, prov))
    sim = Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=servers + lbs + routers + links + sinks)
    for ref, t_s in spec.get("schedule") or []:      # Simulation.schedule() before run(), in this order (core/simulation.py:195-206)
        sim.schedule(Event(time=Instant.from_seconds(t_s), event_type="Request", target=entity(ref)))
    summary = sim.run()
    out = {}
    meta = dict(spec=spec, total_events=[summary.total_events_processed], final_ns=[sim._current_time.nanoseconds],
                duration_s=[summary.duration_s])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    out["generated"] = np.array([s.generated_count for s in sources], np.int64)
    out["accepted"] = np.array([s.stats_accepted for s in servers], np.int64)
    out["dropped"] = np.array([s.stats_dropped for s in servers], np.int64)
    out["completed"] = np.array([s._requests_completed for s in servers], np.int64)
    out["rejected"] = np.array([s._requests_rejected for s in servers], np.int64)
    out["depth"] = np.array([s.depth for s in servers], np.int64)
    out["active"] = np.array([s.active_requests for s in servers], np.int64)
    out["total_service_s"] = np.array([s._total_service_time for s in servers], np.float64)
    out["received"] = np.array([k.events_received for k in sinks], np.int64)
    out["routed"] = np.array([r.stats_routed for r in routers], np.int64)
    out["packets_sent"] = np.array([l.packets_sent for l in links], np.int64)
    out["packets_dropped"] = np.array([l.packets_dropped for l in links], np.int64)
    if lbs:
        out["lb_stats"] = np.array([[lb.stats.requests_received, lb.stats.requests_forwarded, lb.stats.requests_failed,
                                     lb.stats.no_backend_available, len(lb._in_flight)] for lb in lbs], np.int64)
        tot, toff = [], [0]
        for j, lb in enumerate(lbs):
            tot.extend(lb.get_backend_info(servers[b]).total_requests for b in spec["lbs"][j]["backends"])
            toff.append(len(tot))
        out["lb_backend_total_requests"] = np.asarray(tot, np.int64)
        out["lb_backend_off"] = np.asarray(toff, np.int64)
        # the selections of the strategy's RoundRobin: RoundRobin._index; ConsistentHash: its key-less fallback's; Random: -1
        out["lb_rr_index"] = np.array([lb.strategy._fallback._index if isinstance(lb.strategy, ConsistentHash) else
                                       getattr(lb.strategy, "_index", -1) for lb in lbs], np.int64)
    sink_t, sink_lat, off = [], [], [0]
    for k in sinks:
        sink_t.extend(t.nanoseconds for t in k.completion_times)
        sink_lat.extend(k.latencies_s)
        off.append(len(sink_t))
    out["sink_t_ns"] = np.asarray(sink_t, np.int64)
    out["sink_latency_s"] = np.asarray(sink_lat, np.float64)
    out["sink_off"] = np.asarray(off, np.int64)
    return out, meta


def run_lb_case(spec):
    """BASELINE configs[4] in miniature, reference components only (examples/visual/chash_example.py wiring):
    S x Source(Poisson rate_i, PhiloxClientProvider) -> LoadBalancer(ConsistentHash(vnodes)) -> B x Server(c, Exp mean,
    queue_cap) -> Sink (one shared Sink, or one per backend).  Node numbering of the trace = the oracle graph's:
    sources 0..S-1, LB = S, backends S+1..S+B, sinks after that."""
    S, B, seed = spec["n_sources"], spec["n_backends"], spec["seed"]
    shared = spec.get("shared_sink", True)
    sinks = [Sink("sink")] if shared else [Sink(f"sink{j}") for j in range(B)]
    conc = _per_chain(spec.get("concurrency", 1), B)
    qcap = _per_chain(spec.get("queue_cap"), B)
    mean = _per_chain(spec["mean"], B)
    servers = [Server(f"srv{j}", concurrency=conc[j],
                      service_time=PhiloxExponentialLatency(mean[j], hs.Stream(seed, S + j, hs.STREAM_SERVICE)),
                      queue_capacity=qcap[j], downstream=sinks[0] if shared else sinks[j]) for j in range(B)]
    strategy = spec.get("strategy", "chash")
    if strategy == "round_robin":              # the LoadBalancer's DEFAULT strategy (load_balancer.py:112, strategies.py:50-73)
        from happysimulator.components.load_balancer.strategies import RoundRobin
        lb = LoadBalancer("lb", backends=servers, strategy=RoundRobin())
    elif strategy == "random":                 # strategies.py:137-150, its random.choice plugged per Request (_PerRequestChoice)
        import happysimulator.components.load_balancer.strategies as strat_mod
        strat_mod.random = _PerRequestChoice
        lb = LoadBalancer("lb", backends=servers, strategy=strat_mod.Random())
        spec["n_clients"] = B
    else:
        lb = LoadBalancer("lb", backends=servers, strategy=ConsistentHash(virtual_nodes=spec["vnodes"]))
    rate = _per_chain(spec["rate"], S)
    stop = spec.get("stop_after_s")
    stop_instant = None if stop is None else Instant.from_seconds(stop)
    sources = []
    profiles = spec.get("profile") or [None] * S              # per source: None | ["ramp", d, s, e] | ["spike", b, s, w, d]
    for i in range(S):
        pr = profiles[i]
        profile = (ConstantRateProfile(rate=rate[i]) if pr is None else
                   LinearRampProfile(duration_s=pr[1], start_rate=pr[2], end_rate=pr[3]) if pr[0] == "ramp" else
                   SpikeProfile(baseline_rate=pr[1], spike_rate=pr[2], warmup_s=pr[3], spike_duration_s=pr[4]))
        prov = PhiloxPoissonArrival(profile, Instant.Epoch, hs.Stream(seed, i, hs.STREAM_ARRIVAL))
        ep = PhiloxClientProvider(lb, spec["n_clients"], hs.Stream(seed, i, hs.STREAM_KEY), stop_instant)
        sources.append(Source(f"src{i}", ep, prov))
    # probes on backend Servers / Sinks: [["server" | "sink", index, metric, interval], ...] in `probes=[...]` order
    probes, probe_data = [], []
    for who, idx, metric, interval in spec.get("probes") or []:
        probe, data = Probe.on({"server": servers, "sink": sinks, "source": sources}[who][idx], PROBE_METRICS[metric][1],
                               interval=interval)
        data._ns = []

        def add_stat(value, time, _orig=data.add_stat, _d=data):
            _d._ns.append((time.nanoseconds, value))
            _orig(value, time)

        data.add_stat = add_stat
        probes.append(probe)
        probe_data.append(data)
    sim = Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=[lb, *servers, *sinks],
                     probes=probes)
    node_of = {id(lb): S}
    for j, probe in enumerate(probes):
        node_of[id(probe)] = S + 1 + B + len(sinks) + j
    for i, src in enumerate(sources):
        node_of[id(src)] = i
    for j, srv in enumerate(servers):
        for obj in (srv, srv._queue, srv._driver, srv._worker):
            node_of[id(obj)] = S + 1 + j
    for j, k in enumerate(sinks):
        node_of[id(k)] = S + 1 + B + j
    trace = []
    if spec.get("trace"):
        heap = sim._event_heap
        orig_pop = heap.pop

        cb_probe = {id(probe._event_provider.data_sink): S + 1 + B + len(sinks) + j for j, probe in enumerate(probes)}

        def pop():
            e = orig_pop()
            k, nd = classify(e, node_of)
            if k == EV["probe"]:
                fn = e.target._fn if hasattr(e.target, "_fn") else e.target.fn
                cells = {id(cell.cell_contents) for cell in (fn.__closure__ or ())}
                nd = next(c for key, c in cb_probe.items() if key in cells)
            trace.append((e.time.nanoseconds, k, nd, e._sort_index))
            return e

        heap.pop = pop
    summary = run_sim_windows(sim, spec)
    out = {}
    if probes:
        pt, pv, poff = [], [], [0]
        for d in probe_data:
            pt.extend(t for t, _ in d._ns)
            pv.extend(int(v) for _, v in d._ns)
            poff.append(len(pt))
        out["probe_t_ns"], out["probe_v"], out["probe_off"] = (np.asarray(pt, np.int64), np.asarray(pv, np.int64),
                                                                np.asarray(poff, np.int64))
    meta = dict(spec=spec, total_events=[summary.total_events_processed], final_ns=[sim._current_time.nanoseconds],
                duration_s=[summary.duration_s])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    out["generated"] = np.array([s.generated_count for s in sources], np.int64)
    out["accepted"] = np.array([s.stats_accepted for s in servers], np.int64)
    out["dropped"] = np.array([s.stats_dropped for s in servers], np.int64)
    out["completed"] = np.array([s._requests_completed for s in servers], np.int64)
    out["rejected"] = np.array([s._requests_rejected for s in servers], np.int64)
    out["depth"] = np.array([s.depth for s in servers], np.int64)
    out["active"] = np.array([s.active_requests for s in servers], np.int64)
    out["total_service_s"] = np.array([s._total_service_time for s in servers], np.float64)
    out["received"] = np.array([k.events_received for k in sinks], np.int64)
    st = lb.stats
    out["lb_stats"] = np.array([st.requests_received, st.requests_forwarded, st.requests_failed,
                                st.no_backend_available, len(lb._in_flight)], np.int64)
    out["backend_total_requests"] = np.array([lb.get_backend_info(s).total_requests for s in servers], np.int64)
    if strategy == "chash":
        # the ring itself (hash is a 128-bit int: split) and the client -> backend map the strategy implements
        ring = lb.strategy._ring
        out["ring_hash_hi"] = np.array([h >> 64 for h, _ in ring], np.uint64)
        out["ring_hash_lo"] = np.array([h & ((1 << 64) - 1) for h, _ in ring], np.uint64)
        out["ring_backend"] = np.array([int(nm[3:]) for _, nm in ring], np.int32)
        probe = []
        for cid in range(min(spec["n_clients"], 4096)):
            ev = Event(time=Instant.Epoch, event_type="Request", target=lb, context={"metadata": {"client_id": str(cid)}})
            probe.append(int(lb.strategy.select(servers, ev).name[3:]))
        out["client_backend"] = np.array(probe, np.int32)
    elif strategy == "round_robin":
        out["rr_index"] = np.array([lb.strategy._index], np.int64)
    sink_t, sink_lat, off = [], [], [0]
    for k in sinks:
        sink_t.extend(t.nanoseconds for t in k.completion_times)
        sink_lat.extend(k.latencies_s)
        off.append(len(sink_t))
    out["sink_t_ns"] = np.asarray(sink_t, np.int64)
    out["sink_latency_s"] = np.asarray(sink_lat, np.float64)
    out["sink_off"] = np.asarray(off, np.int64)
    if spec.get("trace"):
        out["trace"] = np.asarray(trace, np.int64).reshape(-1, 4)
    return out, meta


def _stage(svc, mean, conc=1, qcap=None):
    return dict(svc=svc, mean=mean, conc=conc, qcap=qcap)


# Tandem queues (tests/tandem_specs.py): Philox streams, full traces.
TANDEM_CASES = [
    dict(name="tandem_2stage_philox", topology="tandem", end_s=20.0, seed=42, chains=[
        dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True, stages=[_stage("exp", 0.1), _stage("exp", 0.08)])]),
    dict(name="tandem_lock_step_consts", topology="tandem", end_s=3.0, seed=7, chains=[
        dict(arr="constant", rate=10.0, stop_after_s=None, sink=True, stages=[_stage("const", 0.1)] * 4),
        dict(arr="constant", rate=10.0, stop_after_s=None, sink=True, stages=[_stage("const", 0.1), _stage("const", 0.0), _stage("const", 0.05, 2)]),
        dict(arr="constant", rate=50.0, stop_after_s=None, sink=True,
             stages=[_stage("const", 0.1, 3), _stage("const", 0.1, 1, 1), _stage("const", 0.0, 4), _stage("const", 0.05, 4)]),
        dict(arr="constant", rate=10.0, stop_after_s=1.0, sink=False, stages=[_stage("const", 0.2, 2, 1), _stage("const", 0.1, 2, 1)])]),
    dict(name="tandem_4stage_mixed", topology="tandem", end_s=10.0, seed=2026, chains=[
        dict(arr="poisson", rate=12.0, stop_after_s=None, sink=True,
             stages=[_stage("exp", 0.05, 2), _stage("const", 0.06), _stage("exp", 0.1, 3, 4), _stage("exp", 0.02)]),
        dict(arr="constant", rate=16.0, stop_after_s=6.0, sink=True, stages=[_stage("exp", 0.05), _stage("exp", 0.05, 1, 2)]),
        dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True, stages=[_stage("exp", 0.1)]),
        dict(arr="poisson", rate=20.0, stop_after_s=None, sink=False, stages=[_stage("exp", 0.03), _stage("exp", 0.04), _stage("const", 0.01)])]),
]

LB_CASES = [
    # round 5: the load-balancer graph driven window by window on the reference (see ring_6_windows)
    dict(name="lb_windows", topology="lb", n_sources=3, n_backends=5, rate=[12.0, 9.0, 7.0], mean=0.1, concurrency=[1, 2, 1, 1, 3],
         vnodes=60, n_clients=900, windows=[0.5, 2.0, 2.0, 1.0, 6.5], end_s=10.0, seed=321, trace=False),
    # the chash_example wiring: 1 source, 3 backends with concurrency 3, 150 vnodes, 200 clients
    dict(name="lb_chash_example", topology="lb", n_sources=1, n_backends=3, rate=30.0, mean=0.1, concurrency=3,
         vnodes=150, n_clients=200, end_s=20.0, seed=42, trace=True),
    dict(name="lb_4src_8be", topology="lb", n_sources=4, n_backends=8, rate=[12.0, 9.0, 15.0, 6.0], mean=0.1,
         vnodes=150, n_clients=5000, end_s=20.0, seed=7, trace=True),
    # probes on a load-balancer graph: backend Servers (every Server metric) and the shared Sink
    dict(name="lb_probes", topology="lb", n_sources=3, n_backends=5, rate=[14.0, 10.0, 8.0], mean=0.12, concurrency=[1, 2, 1, 3, 1],
         queue_cap=[None, None, 2, None, None], vnodes=40, n_clients=3000, end_s=12.0, seed=19,
         probes=[["server", 0, "depth", 0.25], ["server", 1, "active_requests", 0.1], ["server", 2, "stats_dropped", 0.5],
                 ["server", 2, "stats_accepted", 0.5], ["server", 3, "requests_completed", 0.3], ["sink", 0, "events_received", 0.2],
                 ["server", 4, "depth", 0.7], ["source", 1, "generated_count", 0.4]], trace=True),
    dict(name="lb_probes_per_backend_sinks", topology="lb", n_sources=2, n_backends=4, rate=[18.0, 9.0], mean=0.1,
         concurrency=[1, 1, 2, 1], vnodes=30, n_clients=500, end_s=10.0, seed=23, shared_sink=False,
         probes=[["sink", 2, "events_received", 0.25], ["server", 1, "depth", 0.3], ["sink", 0, "events_received", 1.0]], trace=True),
    # Source.with_profile in front of the LoadBalancer: a ramp, a spike and a constant source; a probe on the ramp's Source
    dict(name="lb_profiles", topology="lb", n_sources=3, n_backends=6, rate=[16.0, 30.0, 9.0], mean=0.1, concurrency=[1, 2, 1, 1, 3, 1],
         vnodes=60, n_clients=2000, end_s=10.0, seed=29,
         profile=[["ramp", 6.0, 4.0, 16.0], ["spike", 5.0, 30.0, 3.0, 1.5], None],
         probes=[["source", 0, "generated_count", 0.5], ["server", 4, "depth", 0.25]], trace=True),
    dict(name="lb_cap2_overload", topology="lb", n_sources=3, n_backends=4, rate=20.0, mean=0.1, concurrency=1,
         queue_cap=2, vnodes=20, n_clients=1000, end_s=15.0, seed=11, trace=True),
    dict(name="lb_per_backend_sinks", topology="lb", n_sources=6, n_backends=16, rate=16.0, mean=0.1,
         concurrency=[1, 2] * 8, vnodes=100, n_clients=100000, end_s=15.0, seed=3, shared_sink=False, trace=True),
    dict(name="lb_stop_after", topology="lb", n_sources=2, n_backends=4, rate=20.0, mean=0.1, vnodes=50,
         n_clients=777, stop_after_s=6.0, end_s=12.0, seed=5, trace=True),
    dict(name="lb_64src_256be", topology="lb", n_sources=64, n_backends=256, rate=24.0, mean=0.1, vnodes=150,
         n_clients=65536, end_s=8.0, seed=2026, trace=False),
    # the LoadBalancer's DEFAULT strategy, RoundRobin (backends[_index % len] in the LB's processing order), and Random
    dict(name="lb_rr_3src_5be", topology="lb", strategy="round_robin", n_sources=3, n_backends=5, rate=[14.0, 9.0, 11.0], mean=0.12,
         concurrency=[1, 2, 1, 1, 3], queue_cap=[None, None, 2, None, None], vnodes=1, n_clients=1, end_s=12.0, seed=31, trace=True),
    dict(name="lb_rr_per_backend_sinks", topology="lb", strategy="round_robin", n_sources=5, n_backends=7, rate=12.0, mean=0.1,
         vnodes=1, n_clients=1, end_s=10.0, seed=32, shared_sink=False, stop_after_s=7.0,
         probes=[["server", 2, "depth", 0.25], ["sink", 6, "events_received", 0.5]], trace=True),
    dict(name="lb_rr_64src_96be", topology="lb", strategy="round_robin", n_sources=64, n_backends=96, rate=20.0, mean=0.06,
         vnodes=1, n_clients=1, end_s=6.0, seed=33, trace=False),
    dict(name="lb_random_4src_6be", topology="lb", strategy="random", n_sources=4, n_backends=6, rate=[10.0, 16.0, 8.0, 12.0], mean=0.1,
         concurrency=[1, 1, 2, 1, 1, 1], vnodes=1, n_clients=6, end_s=12.0, seed=34, trace=True),
    dict(name="lb_random_32src_64be", topology="lb", strategy="random", n_sources=32, n_backends=64, rate=25.0, mean=0.07,
         vnodes=1, n_clients=64, end_s=6.0, seed=35, trace=False),
]

# arbitrary graphs (run_graph_case): what the engines still refuse, pinned for the oracle -- a router with eight targets among Sinks,
# links, Servers; links with several senders; Servers behind Servers next to links; six Sources on one Server
GRAPH_CASES = [
    dict(name="graph_fanout_8", topology="graph", n_sinks=2, end_s=8.0, seed=171,
         servers=[dict(mean=0.05, c=1, cap=None, out=["router", 0]), dict(mean=0.08, c=2, cap=3, out=["router", 0]),
                  dict(mean=0.04, c=1, cap=None, out=["link", 1]), dict(mean=0.1, c=4, cap=None, out=["sink", 1])],
         links=[dict(lat=0.001, jk="exp", jm=0.004, loss=0.0, to=2), dict(lat=0.002, jk="const", jm=0.0005, loss=0.1, to=3),
                dict(lat=0.0005, jk=None, jm=None, loss=0.0, to=0)],
         routers=[dict(targets=[["sink", 0], ["link", 0], ["server", 2], ["link", 1], ["sink", 1], ["link", 0], ["server", 3], ["link", 2]])],
         sources=[dict(kind="poisson", rate=7.0, to=0), dict(kind="poisson", rate=5.0, to=1), dict(kind="constant", rate=4.0, to=0)]),
    dict(name="graph_shared_links_tandem", topology="graph", n_sinks=3, end_s=10.0, seed=173,
         servers=[dict(mean=0.03, c=1, cap=None, out=["server", 1]), dict(mean=0.05, c=2, cap=4, out=["link", 0]),
                  dict(mean=0.06, c=1, cap=None, out=["link", 0]), dict(mean=0.04, c=1, cap=2, out=["router", 1]),
                  dict(mean=0.07, c=3, cap=None, out=["router", 0])],
         links=[dict(lat=0.002, jk="exp", jm=0.003, loss=0.05, to=3), dict(lat=0.001, jk=None, jm=None, loss=0.0, to=4)],
         routers=[dict(targets=[["sink", 0], ["link", 1], ["sink", 2]]), dict(targets=[["link", 1], ["sink", 1], ["server", 4], ["link", 0]])],
         sources=[dict(kind="poisson", rate=6.0, to=0), dict(kind="poisson", rate=4.0, to=2), dict(kind="constant", rate=2.0, to=0),
                  dict(kind="poisson", rate=3.0, to=0), dict(kind="constant", rate=2.0, to=0), dict(kind="poisson", rate=2.0, to=0),
                  dict(kind="poisson", rate=1.0, to=0)]),
    # round 6: LoadBalancers INSIDE a general graph -- three of them (ConsistentHash behind a Server and a router, RoundRobin behind a
    # router and a Source, Random behind its own Source), every Source handing out client ids
    dict(name="graph_three_load_balancers", topology="graph", n_sinks=2, end_s=8.0, seed=181,
         servers=[dict(mean=0.03, c=1, cap=None, out=["lb", 0]), dict(mean=0.05, c=2, cap=None, out=["router", 0]),
                  dict(mean=0.06, c=1, cap=3, out=["sink", 0]), dict(mean=0.04, c=1, cap=None, out=["link", 0]),
                  dict(mean=0.08, c=3, cap=None, out=["sink", 1]), dict(mean=0.05, c=1, cap=2, out=["sink", 0])],
         links=[dict(lat=0.002, jk="exp", jm=0.003, loss=0.05, to=4)],
         routers=[dict(targets=[["sink", 1], ["lb", 0], ["lb", 1], ["link", 0]])],
         lbs=[dict(strategy="chash", vnodes=17, backends=[2, 3, 5]), dict(strategy="round_robin", vnodes=0, backends=[4, 2]),
              dict(strategy="random", vnodes=0, backends=[0, 1, 3])],
         sources=[dict(kind="poisson", rate=8.0, to=0, n_clients=50), dict(kind="poisson", rate=6.0, to=1, n_clients=5),
                  dict(kind="constant", rate=4.0, to=["lb", 1], n_clients=1000), dict(kind="poisson", rate=9.0, to=["lb", 2], n_clients=3),
                  dict(kind="poisson", rate=5.0, to=["lb", 0], n_clients=1000)]),
    # ... and key-less: two RoundRobin LoadBalancers sharing a backend, Requests `schedule()`d for Servers, a router, a link and the
    # LoadBalancers themselves (bursts on one nanosecond, on a Source's constant tick, at and beyond the end)
    dict(name="graph_round_robin_schedule", topology="graph", n_sinks=2, end_s=6.0, seed=183,
         servers=[dict(mean=0.04, c=1, cap=None, out=["lb", 1]), dict(mean=0.05, c=1, cap=2, out=["router", 0]),
                  dict(mean=0.03, c=2, cap=None, out=["sink", 0]), dict(mean=0.07, c=1, cap=None, out=["sink", 1])],
         links=[dict(lat=0.001, jk="const", jm=0.0005, loss=0.0, to=3)],
         routers=[dict(targets=[["sink", 0], ["link", 0], ["lb", 1]])],
         lbs=[dict(strategy="round_robin", vnodes=0, backends=[0, 1, 2]), dict(strategy="round_robin", vnodes=0, backends=[3, 2])],
         sources=[dict(kind="poisson", rate=9.0, to=["lb", 0]), dict(kind="constant", rate=2.0, to=["lb", 0]), dict(kind="poisson", rate=3.0, to=1)],
         schedule=[[["lb", 0], 0.5], [["lb", 0], 0.5], [["server", 1], 0.5], [["lb", 1], 1.0], [["router", 0], 1.0], [["link", 0], 0.0],
                   [["lb", 0], 0.0], [["lb", 1], 6.0], [["server", 0], 6.0], [["lb", 0], 6.5], [["lb", 1], 2.25]]),
    # ... and Requests WITHOUT a key at ConsistentHash LoadBalancers -- plain Sources next to client-keyed ones, `schedule()`d Requests:
    # the strategy's own fallback RoundRobin (strategies.py:362,420-421), advanced by the key-less Requests only
    dict(name="graph_keyless_consistent_hash", topology="graph", n_sinks=2, end_s=6.0, seed=185,
         servers=[dict(mean=0.04, c=1, cap=None, out=["lb", 1]), dict(mean=0.05, c=2, cap=None, out=["sink", 0]),
                  dict(mean=0.03, c=1, cap=4, out=["sink", 0]), dict(mean=0.06, c=1, cap=None, out=["sink", 1]),
                  dict(mean=0.05, c=1, cap=None, out=["router", 0])],
         links=[dict(lat=0.001, jk="exp", jm=0.002, loss=0.0, to=3)],
         routers=[dict(targets=[["sink", 1], ["lb", 1], ["link", 0]])],
         lbs=[dict(strategy="chash", vnodes=7, backends=[0, 1, 4]), dict(strategy="chash", vnodes=100, backends=[2, 3, 1])],
         sources=[dict(kind="poisson", rate=7.0, to=["lb", 0], n_clients=40), dict(kind="poisson", rate=6.0, to=["lb", 0]),
                  dict(kind="constant", rate=3.0, to=["lb", 1]), dict(kind="poisson", rate=4.0, to=0, n_clients=9),
                  dict(kind="poisson", rate=3.0, to=4)],
         schedule=[[["lb", 0], 0.5], [["lb", 1], 0.5], [["lb", 0], 0.5], [["server", 0], 1.0], [["router", 0], 2.0], [["lb", 1], 6.0],
                   [["lb", 0], 6.25]]),
]

RING_CASES = [
    # round 5: the reference driven WINDOW BY WINDOW (`_run_window`, core/simulation.py:527-541: growing ends, a repeated end, an
    # earlier end, an end a nanosecond behind the previous one) -- what it leaves is what ONE run to end_s leaves (run_sim_windows)
    dict(name="ring_6_windows", topology="ring", n=6, ext_rate=[6.0, 4.0, 0.0, 7.0, 5.0, 3.0], mean=0.08, lat_min=0.002, jitter_mean=0.005,
         windows=[1.0, 1.000000001, 3.5, 3.5, 2.0, 7.25], end_s=9.0, seed=123, trace=False),
    # Simulation.schedule() on networked stations: bursts at one timestamp, a station without a Source, one beyond the end
    dict(name="ring_4_schedule", topology="ring", n=4, ext_rate=[5.0, 0.0, 7.0, 4.0], mean=0.08, lat_min=0.002,
         jitter_mean=0.004,
         schedule=[[1, 0.5], [1, 0.5], [0, 2.000000003], [1, 0.5], [3, 7.25], [2, 1.0000001], [1, 3.3], [0, 9.5], [2, 10.4]],
         end_s=10.0, seed=81, trace=True),
    # time-varying arrival profiles on networked stations (ramp up, ramp down to a trickle, a spike) + one probe
    dict(name="ring_5_profiles", topology="ring", n=5, ext_rate=[6.0, 30.0, 4.0, 5.0, 40.0], mean=0.06, lat_min=0.002,
         jitter_mean=0.005,
         profile=[["ramp", 8.0, 1.0, 14.0], ["ramp", 6.0, 30.0, 2.0], None, ["spike", 3.0, 40.0, 4.0, 2.0], None],
         probes=[None, None, ["depth", 0.5], None, None], end_s=12.0, seed=71, trace=True),
    # probes on networked stations (depth / active / counters sampled between the messages and the local events)
    dict(name="ring_6_probes", topology="ring", n=6, ext_rate=[8.0, 5.0, 9.0, 0.0, 6.0, 7.0], mean=0.09, concurrency=1,
         queue_cap=None, lat_min=0.002, jitter_mean=0.006,
         probes=[["depth", 0.25], ["active_requests", 0.1], None, ["stats_accepted", 0.5], ["events_received", 0.3],
                 ["requests_completed", 0.4]],
         end_s=12.0, seed=61, trace=True),
    dict(name="ring_5_multi_probes", topology="ring", n=5, ext_rate=[8.0, 5.0, 0.0, 6.0, 7.0], mean=0.09, concurrency=2,
         queue_cap=4, lat_min=0.002, jitter_mean=0.006,
         probes=[[["depth", 0.25], ["active_requests", 0.25], ["stats_dropped", 0.5], ["events_received", 0.2]], None,
                 [["stats_accepted", 0.5], ["requests_completed", 0.5]], [["depth", 0.3]], None],
         end_s=10.0, seed=63, trace=True),
    # several Sources feeding one Server of a network (entities of their own), two `sources=[...]` orders
    dict(name="ring_5_multi_source", topology="ring", n=5, ext_rate=[6.0, 0.0, 5.0, 4.0, 0.0], mean=0.07, concurrency=1,
         queue_cap=None, lat_min=0.002, jitter_mean=0.006,
         more_sources=[[["constant", 4.0]], None, [["poisson", 3.0], ["constant", 2.0]], [["constant", 4.0], ["poisson", 2.0], ["constant", 1.0]], None],
         probes=[["depth", 0.25], None, None, ["stats_accepted", 0.5], None], end_s=10.0, seed=65, trace=True),
    dict(name="ring_4_multi_source_order", topology="ring", n=4, ext_rate=[5.0, 4.0, 0.0, 6.0], mean=0.08, concurrency=2,
         queue_cap=3, lat_min=0.001, jitter_mean=None,
         more_sources=[[["constant", 5.0], ["constant", 5.0]], [["poisson", 4.0]], None, [["constant", 3.0]]],
         sources_order="extras_first", schedule=[[2, 0.2], [0, 0.2], [0, 0.4]], end_s=8.0, seed=66, trace=True),
    # jitter = ConstantLatency (the reference's datacenter_network preset: 0.5 ms + 0.1 ms, components/network/conditions.py:60-63):
    # a constant on top of the base latency, no random numbers; and a ring mixing the three kinds of jitter with loss
    dict(name="ring_5_const_jitter", topology="ring", n=5, ext_rate=[9.0, 6.0, 8.0, 0.0, 7.0], mean=0.07, lat_min=0.0005,
         jitter_mean=0.0001, jitter_kind="const", end_s=10.0, seed=91, trace=True),
    dict(name="ring_6_mixed_jitter", topology="ring", n=6, ext_rate=[7.0, 5.0, 8.0, 6.0, 0.0, 9.0], mean=0.08, concurrency=2,
         queue_cap=5, lat_min=0.002, jitter_mean=[0.004, 0.0013, None, 0.00025, 0.006, 0.0000007],
         jitter_kind=["exp", "const", None, "const", "exp", "const"], loss=[0.0, 0.1, 0.0, 0.3, 0.05, 0.0],
         probes=[["depth", 0.25], None, ["active_requests", 0.3], None, None, ["stats_accepted", 0.5]], end_s=12.0, seed=93, trace=True),
    # NetworkLink(packet_loss_rate): lost packets vanish at the link (link.py:131-138)
    dict(name="ring_8_loss", topology="ring", n=8, ext_rate=6.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, loss=0.2,
         end_s=20.0, seed=42, trace=True),
    # per-link loss rates incl. 0 and 1, fixed latency, c = 2, and a bandwidth limit (no payload_size in the metadata of
    # SimpleEventProvider's requests => transmission time 0, bytes_transmitted 0: link.py:209-234)
    dict(name="ring_5_loss_mixed", topology="ring", n=5, ext_rate=[6.0, 3.0, 7.0, 2.0, 5.0], mean=0.08, concurrency=2,
         lat_min=0.002, jitter_mean=None, loss=[0.0, 0.5, 1.0, 0.1, 0.03], bandwidth_bps=1e6, end_s=15.0, seed=13,
         trace=True),
    dict(name="ring_6_router_k", topology="ring", n=6, ext_rate=[5.0, 6.0, 4.0, 7.0, 3.0, 5.0], mean=0.08, lat_min=0.002,
         jitter_mean=0.006, rt_pattern=["sls", "lss", "ssls", "l", "sl", "ssl"], end_s=12.0, seed=321, trace=True),
    dict(name="ring_8_s42", topology="ring", n=8, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=20.0,
         seed=42, trace=True),
    dict(name="ring_3_short_hops", topology="ring", n=3, ext_rate=4.0, mean=0.1, lat_min=0.0005, jitter_mean=0.002,
         end_s=15.0, seed=7, trace=True),
    dict(name="ring_5_const_link", topology="ring", n=5, ext_rate=[4.0, 0.0, 6.0, 2.0, 4.0], mean=0.08, lat_min=0.003,
         jitter_mean=None, end_s=15.0, seed=11, trace=True),
    dict(name="ring_6_c2_cap3", topology="ring", n=6, ext_rate=9.0, mean=0.1, concurrency=2, queue_cap=3, lat_min=0.001,
         jitter_mean=0.005, end_s=10.0, seed=3, trace=True),
    dict(name="ring_64_s2026", topology="ring", n=64, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=8.0,
         seed=2026, trace=False),
]

# =====================================================================================================================
# The reference's OWN `ParallelSimulation(...).run()` (parallel/simulation.py:164-223, parallel/coordinator.py:75-227)
# =====================================================================================================================
# SURVEY 8(a) row X2.  What the live class accepts, probed here (tests/test_oracle_live_reference.py pins each fact):
#   * a LINKED run refuses a library `Server` unless the partition also lists the Server's private queue / driver / worker
#     entities (`routing.py:52-60`: their events count as "not in this partition");
#   * a library `NetworkLink` cannot carry a request across partitions: it forwards at `self.now`, so the coordinator raises
#     "violates min_latency: delay=0" (`coordinator.py:213-219`); `PartitionLink(latency=<library distribution>)` raises
#     AttributeError (`coordinator.py:209` calls `.sample()`, which no LatencyDistribution has);
#   * what DOES run across a link is the pattern of the reference's own tests (`tests/integration/test_parallel_simulation.py:
#     21-38`, SURVEY 8(d) workload 3): an entity that returns `Event(time=self.now + delay, target=<entity over there>)`.
# `FutureForward` below is that pattern with NetworkLink's delay arithmetic (link.py:190-216): one event per hop where a
# NetworkLink has two (Request@Link and its continuation) -- so per-ENTITY statistics and Sink records of such a run are those of
# the same topology with NetworkLinks, and the totals differ by the continuations (`packets_sent`).
from happysimulator.components.common import Counter  # noqa: E402
from happysimulator.core.entity import Entity  # noqa: E402
from happysimulator.parallel import ParallelSimulation, PartitionLink, SimulationPartition  # noqa: E402


class FutureForward(Entity):
    """A hop that delivers in the future: `Event(time=now + delay)` with delay = ConstantLatency + optional (Philox-plugged)
    exponential jitter, each through `get_latency(now).to_seconds()` like NetworkLink._calculate_delay (link.py:190-216)."""

    def __init__(self, name, egress, latency, jitter=None):
        super().__init__(name)
        self.egress, self.latency, self.jitter = egress, latency, jitter
        self.entered = 0

    def handle_event(self, event):
        self.entered += 1
        delay = self.latency.get_latency(self.now).to_seconds()
        if self.jitter is not None:
            delay += self.jitter.get_latency(self.now).to_seconds()
        return [Event(time=self.now + max(0.0, delay), event_type=event.event_type, target=self.egress,
                      context=event.context.copy())]


def _server_parts(srv):
    """What a linked partition must list next to a library Server (see above)."""
    return [srv, srv._queue, srv._driver, srv._worker]


def _count_time_travel(sims):
    """Wrap every partition's heap pop: an event popped with time < the clock is what `_execute_until` drops as time travel
    (core/simulation.py:480-489)."""
    dropped = {name: 0 for name in sims}
    for name, sim in sims.items():
        heap, orig = sim._event_heap, sim._event_heap.pop

        def pop(_orig=orig, _sim=sim, _name=name):
            e = _orig()
            if e.time < _sim._current_time:
                dropped[_name] += 1
            return e

        heap.pop = pop
    return dropped


def _pipeline(spec, hop_kind):
    """stages[k] = list of per-lane Server descriptions of partition k; lane j flows stage 0 -> 1 -> ... over hops.
    Stream bases = station index in (stage-major, lane-minor) order; a hop draws its jitter from its SENDER's LINK stream."""
    seed, lanes, stages = spec["seed"], spec["lanes"], spec["stages"]
    servers, sinks, hops, sources = [], [], [], []
    for k, stage in enumerate(stages):
        row = []
        for j in range(lanes):
            base = k * lanes + j
            svc, mean = stage["svc"], stage["mean"][j] if isinstance(stage["mean"], list) else stage["mean"]
            st = (PhiloxExponentialLatency(mean, hs.Stream(seed, base, hs.STREAM_SERVICE)) if svc == "exp" else ConstantLatency(mean))
            row.append(Server(f"srv{k}_{j}", concurrency=stage.get("concurrency", 1), service_time=st,
                              queue_capacity=stage.get("queue_cap")))
        servers.append(row)
    for j in range(lanes):
        sinks.append(Sink(f"sink{j}"))
        servers[-1][j].downstream = sinks[j]
    for k in range(len(stages) - 1):
        row = []
        for j in range(lanes):
            base = k * lanes + j
            lat = ConstantLatency(spec["hop_latency"])
            jit = (PhiloxExponentialLatency(spec["hop_jitter"], hs.Stream(seed, base, hs.STREAM_LINK))
                   if spec.get("hop_jitter") else None)
            if hop_kind == "network":
                hop = NetworkLink(f"hop{k}_{j}", latency=lat, jitter=jit, egress=servers[k + 1][j])
            else:
                hop = FutureForward(f"hop{k}_{j}", servers[k + 1][j], lat, jit)
            servers[k][j].downstream = hop
            row.append(hop)
        hops.append(row)
    for j in range(lanes):
        rate = spec["rate"][j] if isinstance(spec["rate"], list) else spec["rate"]
        prov = PhiloxPoissonArrival(ConstantRateProfile(rate=rate), Instant.Epoch, hs.Stream(seed, j, hs.STREAM_ARRIVAL))
        sources.append(Source(f"src{j}", SimpleEventProvider(servers[0][j], "Request", None), prov))
    return sources, servers, hops, sinks


def _pipeline_results(servers, hops, sinks, sources):
    flat = [s for row in servers for s in row]
    out = {
        "generated": np.array([s.generated_count for s in sources], np.int64),
        "accepted": np.array([s.stats_accepted for s in flat], np.int64),
        "dropped": np.array([s.stats_dropped for s in flat], np.int64),
        "completed": np.array([s._requests_completed for s in flat], np.int64),
        "depth": np.array([s.depth for s in flat], np.int64),
        "active": np.array([s.active_requests for s in flat], np.int64),
        "total_service_s": np.array([s._total_service_time for s in flat], np.float64),
        "received": np.array([k.events_received for k in sinks], np.int64),
        "hop_entered": np.array([(h.entered if isinstance(h, FutureForward) else h.packets_sent + h.packets_dropped)
                                 for row in hops for h in row], np.int64),
    }
    t, lat, off = [], [], [0]
    for k in sinks:
        t.extend(x.nanoseconds for x in k.completion_times)
        lat.extend(k.latencies_s)
        off.append(len(t))
    out["sink_t_ns"], out["sink_latency_s"], out["sink_off"] = np.asarray(t, np.int64), np.asarray(lat, np.float64), np.asarray(off, np.int64)
    return out


def run_parallel_linked_case(spec):
    """A pipeline of partitions (one stage per partition) three ways:
    `windowed`   -- the reference's own ParallelSimulation(partitions, links=[PartitionLink ...]).run(), FutureForward hops;
    `seq_future` -- the same entities in ONE reference Simulation (what the windowed run claims to equal);
    `seq_network`-- the topology with library NetworkLinks in ONE reference Simulation: what the engine's linked partitions compute
                    (hs.ParallelSimulation == the single heap)."""
    end = Instant.from_seconds(spec["end_s"])
    out = {}
    meta = dict(spec=spec)
    # --- seq_network
    sources, servers, hops, sinks = _pipeline(spec, "network")
    sim = Simulation(end_time=end, sources=sources, entities=[s for r in servers for s in r] + [h for r in hops for h in r] + sinks)
    summ = sim.run()
    for k, v in _pipeline_results(servers, hops, sinks, sources).items():
        out["seqnet_" + k] = v
    meta["seq_network"] = dict(total_events=summ.total_events_processed, final_ns=sim._current_time.nanoseconds, duration_s=summ.duration_s,
                               packets_sent=[h.packets_sent for r in hops for h in r])
    # --- seq_future
    sources, servers, hops, sinks = _pipeline(spec, "future")
    sim = Simulation(end_time=end, sources=sources, entities=[s for r in servers for s in r] + [h for r in hops for h in r] + sinks)
    summ = sim.run()
    for k, v in _pipeline_results(servers, hops, sinks, sources).items():
        out["seqfut_" + k] = v
    meta["seq_future"] = dict(total_events=summ.total_events_processed, final_ns=sim._current_time.nanoseconds, duration_s=summ.duration_s)
    # --- windowed: the reference's coordinator
    sources, servers, hops, sinks = _pipeline(spec, "future")
    parts = []
    for k, row in enumerate(servers):
        ents = [x for s in row for x in _server_parts(s)] + (hops[k] if k < len(hops) else []) + (sinks if k == len(servers) - 1 else [])
        parts.append(SimulationPartition(name=f"P{k}", entities=ents, sources=sources if k == 0 else []))
    # `packet_loss` (round 6): PartitionLink(packet_loss=p_k) on the link out of partition k -- the coordinator drops the cross-partition
    # event at the exchange with ONE random.Random(coord_seed) (parallel/coordinator.py:68,204); the sequential twins above have no
    # coordinator and lose nothing
    ploss = spec.get("packet_loss") or [0.0] * (len(servers) - 1)
    links = [PartitionLink(f"P{k}", f"P{k + 1}", min_latency=spec["hop_latency"], packet_loss=ploss[k]) for k in range(len(servers) - 1)]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                      # the GIL warning
        ps = ParallelSimulation(parts, end_time=end, links=links, **({"seed": spec["coord_seed"]} if "coord_seed" in spec else {}))
    dropped = _count_time_travel(ps.simulations)
    summ = ps.run()
    for k, v in _pipeline_results(servers, hops, sinks, sources).items():
        out["win_" + k] = v
    meta["windowed"] = dict(
        total_events=summ.total_events_processed, duration_s=summ.duration_s, total_windows=summ.total_windows,
        total_cross_partition_events=summ.total_cross_partition_events, window_size_s=summ.window_size_s,
        partition_events={k: v.total_events_processed for k, v in summ.partitions.items()},
        partition_duration_s={k: v.duration_s for k, v in summ.partitions.items()},
        time_travel_drops=dropped)
    same = all(np.array_equal(out["win_" + k], out["seqfut_" + k]) for k in ("accepted", "completed", "received", "sink_t_ns"))
    meta["windowed_equals_sequential"] = bool(same)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out, meta


def run_parallel_independent_case(spec):
    """`ParallelSimulation(partitions).run()` WITHOUT links (parallel/simulation.py:170-195): one partition per chain of the spec
    (build_chains: stream base = chain id, the run's seed for all -- how hs.ParallelSimulation numbers its partitions), every
    partition a Simulation of its own on the reference's thread pool.  Also each sub-case of the reference's own tests
    (tests/integration/test_parallel_simulation.py:75-109,239-289): constant Sources feeding Counters."""
    out, meta = {}, dict(spec=spec)
    if spec.get("counters"):
        res = []
        for sub in spec["counters"]:
            counters = [Counter(f"counter{i}") for i in range(len(sub["rates"]))]
            srcs = [Source.constant(rate=r, target=c, event_type="Ping") for r, c in zip(sub["rates"], counters)]
            parts = [SimulationPartition(name=f"P{i}", entities=[c], sources=[s_]) for i, (c, s_) in enumerate(zip(counters, srcs))]
            kw = dict(links=[]) if sub.get("empty_links") else {}
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                summ = ParallelSimulation(parts, duration=sub["duration"], **kw).run()
            # the sequential twin (test_deterministic_equivalence): separate Simulations
            seq = []
            for r in sub["rates"]:
                c = Counter("c")
                Simulation(duration=sub["duration"], sources=[Source.constant(rate=r, target=c, event_type="Ping")], entities=[c]).run()
                seq.append(c.total)
            res.append(dict(totals=[c.total for c in counters], sequential_totals=seq, generated=[s_.generated_count for s_ in srcs],
                            total_events=summ.total_events_processed, duration_s=summ.duration_s,
                            partition_events=[summ.partitions[f"P{i}"].total_events_processed for i in range(len(parts))],
                            partition_duration_s=[summ.partitions[f"P{i}"].duration_s for i in range(len(parts))],
                            total_windows=summ.total_windows, total_cross_partition_events=summ.total_cross_partition_events,
                            events_per_second=summ.events_per_second,
                            entity_events_handled={k: v.events_handled for k, v in summ.entities.items()}))
        meta["counters"] = res
        out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        return out, meta
    n = spec["n_chains"]
    sources, entities, handles = build_chains(spec, list(range(n)), spec["seed"])
    parts = []
    for i, (src, srv, snk) in enumerate(handles):
        parts.append(SimulationPartition(name=f"P{i}", entities=[srv] + ([snk] if snk is not None else []),
                                         sources=[s_ for s_ in sources if s_._event_provider._target is srv]))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ps = ParallelSimulation(parts, end_time=Instant.from_seconds(spec["end_s"]))
    summ = ps.run()
    meta.update(total_events=[summ.partitions[f"P{i}"].total_events_processed for i in range(n)],
                duration_s=[summ.partitions[f"P{i}"].duration_s for i in range(n)],
                final_ns=[ps.simulations[f"P{i}"]._current_time.nanoseconds for i in range(n)],
                parallel=dict(total_events=summ.total_events_processed, duration_s=summ.duration_s,
                              events_per_second=summ.events_per_second, total_windows=summ.total_windows,
                              total_cross_partition_events=summ.total_cross_partition_events, window_size_s=summ.window_size_s,
                              entity_events_handled={k: v.events_handled for k, v in summ.entities.items()}))
    out["generated"] = np.array([h[0].generated_count for h in handles], np.int64)
    for key, attr in (("accepted", "stats_accepted"), ("dropped", "stats_dropped"), ("completed", "_requests_completed"),
                      ("rejected", "_requests_rejected"), ("depth", "depth"), ("active", "active_requests")):
        out[key] = np.array([getattr(h[1], attr) for h in handles], np.int64)
    out["total_service_s"] = np.array([h[1]._total_service_time for h in handles], np.float64)
    out["received"] = np.array([h[2].events_received if h[2] is not None else 0 for h in handles], np.int64)
    t, lat, off = [], [], [0]
    for h in handles:
        if h[2] is not None:
            t.extend(x.nanoseconds for x in h[2].completion_times)
            lat.extend(h[2].latencies_s)
        off.append(len(t))
    out["sink_t_ns"], out["sink_latency_s"], out["sink_off"] = np.asarray(t, np.int64), np.asarray(lat, np.float64), np.asarray(off, np.int64)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out, meta


PARALLEL_CASES = [
    # the reference's own known-answer tests of ParallelSimulation, verbatim configurations
    dict(name="parallel_ref_counters", kind="independent", counters=[
        dict(rates=[10, 10], duration=10.0),                    # test_two_independent_partitions: 100 / 100
        dict(rates=[5], duration=20.0),                         # test_single_partition_matches_simulation: 100
        dict(rates=[10], duration=5.0, empty_links=True),       # test_empty_links_independent: 50, no windows
        dict(rates=[10, 10], duration=10.0),                    # test_deterministic_equivalence (sequential twins recorded)
        dict(rates=[3, 7, 11], duration=7.3)]),
    # Philox-plugged M/M/c partitions, one chain each, through the reference's ParallelSimulation (threads)
    dict(name="parallel_philox_independent_6", kind="independent", n_chains=6, arr=["poisson", "constant", "poisson", "poisson", "constant", "poisson"],
         rate=[8.0, 10.0, 30.0, 12.0, 4.0, 9.0], svc=["exp", "exp", "exp", "const", "exp", "exp"], mean=[0.1, 0.08, 0.05, 0.06, 0.3, 0.1],
         concurrency=[1, 1, 2, 1, 2, 1], queue_cap=[None, None, 4, None, None, 2], end_s=15.0, rng="philox", seed=61, mode="partitions"),
    # linked partitions where the windowed run IS its sequential twin: downstream partitions never hold an event of their own
    # beyond the clock (zero service time), hops deliver in send order (no jitter) -- no overshoot can run ahead of an arrival
    dict(name="parallel_linked_pipeline", kind="linked", lanes=4, rate=[8.0, 5.0, 12.0, 3.0], seed=71, end_s=12.0, hop_latency=0.05,
         stages=[dict(svc="exp", mean=[0.1, 0.15, 0.06, 0.2]), dict(svc="const", mean=0.0)]),
    dict(name="parallel_linked_three_stages", kind="linked", lanes=3, rate=[6.0, 9.0, 4.0], seed=72, end_s=10.0, hop_latency=0.02,
         stages=[dict(svc="exp", mean=[0.1, 0.08, 0.2], concurrency=2, queue_cap=3), dict(svc="const", mean=0.0), dict(svc="const", mean=0.0)]),
    # failing by design: an ACTIVE downstream partition (its own service completions lie beyond the window end) and jittered hops
    # -- every window's one-event overshoot (core/simulation.py:472) runs a completion early, and the arrivals injected at the next
    # barrier lie behind the partition's clock: dropped as time travel (core/simulation.py:480-489).  The reference's windowed run
    # departs from its own sequential run; the engine's linked partitions equal the SEQUENTIAL one.
    dict(name="parallel_linked_hazard", kind="linked", lanes=3, rate=[9.0, 7.0, 11.0], seed=73, end_s=15.0, hop_latency=0.02,
         hop_jitter=0.01, stages=[dict(svc="exp", mean=[0.08, 0.1, 0.06]), dict(svc="exp", mean=[0.07, 0.09, 0.05])]),
    # round 6: PartitionLink(packet_loss=...) -- the coordinator's own random.Random(seed) decides at the exchange; four hops share
    # the lossy PartitionLink, so the draws interleave over them in the sending partition's processing order
    dict(name="parallel_linked_loss", kind="linked", lanes=4, rate=[8.0, 5.0, 12.0, 3.0], seed=74, end_s=12.0, hop_latency=0.05,
         packet_loss=[0.25], coord_seed=74, stages=[dict(svc="exp", mean=[0.1, 0.15, 0.06, 0.2]), dict(svc="const", mean=0.0)]),
    # ... three stages, the FIRST PartitionLink lossy: what the second partition forwards depends on what survived
    dict(name="parallel_linked_loss_three", kind="linked", lanes=3, rate=[6.0, 9.0, 4.0], seed=75, end_s=10.0, hop_latency=0.02,
         packet_loss=[0.4, 0.0], coord_seed=75,
         stages=[dict(svc="exp", mean=[0.1, 0.08, 0.2], concurrency=2, queue_cap=3), dict(svc="const", mean=0.0), dict(svc="const", mean=0.0)]),
]


CASES = [
    # round 5: station chains driven window by window on the reference (see ring_6_windows)
    dict(name="philox_windows_8", n_chains=8, arr="poisson", rate=8.0, svc="exp", mean=0.1, concurrency=[1, 2, 1, 1, 3, 1, 1, 2],
         queue_cap=[None, None, 3, None, None, 1, None, None], windows=[0.7, 2.2, 2.2, 1.5, 2.200000001, 9.0], end_s=12.0,
         rng="philox", seed=77, mode="single", trace=False),
    # --- Oracle-A: stock MT19937 streams -------------------------------------------------
    dict(name="quickstart_mt42", n_chains=1, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=60.0,
         rng="mt", seed=42, mode="single", trace=True),
    dict(name="quickstart_mt7_c2", n_chains=1, arr="poisson", rate=14.0, svc="exp", mean=0.1, concurrency=2,
         end_s=40.0, rng="mt", seed=7, mode="single", trace=True),
    dict(name="const_r8", n_chains=1, arr="constant", rate=8.0, svc="const", mean=0.1, end_s=60.0,
         rng="mt", seed=1, mode="single", trace=True),
    dict(name="const_r10", n_chains=1, arr="constant", rate=10.0, svc="const", mean=0.1, end_s=60.0,
         rng="mt", seed=1, mode="single", trace=True),
    dict(name="const_r12_overload", n_chains=1, arr="constant", rate=12.0, svc="const", mean=0.1, end_s=60.0,
         rng="mt", seed=1, mode="single", trace=True),
    dict(name="const_r25_cap3", n_chains=1, arr="constant", rate=25.0, svc="const", mean=0.1, queue_cap=3,
         end_s=20.0, rng="mt", seed=1, mode="single", trace=True),
    dict(name="const_r20_c2_cap2", n_chains=1, arr="constant", rate=20.0, svc="const", mean=0.25, concurrency=2,
         queue_cap=2, end_s=20.0, rng="mt", seed=1, mode="single", trace=True),
    dict(name="const_4chains_ties", n_chains=4, arr="constant", rate=[10.0, 10.0, 5.0, 20.0], svc="const",
         mean=[0.1, 0.05, 0.2, 0.05], end_s=10.0, rng="mt", seed=1, mode="single", trace=True),
    # --- Oracle-B: Philox-plugged reference ---------------------------------------------
    dict(name="philox_1chain_s42", n_chains=1, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=60.0,
         rng="philox", seed=42, mode="single", trace=True),
    dict(name="philox_16chains_single", n_chains=16, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=30.0,
         rng="philox", seed=42, mode="single", trace=True),
    dict(name="philox_16chains_replicas", n_chains=16, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=30.0,
         rng="philox", seed=42, mode="replicas", trace=False),
    dict(name="philox_c3_rho08", n_chains=4, arr="poisson", rate=24.0, svc="exp", mean=0.1, concurrency=3,
         end_s=30.0, rng="philox", seed=7, mode="single", trace=True),
    dict(name="philox_c24_c32", n_chains=3, arr=["poisson", "poisson", "constant"], rate=[200.0, 330.0, 250.0], svc=["exp", "exp", "const"],
         mean=0.1, concurrency=[24, 32, 20], queue_cap=[None, 5, None], end_s=6.0, rng="philox", seed=23, mode="single", trace=True),
    dict(name="philox_overload_cap4", n_chains=4, arr="poisson", rate=15.0, svc="exp", mean=0.1, queue_cap=4,
         end_s=30.0, rng="philox", seed=11, mode="single", trace=True),
    dict(name="philox_c2_cap1_overload", n_chains=3, arr="poisson", rate=30.0, svc="exp", mean=0.1, concurrency=2,
         queue_cap=1, end_s=20.0, rng="philox", seed=5, mode="single", trace=True),
    dict(name="philox_stop_after", n_chains=2, arr="poisson", rate=8.0, svc="exp", mean=0.1, stop_after_s=10.0,
         end_s=20.0, rng="philox", seed=3, mode="single", trace=True),
    dict(name="philox_mixed_8", n_chains=8, arr=["poisson", "constant"] * 4, rate=[8.0, 10.0, 3.0, 20.0, 12.0, 5.0, 9.5, 40.0],
         svc=["exp", "exp", "const", "const", "exp", "exp", "exp", "const"],
         mean=[0.1, 0.08, 0.2, 0.04, 0.05, 0.25, 0.1, 0.02], concurrency=[1, 1, 1, 1, 2, 2, 1, 1],
         end_s=20.0, rng="philox", seed=99, mode="single", trace=True),
    dict(name="philox_no_downstream", n_chains=2, arr="poisson", rate=8.0, svc="exp", mean=0.1, downstream=False,
         end_s=15.0, rng="philox", seed=8, mode="single", trace=True),
    # --- time-varying profiles: the general path of ArrivalTimeProvider.next_arrival_time ------
    dict(name="profile_ramp_poisson", n_chains=3, arr="poisson", rate=1.0, svc="exp", mean=[0.05, 0.03, 0.08],
         concurrency=[1, 2, 1], profile=[["ramp", 15.0, 5.0, 30.0], ["ramp", 10.0, 40.0, 4.0], ["ramp", 8.0, 0.0, 20.0]],
         end_s=25.0, rng="philox", seed=21, mode="single", trace=True),
    dict(name="profile_spike_const", n_chains=2, arr="constant", rate=1.0, svc="exp", mean=[0.02, 0.05],
         queue_cap=[None, 5], profile=[["spike", 10.0, 150.0, 3.0, 2.0], ["spike", 4.0, 60.0, 1.5, 4.0]],
         end_s=10.0, rng="philox", seed=22, mode="single", trace=True),
    dict(name="profile_mixed_4", n_chains=4, arr=["poisson", "constant", "poisson", "constant"], rate=[8.0, 1.0, 1.0, 1.0],
         svc=["exp", "const", "exp", "exp"], mean=[0.1, 0.04, 0.03, 0.06],
         profile=[None, ["ramp", 12.0, 20.0, 2.0], ["spike", 6.0, 90.0, 4.0, 3.0], ["ramp", 20.0, 2.0, 25.0]],
         end_s=16.0, rng="philox", seed=23, mode="single", trace=True),
    # --- one Sink behind several servers: completion_times / latencies_s in global processing order -------------
    dict(name="philox_shared_sink_6", n_chains=6, arr="poisson", rate=[8.0, 5.0, 12.0, 3.0, 9.0, 20.0], svc="exp",
         mean=[0.1, 0.15, 0.05, 0.2, 0.08, 0.04], concurrency=[1, 1, 2, 1, 1, 3], shared_sink=True,
         end_s=20.0, rng="philox", seed=51, mode="single", trace=True),
    # --- Simulation.schedule(): one-off Requests injected before run() --------------------------
    dict(name="schedule_only", n_chains=3, arr="poisson", rate=0.0, svc=["const", "exp", "exp"], mean=[0.5, 0.2, 0.05],
         concurrency=[1, 2, 1], queue_cap=[None, None, 2],
         schedule=[[0, 1.0], [0, 1.2], [0, 1.2], [1, 0.3], [2, 0.5], [1, 0.31], [1, 0.32], [2, 0.5], [2, 0.5], [2, 0.5],
                   [2, 0.5], [0, 7.25], [1, 4.0], [0, 30.0]],
         end_s=10.0, rng="philox", seed=41, mode="single", trace=True),
    dict(name="schedule_with_sources", n_chains=4, arr=["poisson", "constant", "poisson", "poisson"],
         rate=[8.0, 5.0, 0.0, 12.0], svc="exp", mean=[0.1, 0.15, 0.3, 0.06], concurrency=[1, 1, 1, 2],
         schedule=[[0, 2.000000001], [1, 3.3000001], [2, 1.0], [2, 1.1], [3, 0.7500003], [0, 2.000000001], [3, 9.1],
                   [1, 0.05], [2, 6.123456789], [0, 11.9999], [3, 12.5]],
         end_s=12.0, rng="philox", seed=42, mode="single", trace=True),
    # --- several Sources feeding one Server (each an entity of its own; load/source.py:142-180) --------------------------------
    dict(name="multi_source_4chains", n_chains=4, arr=["poisson", "constant", "poisson", "constant"], rate=[6.0, 4.0, 9.0, 5.0],
         svc="exp", mean=[0.06, 0.08, 0.05, 0.1], concurrency=[1, 2, 1, 1], queue_cap=[None, 3, None, None],
         more_sources=[[["poisson", 5.0]], [["constant", 4.0], ["poisson", 3.0], ["constant", 2.0]], None, [["constant", 5.0]]],
         end_s=12.0, rng="philox", seed=81, mode="single", trace=True),
    dict(name="multi_source_order", n_chains=3, arr=["constant", "poisson", "constant"], rate=[4.0, 7.0, 2.0],
         svc=["const", "exp", "exp"], mean=[0.05, 0.07, 0.2], concurrency=[1, 1, 2], queue_cap=[2, None, None],
         more_sources=[[["constant", 4.0], ["constant", 2.0]], [["poisson", 6.0]], [["constant", 2.0], ["constant", 1.0]]],
         sources_order="extras_first", probes=[["depth", 0.25], None, [["stats_accepted", 0.5], ["active_requests", 0.5]]],
         schedule=[[0, 0.25], [2, 0.5], [2, 0.5], [1, 1.0]], stop_after_s=8.0,
         end_s=10.0, rng="philox", seed=82, mode="single", trace=True),
    dict(name="multi_source_replicas", n_chains=5, arr=["poisson", "constant", "poisson", "constant", "poisson"], rate=[6.0, 4.0, 9.0, 5.0, 3.0],
         svc="exp", mean=[0.06, 0.08, 0.05, 0.1, 0.12], concurrency=[1, 2, 1, 1, 1],
         more_sources=[[["poisson", 5.0]], [["constant", 4.0], ["constant", 2.0]], None, [["constant", 5.0]], [["poisson", 2.0], ["poisson", 2.0]]],
         sources_order="extras_first", end_s=10.0, rng="philox", seed=83, mode="replicas"),
    # --- probes: Probe.on(target, metric, interval) = a daemon Source sampling an attribute --------
    dict(name="probe_depth_4chains", n_chains=4, arr="poisson", rate=[12.0, 9.0, 30.0, 8.0], svc="exp", mean=[0.1, 0.1, 0.05, 0.1],
         concurrency=[1, 1, 2, 1], queue_cap=[None, None, 6, None],
         probes=[["depth", 0.5], ["active_requests", 0.25], ["stats_dropped", 1.0], ["events_received", 0.3]],
         end_s=20.0, rng="philox", seed=31, mode="single", trace=True),
    # several probes on one station (reference: each Probe.on() is its own daemon Source, numbered in probes=[...] order)
    dict(name="probe_multi_4chains", n_chains=4, arr=["poisson", "constant", "poisson", "poisson"], rate=[12.0, 4.0, 30.0, 8.0],
         svc="exp", mean=[0.1, 0.1, 0.05, 0.1], concurrency=[1, 1, 2, 1], queue_cap=[None, None, 6, None],
         probes=[[["depth", 0.5], ["active_requests", 0.5], ["stats_accepted", 0.2]],
                 [["events_received", 0.25], ["depth", 0.25], ["generated_count", 1.0], ["requests_completed", 0.5]],
                 [["stats_dropped", 1.0]], None],
         end_s=15.0, rng="philox", seed=33, mode="single", trace=True),
    dict(name="probe_const_ties", n_chains=3, arr="constant", rate=[10.0, 20.0, 4.0], svc="const", mean=[0.1, 0.07, 0.2],
         probes=[["depth", 0.1], ["generated_count", 0.05], ["requests_completed", 0.25]],
         end_s=6.0, rng="philox", seed=32, mode="single", trace=True),
    dict(name="philox_256chains_single", n_chains=256, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=10.0,
         rng="philox", seed=2026, mode="single", trace=False),
    dict(name="philox_64chains_replicas", n_chains=64, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=20.0,
         rng="philox", seed=1000, mode="replicas", trace=False),
]


def main(argv):
    only = set(argv[1:])
    for spec in PARALLEL_CASES:
        if only and spec["name"] not in only:
            continue
        fn = run_parallel_linked_case if spec["kind"] == "linked" else run_parallel_independent_case
        out, meta = fn(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        brief = {k: meta[k] for k in ("windowed", "seq_future", "seq_network", "windowed_equals_sequential", "parallel") if k in meta}
        print(f"{spec['name']}: {brief} -> {os.path.getsize(path)} B")
    for spec in LB_CASES:
        if only and spec["name"] not in only:
            continue
        out, meta = run_lb_case(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{spec['name']}: events={sum(meta['total_events'])} final={meta['final_ns'][-1]} "
              f"sink_records={len(out['sink_t_ns'])} lb={out['lb_stats'].tolist()} -> {os.path.getsize(path)} B")
    for spec in TANDEM_CASES:
        if only and spec["name"] not in only:
            continue
        out, meta = run_tandem_case(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{spec['name']}: events={sum(meta['total_events'])} final={meta['final_ns'][-1]} "
              f"sink_records={len(out['sink_t_ns'])} -> {os.path.getsize(path)} B")
    for spec in GRAPH_CASES:
        if only and spec["name"] not in only:
            continue
        out, meta = run_graph_case(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{spec['name']}: events={sum(meta['total_events'])} final={meta['final_ns'][-1]} "
              f"sink_records={len(out['sink_t_ns'])} -> {os.path.getsize(path)} B")
    for spec in RING_CASES:
        if only and spec["name"] not in only:
            continue
        out, meta = run_ring_case(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{spec['name']}: events={sum(meta['total_events'])} final={meta['final_ns'][-1]} "
              f"sink_records={len(out['sink_t_ns'])} -> {os.path.getsize(path)} B")
    for spec in CASES:
        if only and spec["name"] not in only:
            continue
        out, meta = run_case(dict(spec))
        path = os.path.join(HERE, spec["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{spec['name']}: events={sum(meta['total_events'])} final={meta['final_ns'][-1]} "
              f"sink_records={len(out['sink_t_ns'])} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main(sys.argv)
