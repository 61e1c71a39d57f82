"""GPU: the reference-shaped Python API end to end (Simulation / Source / Server / Sink / ParallelRunner),
checked against the live-reference goldens.  These read like the reference's own integration tests
(tests/integration/core_simulation/test_simulation_basic_counter.py, tests/unit/test_ergonomic_api.py)."""
import numpy as np
import pytest

import happy_simulator_amd as hs
from oracle import hs_oracle as O
import helpers as H
from happy_simulator_amd import Instant

pytestmark = pytest.mark.gpu


def test_quick_start_matches_philox_plugged_reference():
    gold = H.Golden("philox_1chain_s42")
    sink = hs.Sink()
    server = hs.Server("srv", service_time=hs.ExponentialLatency(0.1), downstream=sink)
    source = hs.Source.poisson(rate=8, target=server)
    sim = hs.Simulation(end_time=Instant.from_seconds(60), sources=[source], entities=[server, sink], seed=42)
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0] == 3630
    assert summary.duration_s == gold.meta["duration_s"][0]
    assert summary.events_per_second == summary.total_events_processed / summary.duration_s
    assert source.generated_count == gold.generated[0]
    assert server.stats_accepted == gold.accepted[0] and server.stats_dropped == 0
    assert server.stats.requests_completed == gold.completed[0]
    assert server.stats.total_service_time == gold.total_service_s[0]
    assert sink.events_received == gold.received[0]
    assert sink.latencies_s == gold.sink_latency_s.tolist()
    assert [t.nanoseconds for t in sink.completion_times] == gold.sink_t_ns.tolist()
    es = summary.entities
    assert es["srv"].entity_type == "Server" and es["srv"].events_handled == 0      # attr sniffing, simulation.py:579-583
    assert es["srv"].queue_stats.total_accepted == gold.accepted[0] and es["srv"].queue_stats.peak_depth == 0
    assert es["Sink"].events_handled == sink.events_received
    st = sink.latency_stats()
    assert st["count"] == sink.events_received and st["min"] <= st["p50"] <= st["p99"] <= st["max"]
    assert "Events processed: 3630" in str(summary)


def test_basic_counter_overshoot_known_answer():
    """Reference: tests/integration/core_simulation/test_simulation_basic_counter.py:7-34."""
    counter = hs.Counter()
    source = hs.Source.constant(rate=1, target=counter)
    sim = hs.Simulation(start_time=Instant.Epoch, end_time=Instant.from_seconds(60), sources=[source],
                        entities=[counter])
    summary = sim.run()
    assert source.generated_count == 61     # the overshoot tick is processed ...
    assert counter.total == 60              # ... but its payload is not
    assert summary.duration_s == 61.0


@pytest.mark.parametrize("rate,name", [(8, "const_r8"), (10, "const_r10"), (12, "const_r12_overload")])
def test_constant_rate_goldens_through_the_api(rate, name):
    gold = H.Golden(name)
    sink = hs.Sink()
    server = hs.Server("srv", service_time=hs.ConstantLatency(0.1), downstream=sink)
    source = hs.Source.constant(rate=rate, target=server)
    summary = hs.Simulation(duration=60, sources=[source], entities=[server, sink]).run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    assert source.generated_count == gold.generated[0]
    assert sink.events_received == gold.received[0]
    assert server.stats_accepted == gold.accepted[0] and server.depth == gold.depth[0]
    assert server.stats.total_service_time == gold.total_service_s[0]
    assert sink.latencies_s == gold.sink_latency_s.tolist()


def test_sixteen_chains_in_one_simulation():
    gold = H.Golden("philox_16chains_single")
    chains = []
    for i in range(16):
        sink = hs.Sink(f"sink{i}")
        srv = hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sink)
        chains.append((hs.Source.poisson(rate=8, target=srv, name=f"src{i}"), srv, sink))
    summary = hs.Simulation(duration=30.0, sources=[c[0] for c in chains],
                            entities=[e for c in chains for e in c[1:]], seed=42).run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    for i, (src, srv, sink) in enumerate(chains):
        assert src.generated_count == gold.generated[i]
        assert srv.stats.requests_completed == gold.completed[i]
        t, lat = gold.sink_records(i)
        assert sink.latencies_s == lat.tolist()
        np.testing.assert_array_equal(sink.completion_ns, t)


def _build_mm1():
    sink = hs.Sink()
    server = hs.Server("srv", service_time=hs.ExponentialLatency(0.1), downstream=sink)
    source = hs.Source.poisson(rate=8, target=server)
    sim = hs.Simulation(end_time=Instant.from_seconds(20), sources=[source], entities=[server, sink])
    sim.sink = sink
    return sim


def test_parallel_runner_replicas_match_reference():
    gold = H.Golden("philox_64chains_replicas")
    results = hs.ParallelRunner(max_workers=8).run_replicas(_build_mm1, n_replicas=64, base_seed=1000)
    assert [r.name for r in results] == [f"replica_{i}" for i in range(64)]
    assert [r.summary.total_events_processed for r in results] == gold.meta["total_events"]
    assert [r.summary.duration_s for r in results] == gold.meta["duration_s"]


def test_parallel_simulation_without_links_equals_separate_runs():
    parts, solo = [], []
    for i in range(5):
        def build(i=i):
            sink = hs.Sink(f"k{i}")
            srv = hs.Server(f"s{i}", service_time=hs.ExponentialLatency(0.05 + 0.01 * i), downstream=sink)
            return hs.Source.poisson(rate=10, target=srv, name=f"src{i}"), srv, sink
        src, srv, sink = build()
        parts.append(hs.SimulationPartition(name=f"p{i}", entities=[srv, sink], sources=[src]))
    ps = hs.ParallelSimulation(parts, duration=10.0, seed=7).run()
    assert set(ps.partitions) == {f"p{i}" for i in range(5)}
    assert ps.total_events_processed == sum(s.total_events_processed for s in ps.partitions.values())
    assert ps.total_events_processed > 5 * 500
    # parallel/summary.py: one launch advanced all five, so each partition's wall time is that launch's
    assert set(ps.partition_wall_times) == set(ps.partitions) and len(set(ps.partition_wall_times.values())) == 1
    assert 0.0 < ps.speedup <= 5.0 and ps.parallelism_efficiency == ps.speedup / 5
    assert ps.total_windows == 0 and ps.coordination_efficiency == 1.0 and "Windows" not in str(ps)
    assert ps.entities["s3"].queue_stats.total_accepted == parts[3].entities[0].stats_accepted > 0
    assert ps.entities["k3"].events_handled == parts[3].entities[1].events_received > 0


# ---- networks of stations through the API (Server -> RandomRouter -> [Sink | NetworkLink -> next Server]) --------
def _build_ring(spec):
    """The reference-side construction of tests/golden/make_golden.py::run_ring_case, with this package's classes."""
    n = spec["n"]
    sinks = [hs.Sink(f"sink{i}") for i in range(n)]
    servers = [hs.Server(f"srv{i}", concurrency=spec.get("concurrency", 1),
                         service_time=hs.ExponentialLatency(spec["mean"]), queue_capacity=spec.get("queue_cap"))
               for i in range(n)]
    links, routers, sources = [], [], []
    for i in range(n):
        jk, jm = H.per_chain(spec.get("jitter_kind", "exp"), n)[i], H.per_chain(spec.get("jitter_mean"), n)[i]
        jit = None if (jk is None or jm is None) else hs.ExponentialLatency(jm) if jk == "exp" else hs.ConstantLatency(jm)
        loss = spec.get("loss", 0.0)
        if spec.get("name") == "ring_5_const_jitter":        # 0.5 ms + ConstantLatency(0.1 ms): the reference's datacenter preset itself
            links.append(hs.datacenter_network(f"link{i}"))
            links[-1].egress = servers[(i + 1) % n]
            assert (links[-1].latency.mean, links[-1].jitter.mean) == (spec["lat_min"], jm) and isinstance(links[-1].jitter, hs.ConstantLatency)
        else:
            links.append(hs.NetworkLink(f"link{i}", latency=hs.ConstantLatency(spec["lat_min"]), jitter=jit,
                                        bandwidth_bps=spec.get("bandwidth_bps"),
                                        packet_loss_rate=loss[i] if isinstance(loss, list) else loss,
                                        egress=servers[(i + 1) % n]))
        pat = (spec.get("rt_pattern") or ["sl"] * n)[i]
        routers.append(hs.RandomRouter(f"router{i}", targets=[sinks[i] if ch == "s" else links[i] for ch in pat]))
        servers[i].downstream = routers[i]
    # the Sources in `sources=[...]` order (several per Server: `more_sources`; slot = position among the Server's Sources)
    order, slot_plan = H.ring_source_plan(spec)
    by_slot = {i: [] for i in range(n)}
    for i in range(n):
        pr = (spec.get("profile") or [None] * n)[i]
        for slot, (kind, rate, is_first) in enumerate(slot_plan[i]):
            if is_first and pr is not None:
                profile = (hs.LinearRampProfile(duration_s=pr[1], start_rate=pr[2], end_rate=pr[3]) if pr[0] == "ramp" else
                           hs.SpikeProfile(baseline_rate=pr[1], spike_rate=pr[2], warmup_s=pr[3], spike_duration_s=pr[4]))
                by_slot[i].append(hs.Source.with_profile(profile, target=servers[i], poisson=True, name=f"src{i}"))
            else:
                make = hs.Source.poisson if kind == H.O.ARR_POISSON else hs.Source.constant
                by_slot[i].append(make(rate=rate, target=servers[i], name=f"src{i}_{slot}"))
    sources = [by_slot[i][slot] for i, slot in order]
    _build_ring.by_slot = by_slot
    return sources, servers, routers, links, sinks


def _check_ring_objects(gold, servers, routers, links, sinks):
    assert [s.stats_accepted for s in servers] == gold.accepted.tolist()
    assert [s.stats_dropped for s in servers] == gold.dropped.tolist()
    assert [s.stats.requests_completed for s in servers] == gold.completed.tolist()
    assert [s.stats.total_service_time for s in servers] == gold.total_service_s.tolist()
    assert [s.depth for s in servers] == gold.depth.tolist()
    assert [r.stats_routed for r in routers] == gold.routed.tolist()
    assert [l.packets_sent for l in links] == gold.packets_sent.tolist()
    if "packets_dropped" in gold.arrays:
        assert [l.packets_dropped for l in links] == gold.packets_dropped.tolist()
        assert [l.bytes_transmitted for l in links] == gold.bytes_transmitted.tolist()
        assert [l.link_stats.packets_dropped for l in links] == gold.packets_dropped.tolist()
    assert [k.events_received for k in sinks] == gold.received.tolist()
    lat = [x for k in sinks for x in k.latencies_s]
    assert lat == gold.sink_latency_s.tolist()


@pytest.mark.parametrize("name", ["ring_8_s42", "ring_5_const_link", "ring_6_c2_cap3", "ring_8_loss", "ring_5_loss_mixed",
                                  "ring_6_router_k", "ring_5_multi_source", "ring_4_multi_source_order", "ring_5_const_jitter",
                                  "ring_6_mixed_jitter"])
def test_ring_network_through_the_api_matches_reference_golden(name):
    gold = H.Golden(name)
    spec = gold.spec
    sources, servers, routers, links, sinks = _build_ring(spec)
    probes = [hs.Probe.on({"server": servers[i], "sink": sinks[i]}[H.PROBE_METRICS[m][0]], m, interval=iv)
              for i, prs in enumerate(H.ring_params(spec)["probe_list"]) for m, iv in prs]
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources,
                        entities=servers + routers + links + sinks, probes=[p for p, _ in probes], seed=spec["seed"])
    for i, t_s in spec.get("schedule") or []:
        sim.schedule(hs.Event(time=Instant.from_seconds(t_s), event_type="Request", target=servers[i]))
    summary = sim.run()
    if "generated_more" in gold.arrays:          # several Sources per Server: each Source's own generated_count
        for i, srcs in _build_ring.by_slot.items():
            assert [x.generated_count for x in srcs] == ([gold.generated[i]] + gold.generated_more[:, i].tolist())[:len(srcs)]
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    _check_ring_objects(gold, servers, routers, links, sinks)
    r0 = routers[0]
    assert sum(r0.target_counts.values()) in (r0.stats_routed, r0.stats_routed - 1)   # -1: the overshoot event


def test_linked_partitions_equal_the_single_heap_run():
    """ParallelSimulation with links (parallel/simulation.py:197-223): one shard per partition, exact result."""
    gold = H.Golden("ring_8_s42")
    spec = gold.spec
    sources, servers, routers, links, sinks = _build_ring(spec)
    src_of = {id(s._event_provider._target): s for s in sources}
    parts = []
    for k, idx in enumerate(([0, 1, 2], [3, 4], [5, 6, 7])):
        ents = [servers[i] for i in idx] + [routers[i] for i in idx] + [links[i] for i in idx] + [sinks[i] for i in idx]
        parts.append(hs.SimulationPartition(name=f"p{k}", entities=ents,
                                            sources=[src_of[id(servers[i])] for i in idx if id(servers[i]) in src_of]))
    plinks = [hs.PartitionLink("p0", "p1", min_latency=0.001), hs.PartitionLink("p1", "p2", min_latency=0.0005),
              hs.PartitionLink("p2", "p0", min_latency=0.001)]
    ps = hs.ParallelSimulation(parts, end_time=Instant.from_seconds(spec["end_s"]), links=plinks, seed=spec["seed"])
    summary = ps.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    # total_windows / window_size_s as the reference's coordinator counts them: W = the smallest PartitionLink.min_latency
    from happy_simulator_amd.parallel import reference_window_count
    assert summary.window_size_s == 0.0005 and summary.engine_exchanges > 0
    assert summary.total_windows == reference_window_count(Instant.Epoch, Instant.from_seconds(spec["end_s"]), 0.0005) > summary.engine_exchanges
    assert summary.total_cross_partition_events == sum(links[i]._entered for i in (2, 4, 7))
    assert set(summary.partitions) == {"p0", "p1", "p2"}
    assert sum(p.total_events_processed for p in summary.partitions.values()) == summary.total_events_processed
    _check_ring_objects(gold, servers, routers, links, sinks)
    # the rest of parallel/summary.py: timing split and merged entity summaries
    assert 0.0 < summary.barrier_overhead_seconds < summary.wall_clock_seconds
    assert summary.coordination_efficiency == 1.0 - summary.barrier_overhead_seconds / summary.wall_clock_seconds
    assert set(summary.partition_wall_times) == {"p0", "p1", "p2"} and summary.speedup > 0.0
    assert summary.parallelism_efficiency == summary.speedup / 3
    assert [summary.entities[f"srv{i}"].queue_stats.total_accepted for i in range(8)] == gold.accepted.tolist()
    assert [summary.entities[f"sink{i}"].events_handled for i in range(8)] == gold.received.tolist()
    assert set(summary.partitions["p1"].entities) == {"srv3", "srv4", "router3", "router4", "link3", "link4", "sink3", "sink4"}
    assert "Coordination efficiency" in str(summary)


def test_linked_partitions_validation():
    spec = dict(n=4, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01)
    sources, servers, routers, links, sinks = _build_ring(spec)
    mk = lambda idx, nm: hs.SimulationPartition(                                                   # noqa: E731
        name=nm, entities=[servers[i] for i in idx] + [routers[i] for i in idx] + [links[i] for i in idx] +
        [sinks[i] for i in idx], sources=[sources[i] for i in idx])
    a, b = mk([0, 1], "a"), mk([2, 3], "b")
    with pytest.raises(ValueError, match="no PartitionLink exists from 'b' to 'a'"):
        hs.ParallelSimulation([a, b], duration=1.0, links=[hs.PartitionLink("a", "b", min_latency=0.001)])
    with pytest.raises(ValueError, match="less than the PartitionLink min_latency"):
        hs.ParallelSimulation([a, b], duration=1.0, links=[hs.PartitionLink("a", "b", min_latency=0.001),
                                                            hs.PartitionLink("b", "a", min_latency=0.5)])
    with pytest.raises(ValueError, match="min_latency must be > 0"):
        hs.PartitionLink("a", "b", min_latency=0.0)
    with pytest.raises(ValueError, match="unknown dest partition"):
        hs.ParallelSimulation([a, b], duration=1.0, links=[hs.PartitionLink("a", "zz", min_latency=0.001)])
    with pytest.raises(hs.UnsupportedTopology, match="lookahead"):
        hs.Simulation(duration=1.0, sources=[], entities=[
            hs.Server("x", downstream=hs.NetworkLink("l", latency=hs.ExponentialLatency(0.01), egress=hs.Server("y")))
        ]).run()


@pytest.mark.parametrize("name", H.golden_names("lb"))
def test_load_balancer_topology_through_the_api_matches_reference_golden(name):
    """BASELINE configs[4] through the reference-shaped API: the wiring of examples/visual/chash_example.py:118-140
    (Sources with a client-id request factory -> LoadBalancer(ConsistentHash) -> Servers -> Sink)."""
    gold = H.Golden(name)
    spec = gold.spec
    p = H.lb_params(spec)
    S, B = p["S"], p["B"]
    sinks = [hs.Sink("sink")] if p["shared_sink"] else [hs.Sink(f"sink{j}") for j in range(B)]
    nodes = [hs.Server(f"srv{j}", concurrency=p["conc"][j], service_time=hs.ExponentialLatency(p["mean"][j]),
                       queue_capacity=None if p["qcap"][j] < 0 else p["qcap"][j],
                       downstream=sinks[0] if p["shared_sink"] else sinks[j]) for j in range(B)]
    strat = p["strategy"]       # ConsistentHash; RoundRobin = the reference's default (no strategy argument); Random
    lb = (hs.LoadBalancer("lb", backends=nodes, strategy=hs.ConsistentHash(virtual_nodes=p["vnodes"])) if strat == "chash" else
          hs.LoadBalancer("lb", backends=nodes) if strat == "round_robin" else hs.LoadBalancer("lb", backends=nodes, strategy=hs.Random()))
    def profile_of(pr):
        return (hs.LinearRampProfile(duration_s=pr[1], start_rate=pr[2], end_rate=pr[3]) if pr[0] == "ramp" else
                hs.SpikeProfile(baseline_rate=pr[1], spike_rate=pr[2], warmup_s=pr[3], spike_duration_s=pr[4]))

    profs = spec.get("profile") or [None] * S
    # (RoundRobin / Random need no client key: the plain request factory of Source.poisson(rate, target=lb) -- stop_after included)
    srcs = [(hs.Source.poisson(rate=p["rate"][i], name=f"src{i}", event_provider=ep) if profs[i] is None else
             hs.Source.with_profile(profile_of(profs[i]), poisson=True, name=f"src{i}", event_provider=ep))
            for i in range(S)
            for ep in [hs.ClientKeyEventProvider(lb, n_clients=p["n_clients"], stop_after=spec.get("stop_after_s")) if strat == "chash" else
                       hs.SimpleEventProvider(lb, "Request", hs.Source._resolve_stop_after(spec.get("stop_after_s")))]]
    probes = [hs.Probe.on({"server": nodes, "sink": sinks, "source": srcs}[who][i], metric, interval=iv)      # (lb_probes*.npz)
              for who, i, metric, iv in spec.get("probes") or []]
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=srcs, entities=[lb, *nodes, *sinks],
                        probes=[p for p, _ in probes], seed=spec["seed"])
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    for j, (_, data) in enumerate(probes):       # what the reference's probes appended to their Data containers
        a, b = gold.probe_off[j], gold.probe_off[j + 1]
        assert data.times() == [x / 1_000_000_000 for x in gold.probe_t_ns[a:b].tolist()]
        assert [int(v) for v in data.raw_values()] == gold.probe_v[a:b].tolist() and data.count() == b - a > 0
    assert [s.generated_count for s in srcs] == gold.generated.tolist()
    st = lb.stats
    assert [st.requests_received, st.requests_forwarded, st.requests_failed, st.no_backend_available] == gold.lb_stats[:4].tolist()
    assert [lb.get_backend_info(n).total_requests for n in nodes] == gold.backend_total_requests.tolist()
    if strat == "round_robin":
        assert lb.strategy._index == int(gold.rr_index[0]) == st.requests_forwarded
    assert [n.stats_accepted for n in nodes] == gold.accepted.tolist()
    assert [n.stats_dropped for n in nodes] == gold.dropped.tolist()
    assert [n.stats.requests_completed for n in nodes] == gold.completed.tolist()
    assert [n.stats.total_service_time for n in nodes] == gold.total_service_s.tolist()
    assert [n.depth for n in nodes] == gold.depth.tolist() and [n.active_requests for n in nodes] == gold.active.tolist()
    assert [k.events_received for k in sinks] == gold.received.tolist()
    assert sum((k.latencies_s for k in sinks), []) == gold.sink_latency_s.tolist()
    assert [t.nanoseconds for k in sinks for t in k.completion_times] == gold.sink_t_ns.tolist()
    es = summary.entities
    assert es["srv0"].queue_stats.total_accepted == gold.accepted[0]
    assert es[sinks[0].name].events_handled == sinks[0].events_received


def test_time_varying_profiles_through_the_api_match_reference_golden():
    """`Source.with_profile(LinearRampProfile | SpikeProfile, poisson=...)` (load/source.py:271-320): arrival times come
    from the reference's adaptive-Simpson + Brent inversion, restated on the device (csrc/hs_profile.hpp)."""
    gold = H.Golden("profile_mixed_4")
    spec = gold.spec
    p = H.spec_chain_params(spec)

    def profile(i):
        pr = p["profile"][i]
        if pr is None:
            return None
        return hs.LinearRampProfile(*pr[1:]) if pr[0] == "ramp" else hs.SpikeProfile(*pr[1:])

    chains = []
    for i in range(p["n"]):
        sink = hs.Sink(f"sink{i}")
        svc = hs.ExponentialLatency(p["mean"][i]) if spec["svc"][i] == "exp" else hs.ConstantLatency(p["mean"][i])
        srv = hs.Server(f"srv{i}", service_time=svc, downstream=sink)
        pr = profile(i)
        if pr is None:
            src = hs.Source.poisson(rate=p["rate"][i], target=srv, name=f"src{i}")
        else:
            src = hs.Source.with_profile(pr, target=srv, poisson=spec["arr"][i] == "poisson", name=f"src{i}")
        chains.append((src, srv, sink))
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=[c[0] for c in chains],
                        entities=[e for c in chains for e in c[1:]], seed=spec["seed"])
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    assert [c[0].generated_count for c in chains] == gold.generated.tolist()
    assert [c[1].stats.requests_completed for c in chains] == gold.completed.tolist()
    assert [c[1].stats.total_service_time for c in chains] == gold.total_service_s.tolist()
    for i, c in enumerate(chains):
        gt, glat = gold.sink_records(i)
        assert [t.nanoseconds for t in c[2].completion_times] == gt.tolist()
        assert c[2].latencies_s == glat.tolist()


@pytest.mark.parametrize("name", ["probe_depth_4chains", "probe_multi_4chains"])
def test_probes_through_the_api_match_reference_golden(name):
    """`Simulation(probes=[Probe.on(server, "depth", 0.5), ...])`: sample times and values equal what the live reference's
    probes appended to their Data containers (tests/golden/probe_*.npz).  probe_multi_4chains has up to four probes on one
    chain, two of them on the same interval (equal instants: the reference fires them in `probes=[...]` order) and one on
    the nanoseconds of a constant-rate Source."""
    gold = H.Golden(name)
    spec = gold.spec
    p = H.spec_chain_params(spec)
    chains, probes, datas = [], [], {}
    for i in range(p["n"]):
        sink = hs.Sink(f"sink{i}")
        srv = hs.Server(f"srv{i}", concurrency=p["conc"][i], service_time=hs.ExponentialLatency(p["mean"][i]),
                        queue_capacity=None if p["qcap"][i] < 0 else p["qcap"][i], downstream=sink)
        if p["arr"][i] == H.O.ARR_POISSON:
            src = hs.Source.poisson(rate=p["rate"][i], target=srv, name=f"src{i}")
        else:
            src = hs.Source.constant(rate=p["rate"][i], target=srv, name=f"src{i}")
        for j, (metric, interval) in enumerate(p["probe_list"][i]):
            target = {"server": srv, "sink": sink, "source": src}[H.PROBE_METRICS[metric][0]]
            pr, d = hs.Probe.on(target, metric, interval=interval)
            probes.append(pr)
            datas[(i, j)] = d
        chains.append((src, srv, sink))
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=[c[0] for c in chains],
                        entities=[e for c in chains for e in c[1:]], probes=probes, seed=spec["seed"])
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    for (i, j), d in datas.items():
        gt, gv = gold.probe_samples(i, j)
        assert d.raw_values() == gv.tolist()
        assert d.times() == (gt.astype(np.float64) / 1e9).tolist()        # Data stores time.to_seconds()
        assert d.count() == len(gt) and d.max() == max(gv.tolist())
    assert [c[2].events_received for c in chains] == gold.received.tolist()


@pytest.mark.parametrize("name", ["multi_source_4chains", "multi_source_order", "multi_source_replicas"])
def test_several_sources_per_server_through_the_api_match_reference_golden(name):
    """`Source.poisson(rate, target=server)` several times for one Server: every Source is an entity of its
    own with its own arrival stream; `sources=[...]` lists them in make_golden's order (multi_source_order: the further
    Sources first, which changes the pre-run sort indices; it also carries probes, schedule() calls and stop_after);
    multi_source_replicas runs one Simulation per chain through ParallelRunner (one prologue per lane)."""
    gold = H.Golden(name)
    spec = gold.spec
    p = H.spec_chain_params(spec)

    order, slot_plan = H.source_plan(spec, list(range(p["n"])))

    def build(i):
        sink = hs.Sink(f"sink{i}")
        lat = hs.ExponentialLatency(p["mean"][i]) if p["svc"][i] == H.O.LAT_EXP else hs.ConstantLatency(p["mean"][i])
        srv = hs.Server(f"srv{i}", concurrency=p["conc"][i], service_time=lat,
                        queue_capacity=None if p["qcap"][i] < 0 else p["qcap"][i], downstream=sink)
        stop = spec.get("stop_after_s")
        srcs = [(hs.Source.poisson if kind == H.O.ARR_POISSON else hs.Source.constant)(
                    rate=rate, target=srv, name=f"src{i}_{slot}", stop_after=stop)
                for slot, (kind, rate, _) in enumerate(slot_plan[i])]       # slot = position among the Server's Sources in `sources=`
        return srcs, srv, sink

    def check(i, srcs, srv, sink):
        assert srcs[0].generated_count == gold.generated[i]
        for j, x in enumerate(srcs[1:]):
            assert x.generated_count == gold.generated_more[j, i], (i, j)
        assert (srv.stats_accepted, srv.stats_dropped, srv.depth) == (gold.accepted[i], gold.dropped[i], gold.depth[i])
        assert srv._total_service_time == gold.total_service_s[i]
        gt, glat = gold.sink_records(i)
        assert [t.nanoseconds for t in sink.completion_times] == gt.tolist() and sink.latencies_s == glat.tolist()

    if spec["mode"] == "replicas":
        built = {}

        def make(i):
            def fn():
                srcs, srv, sink = built[i] = build(i)
                return hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=[srcs[sl] for c, sl in order if c == i],
                                     entities=[srv, sink])
            return fn
        res = hs.ParallelRunner().run_sweep([hs.RunConfig(name=f"r{i}", build_fn=make(i), seed=spec["seed"] + i)
                                             for i in range(p["n"])])
        assert [r.summary.total_events_processed for r in res] == gold.meta["total_events"]
        for i in range(p["n"]):
            check(i, *built[i])
        return
    chains = [build(i) for i in range(p["n"])]
    listed = [chains[c][0][sl] for c, sl in order]
    probes, datas = [], {}
    for i, prs in enumerate(p["probe_list"]):
        for j, (metric, interval) in enumerate(prs):
            pr, d = hs.Probe.on(chains[i][1], metric, interval=interval)
            probes.append(pr)
            datas[(i, j)] = d
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=listed,
                        entities=[e for c in chains for e in c[1:]], probes=probes, seed=spec["seed"])
    for c, t_s in spec.get("schedule") or []:
        sim.schedule(hs.Event(time=Instant.from_seconds(t_s), event_type="Request", target=chains[c][1]))
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    for i, c in enumerate(chains):
        check(i, *c)
    for (i, j), d in datas.items():
        gt, gv = gold.probe_samples(i, j)
        assert d.raw_values() == gv.tolist() and d.times() == (gt.astype(np.float64) / 1e9).tolist()


def test_schedule_through_the_api_matches_reference_golden():
    """Simulation.schedule(Event(...)) before run() (core/simulation.py:195-206): stations fed only by scheduled Requests,
    and scheduled Requests on top of Sources -- against what the live reference produced for the same calls."""
    for name in ("schedule_only", "schedule_with_sources"):
        gold = H.Golden(name)
        spec = gold.spec
        p = H.spec_chain_params(spec)
        n = p["n"]
        sinks = [hs.Sink(f"sink{i}") for i in range(n)]
        servers = [hs.Server(f"srv{i}", concurrency=p["conc"][i],
                             service_time=(hs.ExponentialLatency if H.per_chain(spec["svc"], n)[i] == "exp"
                                           else hs.ConstantLatency)(p["mean"][i]),
                             queue_capacity=None if p["qcap"][i] < 0 else p["qcap"][i], downstream=sinks[i])
                   for i in range(n)]
        sources = []
        for i in range(n):
            if p["no_source"][i]:
                continue
            make = hs.Source.poisson if H.per_chain(spec["arr"], n)[i] == "poisson" else hs.Source.constant
            sources.append(make(rate=p["rate"][i], target=servers[i], name=f"src{i}"))
        # entity order = the golden's construction order: server, sink per chain (stream base i <-> chain i)
        sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources,
                            entities=[e for pair in zip(servers, sinks) for e in pair], seed=spec["seed"])
        events = [hs.Event(time=Instant.from_seconds(t), event_type="Request", target=servers[c])
                  for c, t in spec["schedule"]]
        sim.schedule(events[:3])
        for ev in events[3:]:
            sim.schedule(ev)
        summary = sim.run()
        assert summary.total_events_processed == gold.meta["total_events"][0]
        assert summary.duration_s == gold.meta["duration_s"][0]
        assert summary.events_cancelled == 0
        assert [s.stats_accepted for s in servers] == gold.accepted.tolist()
        assert [s.stats_dropped for s in servers] == gold.dropped.tolist()
        assert [s.stats.requests_completed for s in servers] == gold.completed.tolist()
        assert [s.stats.total_service_time for s in servers] == gold.total_service_s.tolist()
        assert [k.events_received for k in sinks] == gold.received.tolist()
        assert [x for k in sinks for x in k.latencies_s] == gold.sink_latency_s.tolist()
        with pytest.raises(hs.UnsupportedTopology, match="after run"):
            sim.schedule(events[0])


def test_schedule_cancelled_events_and_refusals():
    """Event.cancel() before run(): the event is skipped when popped and counted (tests/test_event_cancellation.py:86-108
    pins 3 processed / 2 cancelled for the same pattern on a counting entity)."""
    sink = hs.Sink("sink")
    server = hs.Server("srv", service_time=hs.ConstantLatency(0.1), downstream=sink)
    sim = hs.Simulation(end_time=Instant.from_seconds(10.0), sources=[], entities=[server, sink])
    evs = [hs.Event(time=Instant.from_seconds(float(t)), event_type="Request", target=server) for t in range(1, 6)]
    sim.schedule(evs)
    evs[1].cancel()
    evs[3].cancel()
    summary = sim.run()
    assert summary.events_cancelled == 2
    assert sink.events_received == 3 and server.stats_accepted == 3
    assert [t.nanoseconds for t in sink.completion_times] == [1_100_000_000, 3_100_000_000, 5_100_000_000]
    # each request: Request@Server, Notify, Poll, Deliver, Request@worker, continuation, Request@Sink, completion Poll
    assert summary.total_events_processed == 3 * 8
    other = hs.Server("elsewhere", service_time=hs.ConstantLatency(0.1))
    sim2 = hs.Simulation(end_time=Instant.from_seconds(1.0), sources=[], entities=[server])
    sim2.schedule(hs.Event(time=Instant.from_seconds(0.5), event_type="Request", target=other))
    with pytest.raises(hs.UnsupportedTopology, match="only Requests for a Server of this Simulation"):
        sim2.run()
    with pytest.raises(ValueError, match="must have a 'target'"):
        hs.Event(time=Instant.from_seconds(0.5), event_type="Request")


def test_one_sink_behind_several_servers_matches_reference_golden():
    """`servers = [Server(..., downstream=sink) ...]`: the shared Sink's completion_times / latencies_s are in global
    processing order (components/common.py:36-44) -- here the device merge of the per-station logs."""
    gold = H.Golden("philox_shared_sink_6")
    spec = gold.spec
    p = H.spec_chain_params(spec)
    n = p["n"]
    sink = hs.Sink("sink")
    servers = [hs.Server(f"srv{i}", concurrency=p["conc"][i], service_time=hs.ExponentialLatency(p["mean"][i]),
                         downstream=sink) for i in range(n)]
    sources = [hs.Source.poisson(rate=p["rate"][i], target=servers[i], name=f"src{i}") for i in range(n)]
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=servers + [sink],
                        seed=spec["seed"])
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert sink.events_received == gold.received[0] == len(gold.sink_t_ns)
    assert [t.nanoseconds for t in sink.completion_times] == gold.sink_t_ns.tolist()
    assert sink.latencies_s == gold.sink_latency_s.tolist()
    assert [s.stats.requests_completed for s in servers] == gold.completed.tolist()
    st = sink.latency_stats()
    assert st["count"] == len(gold.sink_t_ns) and st["max"] == gold.sink_latency_s.max()
    assert summary.entities["sink"].events_handled == gold.received[0]


def test_network_larger_than_one_cooperative_launch_is_time_shared(monkeypatch):
    """More stations than one cooperative launch of the asynchronous engine holds (65 536 on an MI355X): Simulation.run()
    lets contiguous segments take turns on the device under the asynchronous-rounds protocol.  Forced here with a small
    capacity: every object of a 1 500-station ring ends up exactly as after the single-launch run."""
    spec = dict(n=1500, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, loss=0.1)

    def run(limit):
        if limit:
            monkeypatch.setattr(hs.Simulation, "_resident_stations", lambda self: limit)
        sources, servers, routers, links, sinks = _build_ring(spec)
        probes, datas = zip(*[hs.Probe.on(servers[i], "depth", interval=0.5) for i in (3, 700, 1499)])
        sim = hs.Simulation(end_time=Instant.from_seconds(6.0), sources=sources, entities=servers + routers + links + sinks,
                            probes=list(probes), seed=77)
        summary = sim.run()
        monkeypatch.undo()
        return (summary.total_events_processed, summary.duration_s, [(d.times(), d.raw_values()) for d in datas],
                [s.stats_accepted for s in servers], [s.stats.requests_completed for s in servers],
                [s.stats.total_service_time for s in servers], [r.stats_routed for r in routers],
                [l.packets_sent for l in links], [l.packets_dropped for l in links],
                [k.events_received for k in sinks], [x for k in sinks for x in k.latencies_s])

    whole = run(0)
    shared = run(400)                    # 4 segments of <= 400 stations
    assert whole == shared
    assert whole[0] > 100_000 and sum(whole[8]) > 0 and all(len(v[0]) >= 11 for v in whole[2])


def test_several_sources_per_server_on_the_time_shared_path(monkeypatch):
    """a network with several Sources per Server that exceeds one cooperative launch ran on the device and then died in
    write_back with KeyError 'generated_more' (ShardedNetwork.collect() did not gather the further Sources' tick counts).  A
    600-station ring with up to three further Sources per station, whole vs four time-shared segments: every object equal,
    the further Sources' generated_count included, the list order reversed (`extras_first`)."""
    n = 600
    more = [None] * n
    for i in range(0, n, 7):
        more[i] = [["constant", 4.0]] if i % 2 else [["poisson", 2.0], ["constant", 5.0], ["poisson", 4.0]]
    spec = dict(n=n, ext_rate=[3.0] * n, mean=0.1, lat_min=0.001, jitter_mean=0.01, more_sources=more, sources_order="extras_first")

    def run(limit):
        if limit:
            monkeypatch.setattr(hs.Simulation, "_resident_stations", lambda self: limit)
        sources, servers, routers, links, sinks = _build_ring(spec)
        sim = hs.Simulation(end_time=Instant.from_seconds(4.0), sources=sources, entities=servers + routers + links + sinks, seed=9)
        summary = sim.run()
        monkeypatch.undo()
        return (summary.total_events_processed, summary.duration_s, [src.generated_count for src in sources],
                [s.stats_accepted for s in servers], [s.stats.requests_completed for s in servers],
                [s.stats.total_service_time for s in servers], [r.stats_routed for r in routers],
                [l.packets_sent for l in links], [k.events_received for k in sinks], [x for k in sinks for x in k.latencies_s])

    whole = run(0)
    shared = run(160)
    assert whole == shared
    assert whole[0] > 50_000 and min(whole[2]) >= 1


@pytest.mark.parametrize("name", ["ring_6_probes", "ring_5_multi_probes", "ring_5_profiles", "ring_4_schedule"])
def test_probes_and_profiles_on_networked_stations_match_reference_golden(name):
    """Probe.on(server / sink, metric, interval), Source.with_profile(LinearRamp / Spike) and Simulation.schedule() on the
    stations of a ring (windowed network engine): every object as the live reference left it, probe samples value for
    value."""
    gold = H.Golden(name)
    spec = gold.spec
    sources, servers, routers, links, sinks = _build_ring(spec)
    probes, datas = [], {}
    for i, prs in enumerate(H.ring_params(spec)["probe_list"]):
        for j, (metric, interval) in enumerate(prs):         # several probes of a station: engine slots in this order
            who = H.PROBE_METRICS[metric][0]
            probe, data = hs.Probe.on({"server": servers[i], "sink": sinks[i]}[who], metric, interval=interval)
            probes.append(probe)
            datas[(i, j)] = data
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources,
                        entities=servers + routers + links + sinks, probes=probes, seed=spec["seed"])
    for i, t_s in spec.get("schedule") or []:
        sim.schedule(hs.Event(time=Instant.from_seconds(t_s), event_type="Request", target=servers[i]))
    summary = sim.run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    _check_ring_objects(gold, servers, routers, links, sinks)
    for (i, j), data in datas.items():
        gt, gv = gold.probe_samples(i, j)
        assert data.times() == [x / 1_000_000_000 for x in gt.tolist()]     # Instant.to_seconds()
        assert [int(v) for v in data.raw_values()] == gv.tolist()
        assert data.count() == len(gt) > 0


def test_ring_with_one_shared_sink_matches_oracle():
    """Every router of a ring forwards to the SAME Sink: its completion_times / latencies are the merge of the stations'
    logs in global processing order (device merge at write-back), as the oracle's single Sink node records them."""
    from oracle import hs_oracle as O

    n, end_s, seed = 7, 9.0, 91
    sink = hs.Sink("sink")
    servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.07)) for i in range(n)]
    links = [hs.NetworkLink(f"link{i}", latency=hs.ConstantLatency(0.0015), jitter=hs.ExponentialLatency(0.004),
                            egress=servers[(i + 1) % n]) for i in range(n)]
    routers = [hs.RandomRouter(f"router{i}", targets=[sink, links[i]]) for i in range(n)]
    for i in range(n):
        servers[i].downstream = routers[i]
    sources = [hs.Source.poisson(rate=3.0 + i, target=servers[i], name=f"src{i}") for i in range(n)]
    summary = hs.Simulation(end_time=Instant.from_seconds(end_s), sources=sources,
                            entities=servers + routers + links + [sink], seed=seed).run()
    g = O.Graph()
    src = [g.source(O.ARR_POISSON, 3.0 + i, stream_base=i) for i in range(n)]
    srv = [g.server(O.LAT_EXP, 0.07, stream_base=i) for i in range(n)]
    snk = g.sink()
    lnk = [g.link(0.0015, 0.004, stream_base=i) for i in range(n)]
    rtr = [g.router([snk, lnk[i]], stream_base=i) for i in range(n)]
    for i in range(n):
        g.target[src[i]] = srv[i]
        g.target[srv[i]] = rtr[i]
        g.target[lnk[i]] = srv[(i + 1) % n]
    r = O.run(g, H.ns_from_seconds(end_s), seed=seed)
    assert summary.total_events_processed == r.events_processed
    t, created = r.sinks[snk]
    assert sink.events_received == len(t) > 300
    assert [x.nanoseconds for x in sink.completion_times] == t.tolist()
    assert sink.latencies_s == ((t - created).astype(np.float64) / 1e9).tolist()


def test_profiles_and_schedule_on_a_large_network_through_the_api_match_oracle():
    """Source.with_profile and Simulation.schedule() on a 300-station ring (several wavefronts, two workgroups) through the
    API, every Server / router / link / Sink counter against the oracle.  (Round 1 refused this beyond 64 stations after a
    130-station ring with LinearRampProfile(3 s, 1 -> 9) on station 97 seemed to hang; the cause was that one arrival: the
    reference's own adaptive-Simpson inversion needs ~10^8 rate evaluations for it -- DESIGN.md section 1.2.)"""
    from oracle import hs_oracle as O

    n = 300
    prof = [None] * n
    prof[7], prof[97], prof[250], prof[131] = ["ramp", 3.0, 1.0, 9.0], ["ramp", 5.0, 3.0, 20.0], ["ramp", 4.0, 12.0, 2.0], \
        ["spike", 3.0, 30.0, 1.0, 1.5]
    spec = dict(name="ring_300_api", topology="ring", n=n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, profile=prof,
                schedule=[[200, 0.5], [200, 0.5], [64, 1.25], [299, 3.000000001]], end_s=4.0, seed=63)
    sources, servers, routers, links, sinks = _build_ring(spec)
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources,
                        entities=servers + routers + links + sinks, seed=spec["seed"])
    for i, t_s in spec["schedule"]:
        sim.schedule(hs.Event(time=Instant.from_seconds(t_s), event_type="Request", target=servers[i]))
    summary = sim.run()
    g, nodes = H.oracle_ring_graph(spec)
    p = H.ring_params(spec)
    r = O.run(g, p["end_ns"], seed=spec["seed"], schedule=[(nodes[c]["srv"], t) for c, t in p["schedule"]])
    assert summary.total_events_processed == r.events_processed > 30 * n
    srv, rtr, lnk, snk = ([nodes[i][k] for i in range(n)] for k in ("srv", "rtr", "lnk", "snk"))
    assert [s.stats_accepted for s in servers] == r.accepted[srv].tolist()
    assert [s.stats.requests_completed for s in servers] == r.completed[srv].tolist()
    assert [s.stats.total_service_time for s in servers] == r.total_service_s[srv].tolist()
    assert [s.depth for s in servers] == r.depth[srv].tolist()
    assert [x.stats_routed for x in routers] == r.routed[rtr].tolist()
    assert [x.packets_sent for x in links] == r.packets_sent[lnk].tolist()
    for i in range(n):
        t, created = r.sinks[snk[i]]
        assert [x.nanoseconds for x in sinks[i].completion_times] == t.tolist()


@pytest.mark.parametrize("k", [5, 7, 8, 10, 26, 62, 3, 17])
def test_tie_storms_through_the_api_with_sources_listed_in_another_order(k):
    """The reference numbers its pre-run events in the order `sources=[...]`, `probes=[...]` and the schedule() calls construct
    them; the API hands that order to the engine (hs_stations.source_order / probe_order / sched_rank) and the prologue
    (csrc/hs_exact.hpp) replays the two sort counters.  Tie storms (several Requests injected for one Server at the start
    instant, lock-step constant sources, probes on the same nanoseconds) with the Sources listed BACKWARDS, against the oracle
    built in that same construction order."""
    import random_specs as RS

    spec = RS.tie_spec(k)
    spec.pop("shared_sink", None)
    p = H.spec_chain_params(spec)
    n = p["n"]
    perm = list(range(n))[::-1]
    g, nodes = H.oracle_graph_for(spec, perm, perm)                  # sources, then probes, constructed in `perm` order
    r = O.run(g, p["end_ns"], seed=spec["seed"], schedule=[(nodes[c][1], t) for c, t in p["schedule"]])
    sinks = [hs.Sink(f"sink{i}") if p["downstream"] else None for i in range(n)]
    servers = [hs.Server(f"srv{i}", concurrency=p["conc"][i],
                         service_time=(hs.ExponentialLatency if H.per_chain(spec["svc"], n)[i] == "exp"
                                       else hs.ConstantLatency)(p["mean"][i]),
                         queue_capacity=None if p["qcap"][i] < 0 else p["qcap"][i], downstream=sinks[i]) for i in range(n)]
    stop = None if p["stop_ns"] < 0 else spec["stop_after_s"]
    srcs = {}
    for i in range(n):
        make = hs.Source.poisson if H.per_chain(spec["arr"], n)[i] == "poisson" else hs.Source.constant
        srcs[i] = make(rate=p["rate"][i], target=servers[i], name=f"src{i}", stop_after=stop)
    probes, datas = [], {}
    for c in perm:
        if p["probes"][c] is not None:
            metric, interval = p["probes"][c]
            target = {"server": servers[c], "sink": sinks[c], "source": srcs[c]}[H.PROBE_METRICS[metric][0]]
            pr, datas[c] = hs.Probe.on(target, metric, interval=interval)
            probes.append(pr)
    entities = [e for pair in zip(servers, sinks) for e in pair if e is not None]      # station i = chain i = stream base i
    sim = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=[srcs[c] for c in perm], entities=entities,
                        probes=probes, seed=spec["seed"])
    for c, t in spec.get("schedule") or []:
        sim.schedule(hs.Event(time=Instant.from_seconds(float(t)), event_type="Request", target=servers[c]))
    summary = sim.run()
    assert summary.total_events_processed == r.events_processed
    assert int(round(summary.duration_s * 1e9)) == r.final_time_ns
    srv = [nodes[c][1] for c in range(n)]
    assert [s.stats_accepted for s in servers] == r.accepted[srv].tolist()
    assert [s.stats_dropped for s in servers] == r.dropped[srv].tolist()
    assert [s.stats.requests_completed for s in servers] == r.completed[srv].tolist()
    assert [s.stats.total_service_time for s in servers] == r.total_service_s[srv].tolist()
    assert [srcs[c].generated_count for c in range(n)] == [int(r.generated[nodes[c][0]]) for c in range(n)]
    for c in range(n):
        if sinks[c] is not None:
            assert [t.nanoseconds for t in sinks[c].completion_times] == r.sinks[nodes[c][2]][0].tolist()
        if c in datas:
            t, v = r.sinks[g.probe_nodes[c]]
            assert datas[c].raw_values() == v.tolist()


def test_batched_replicas_carry_profiles_probes_and_scheduled_requests():
    """ParallelRunner batches its replicas into one launch; every optional per-station field must survive the batching:
    a replica with a LinearRampProfile source, a Probe and schedule()d Requests equals the same Simulation run on its own
    (round-1 advisor finding: `_concat` dropped them -- the ramp ran at its peak rate, probes came back empty)."""
    datas = {}

    def build(k=None):
        sink = hs.Sink("sink")
        srv = hs.Server("srv", service_time=hs.ExponentialLatency(0.05), queue_capacity=6, downstream=sink)
        src = hs.Source.with_profile(hs.LinearRampProfile(duration_s=4.0, start_rate=4.0, end_rate=20.0), target=srv,
                                     poisson=True, name="src")
        pr, d = hs.Probe.on(srv, "depth", interval=0.25)
        sim = hs.Simulation(end_time=Instant.from_seconds(6.0), sources=[src], entities=[srv, sink], probes=[pr])
        evs = [hs.Event(time=Instant.from_seconds(t), event_type="Request", target=srv) for t in (0.0, 0.0, 1.5, 3.0)]
        sim.schedule(evs)
        evs[2].cancel()
        sim._parts = (src, srv, sink, d)
        return sim

    sims = []

    def build_and_keep():
        sims.append(build())
        return sims[-1]

    res = hs.ParallelRunner().run_sweep([hs.RunConfig(name=f"r{i}", build_fn=build_and_keep, seed=900 + i) for i in range(5)])
    for i, (r, s) in enumerate(zip(res, sims)):
        alone = build()
        alone._seed = 900 + i
        want = alone.run()
        assert r.summary.total_events_processed == want.total_events_processed > 300
        assert r.summary.duration_s == want.duration_s and r.summary.events_cancelled == want.events_cancelled == 1
        for a, b in zip(s._parts[:3], alone._parts[:3]):
            assert type(a) is type(b)
        assert s._parts[0].generated_count == alone._parts[0].generated_count
        assert s._parts[1].stats_accepted == alone._parts[1].stats_accepted
        assert s._parts[1].stats_dropped == alone._parts[1].stats_dropped
        assert s._parts[2].latencies_s == alone._parts[2].latencies_s
        assert s._parts[3].raw_values() == alone._parts[3].raw_values() and s._parts[3].count() == 24
    assert len({r.summary.total_events_processed for r in res}) > 1          # different seeds, different runs


def test_linked_partitions_with_one_sink_fed_from_every_partition():
    """A collector behind Servers of SEVERAL partitions (advisor finding: each shard's write-back used to overwrite it, only
    the last shard's records survived): the linked run merges all shards' records in completion order and equals the
    single-heap Simulation of the same objects."""
    def build():
        n = 6
        sink = hs.Sink("sink")
        servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.06)) for i in range(n)]
        links = [hs.NetworkLink(f"link{i}", latency=hs.ConstantLatency(0.002), jitter=hs.ExponentialLatency(0.005),
                                egress=servers[(i + 1) % n]) for i in range(n)]
        routers = [hs.RandomRouter(f"router{i}", targets=[sink, links[i]]) for i in range(n)]
        for i in range(n):
            servers[i].downstream = routers[i]
        sources = [hs.Source.poisson(rate=4.0 + i, target=servers[i], name=f"src{i}") for i in range(n)]
        return sink, servers, links, routers, sources

    sink1, servers, links, routers, sources = build()
    want = hs.Simulation(end_time=Instant.from_seconds(8.0), sources=sources, entities=servers + routers + links + [sink1],
                         seed=77).run()
    sink, servers, links, routers, sources = build()
    parts = []
    for k, idx in enumerate(([0, 1], [2, 3, 4], [5])):
        parts.append(hs.SimulationPartition(name=f"p{k}", sources=[sources[i] for i in idx],
                                            entities=[servers[i] for i in idx] + [routers[i] for i in idx] + [links[i] for i in idx]))
    plinks = [hs.PartitionLink("p0", "p1", min_latency=0.002), hs.PartitionLink("p1", "p2", min_latency=0.002),
              hs.PartitionLink("p2", "p0", min_latency=0.002)]
    got = hs.ParallelSimulation(parts, end_time=Instant.from_seconds(8.0), links=plinks, seed=77).run()
    assert got.total_events_processed == want.total_events_processed and got.duration_s == want.duration_s
    assert sink.events_received == sink1.events_received > 200
    assert [t.nanoseconds for t in sink.completion_times] == [t.nanoseconds for t in sink1.completion_times]
    assert sink.latencies_s == sink1.latencies_s


def test_auto_terminate_of_a_schedule_driven_simulation():
    """`Simulation(end_time=None)` = Instant.Infinity = auto-termination (core/simulation.py:311-322, core/event_heap.py:102-104):
    driven by schedule()d Requests only, the run ends when the heap is empty -- the last completion -- and processes nothing
    beyond it; with a Source it would never return (refused, as before)."""
    sinks = [hs.Sink(f"sink{i}") for i in range(3)]
    servers = [hs.Server(f"srv{i}", concurrency=c, service_time=hs.ExponentialLatency(m), queue_capacity=q, downstream=sinks[i])
               for i, (c, m, q) in enumerate([(1, 0.2, None), (2, 0.5, 3), (1, 0.05, 0)])]
    sim = hs.Simulation(sources=[], entities=[e for pair in zip(servers, sinks) for e in pair], seed=5)
    rng = np.random.default_rng(3)
    calls = [(int(rng.integers(0, 3)), float(np.round(rng.uniform(0.0, 4.0), 2))) for _ in range(60)] + [(0, 0.0), (0, 0.0), (1, 0.0)]
    for c, t in calls:
        sim.schedule(hs.Event(time=Instant.from_seconds(t), event_type="Request", target=servers[c]))
    summary = sim.run()
    g = O.Graph()
    nodes = []
    for i, (c, m, q) in enumerate([(1, 0.2, -1), (2, 0.5, 3), (1, 0.05, 0)]):
        sv = g.server(O.LAT_EXP, m, concurrency=c, queue_cap=q, stream_base=i)
        sk = g.sink()
        g.target[sv] = sk
        nodes.append((sv, sk))
    r = O.run(g, 1 << 61, seed=5, schedule=[(nodes[c][0], H.ns_from_seconds(t)) for c, t in calls])
    assert summary.total_events_processed == r.events_processed > 300
    assert int(round(summary.duration_s * 1e9)) == r.final_time_ns
    assert [s.stats_accepted for s in servers] == [int(r.accepted[n[0]]) for n in nodes]
    assert [s.stats_dropped for s in servers] == [int(r.dropped[n[0]]) for n in nodes]
    assert [s.stats.requests_completed for s in servers] == [int(r.completed[n[0]]) for n in nodes]
    for i, n in enumerate(nodes):
        assert [t.nanoseconds for t in sinks[i].completion_times] == r.sinks[n[1]][0].tolist()
    assert all(s.depth == 0 and s.active_requests == 0 for s in servers)              # everything drained
    src = hs.Source.poisson(rate=1, target=servers[0])
    with pytest.raises(hs.UnsupportedTopology, match="never terminates"):
        hs.Simulation(sources=[src], entities=servers).run()


def test_sink_latency_stats_on_the_device_equal_the_reference_formula():
    """Sink.latency_stats() (components/common.py:59-76): above 4 096 records the sort, the left-to-right sum of the sorted
    values and the interpolated percentiles run on the device (hs_sink_latency_stats) -- bit-identical to the list formula."""
    from happy_simulator_amd.entities import _percentile_sorted

    sink = hs.Sink("sink")
    servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sink) for i in range(40)]
    sources = [hs.Source.poisson(rate=8, target=s, name=f"src{i}") for i, s in enumerate(servers)]
    hs.Simulation(duration=30, sources=sources, entities=servers + [sink], seed=12).run()
    assert sink.events_received > 8000
    got = sink.latency_stats()
    lat = sorted(sink.latencies_s)
    want = {"count": len(lat), "avg": sum(lat) / len(lat), "min": lat[0], "max": lat[-1],
            "p50": _percentile_sorted(lat, 0.50), "p99": _percentile_sorted(lat, 0.99)}
    assert got == want


def test_probe_on_a_sink_shared_by_several_servers_matches_the_oracle():
    """`Probe.on(sink, "events_received")` where `sink` sits behind several Servers (VERDICT r2 Missing 5): the probe ticks on the
    first of those stations, its samples are the shared Sink's merged record count before each tick."""
    spec = dict(name="shared_sink_probe", n_chains=5, arr="poisson", rate=[8.0, 5.0, 12.0, 3.0, 9.0], svc="exp",
                mean=[0.1, 0.05, 0.07, 0.2, 0.1], concurrency=[1, 2, 1, 1, 3], queue_cap=None, stop_after_s=None, downstream=True,
                shared_sink=True, probes=[["events_received", 0.13], None, None, None, None], end_s=9.0, rng="philox", seed=77,
                mode="single", trace=False)
    (chain_ids, nodes, r), = H.run_oracle_for_spec(spec)
    p = H.spec_chain_params(spec)
    sink = hs.Sink("sink")
    servers = [hs.Server(f"srv{i}", concurrency=p["conc"][i], service_time=hs.ExponentialLatency(p["mean"][i]), downstream=sink)
               for i in range(5)]
    sources = [hs.Source.poisson(rate=p["rate"][i], target=servers[i], name=f"src{i}") for i in range(5)]
    probe, data = hs.Probe.on(sink, "events_received", interval=0.13)
    summary = hs.Simulation(end_time=Instant.from_seconds(spec["end_s"]), sources=sources, entities=servers + [sink], probes=[probe],
                            seed=spec["seed"]).run()
    assert summary.total_events_processed == r.events_processed
    t, v = r.sinks[r.probe_nodes[0]]
    assert len(t) > 60
    np.testing.assert_array_equal(np.asarray(data._t_ns), t)
    np.testing.assert_array_equal(np.asarray(data._v), v)
    assert sink.events_received == len(r.sinks[nodes[0][2]][0])


def test_probes_on_plain_chains_take_the_object_free_path_and_equal_the_general_lowering(monkeypatch):
    """>= 64 plain Source -> Server -> Sink chains with Probes: the probes go into the engine's arrays without a Station object
    per chain (lowering.plain_probe_arrays) and their samples stay on the device until a Data is first read.  Same configuration
    through the general lowering (plain detection switched off): every sample, statistic and Sink record equal."""
    import happy_simulator_amd.simulation as SIM

    def build():
        sinks = [hs.Sink(f"k{i}") for i in range(96)]
        servers = [hs.Server(f"s{i}", concurrency=1 + (i % 3 == 0), service_time=hs.ExponentialLatency(0.05 + 0.01 * (i % 5)), downstream=sinks[i])
                   for i in range(96)]
        sources = [hs.Source.poisson(rate=6 + i % 7, target=servers[i], name=f"src{i}") for i in range(96)]
        probes, data = [], []
        for i in range(96):
            tgt, metric = [(servers[i], "depth"), (servers[i], "utilization"), (sinks[i], "events_received"), (sources[i], "generated_count"),
                           (servers[i], "stats_accepted"), (servers[i], "requests_completed")][i % 6]
            p, d = hs.Probe.on(tgt, metric, interval=[0.1, 0.25, 0.5, 1.0][i % 4])
            probes.append(p); data.append(d)
            if i % 8 == 0:                                   # a second and third probe on the same Server
                ps, ds = hs.Probe.on_many(servers[i], ["active_requests", "stats_dropped"], interval=0.2)
                probes += ps; data += [ds["active_requests"], ds["stats_dropped"]]
        probes = probes[::-1]; data = data[::-1]            # `probes=[...]` in another order than the stations
        sim = hs.Simulation(duration=4.0, sources=sources, entities=servers + sinks, probes=probes)
        return sim, servers, sinks, sources, data

    monkeypatch.setattr(SIM, "LAZY_PROBES_MIN", 0)           # (the lazy read-back is for runs with hundreds of Probes: force it here)
    sim, servers, sinks, sources, data = build()
    summary = sim.run()
    assert sim.lowered().plain is not None and sim.lowered()._stations is None           # no Station objects were built
    assert all(d._lazy is not None for d in data)                                        # nothing downloaded yet
    fast = ([d.values for d in data], [(s.stats_accepted, s.stats.requests_completed) for s in servers],
            [list(k.completion_times) for k in sinks], [s.generated_count for s in sources], summary.total_events_processed)
    monkeypatch.setattr(SIM, "plain_chains", lambda *a: None)
    sim2, servers2, sinks2, sources2, data2 = build()
    summary2 = sim2.run()
    assert sim2.lowered()._stations is not None                  # Station objects, attach_probes, write_back_probes
    general = ([d.values for d in data2], [(s.stats_accepted, s.stats.requests_completed) for s in servers2],
               [list(k.completion_times) for k in sinks2], [s.generated_count for s in sources2], summary2.total_events_processed)
    assert fast == general
    assert sum(len(v) for v in fast[0]) > 1000


def test_capacity_probes_are_the_references_functions_of_active_requests():
    """`available_capacity`, the callable `has_capacity` and `utilization` are sampled as `active_requests` on the engine and
    mapped by the Data container with the reference's expressions (server.py:153-173,191-200; the live relation is pinned by
    tests/test_oracle_live_reference.py::test_live_reference_capacity_probes_are_functions_of_active_requests)."""
    sinks = [hs.Sink(f"k{i}") for i in range(12)]
    servers = [hs.Server(f"s{i}", concurrency=1 + i % 3, service_time=hs.ExponentialLatency(0.12), downstream=sinks[i]) for i in range(12)]
    sources = [hs.Source.poisson(rate=9 + i, target=servers[i], name=f"src{i}") for i in range(12)]
    probes, data = [], []
    for sv in servers:
        ps, ds = hs.Probe.on_many(sv, ["active_requests", "available_capacity", "has_capacity", "utilization"], interval=0.05)
        probes += ps; data.append(ds)
    hs.Simulation(duration=5.0, sources=sources, entities=servers + sinks, probes=probes).run()
    seen_full = 0
    for sv, ds in zip(servers, data):
        c = sv.concurrency
        act = ds["active_requests"].raw_values()
        assert len(act) == 100 and ds["available_capacity"].times() == ds["active_requests"].times()
        assert ds["available_capacity"].raw_values() == [c - a for a in act]
        assert ds["has_capacity"].raw_values() == [a < c for a in act]
        assert ds["utilization"].raw_values() == [a / c for a in act]
        seen_full += sum(a == c for a in act)
    assert seen_full > 50


# ---- X2 pinned to the reference's OWN ParallelSimulation(...).run() (fixtures: tests/golden/make_golden.py PARALLEL_CASES) --------------
def test_parallel_simulation_equals_the_reference_known_answer_tests():
    """The configurations of the reference's own tests (tests/integration/test_parallel_simulation.py:75-109,239-289) -- constant
    Sources feeding Counters, 1-3 partitions, with and without `links=[]` -- against what the live reference's ParallelSimulation
    computed for them: every Counter total (100 / 100, 100, 50), per-partition and total event counts, durations, no windows."""
    gold = H.Golden("parallel_ref_counters")
    for sub, want in zip(gold.spec["counters"], gold.meta["counters"]):
        counters = [hs.Counter(f"counter{i}") for i in range(len(sub["rates"]))]
        srcs = [hs.Source.constant(rate=r, target=c, event_type="Ping") for r, c in zip(sub["rates"], counters)]
        parts = [hs.SimulationPartition(name=f"P{i}", entities=[c], sources=[s_]) for i, (c, s_) in enumerate(zip(counters, srcs))]
        summ = hs.ParallelSimulation(parts, duration=sub["duration"], **(dict(links=[]) if sub.get("empty_links") else {})).run()
        assert [c.total for c in counters] == want["totals"] == want["sequential_totals"]
        assert [s_.generated_count for s_ in srcs] == want["generated"]
        assert summ.total_events_processed == want["total_events"] and summ.duration_s == want["duration_s"]
        assert [summ.partitions[f"P{i}"].total_events_processed for i in range(len(parts))] == want["partition_events"]
        assert [summ.partitions[f"P{i}"].duration_s for i in range(len(parts))] == want["partition_duration_s"]
        assert summ.total_windows == want["total_windows"] == 0 and summ.total_cross_partition_events == 0
        assert summ.events_per_second == want["events_per_second"]
        assert {k: v.events_handled for k, v in summ.entities.items()} == want["entity_events_handled"]


def test_parallel_simulation_without_links_equals_the_reference_fixture():
    """Six Philox-plugged M/M/c partitions through the reference's ParallelSimulation (one Simulation per partition on its thread
    pool, model-wide entity numbering, one seed) == hs.ParallelSimulation: totals per partition, every statistic and Sink record."""
    gold = H.Golden("parallel_philox_independent_6")
    spec, p = gold.spec, H.spec_chain_params(gold.spec)
    parts, servers, sinks, sources = [], [], [], []
    for i in range(spec["n_chains"]):
        sink = hs.Sink(f"sink{i}")
        svc = hs.ExponentialLatency(p["mean"][i]) if spec["svc"][i] == "exp" else hs.ConstantLatency(p["mean"][i])
        srv = hs.Server(f"srv{i}", concurrency=p["conc"][i], service_time=svc,
                        queue_capacity=None if p["qcap"][i] < 0 else p["qcap"][i], downstream=sink)
        factory = hs.Source.poisson if spec["arr"][i] == "poisson" else hs.Source.constant
        src = factory(rate=spec["rate"][i], target=srv, name=f"src{i}")
        parts.append(hs.SimulationPartition(name=f"P{i}", entities=[srv, sink], sources=[src]))
        servers.append(srv); sinks.append(sink); sources.append(src)
    summ = hs.ParallelSimulation(parts, end_time=Instant.from_seconds(spec["end_s"]), seed=spec["seed"]).run()
    par = gold.meta["parallel"]
    assert [summ.partitions[f"P{i}"].total_events_processed for i in range(len(parts))] == gold.meta["total_events"]
    assert [summ.partitions[f"P{i}"].duration_s for i in range(len(parts))] == gold.meta["duration_s"]
    assert summ.total_events_processed == par["total_events"] and summ.duration_s == par["duration_s"]
    assert summ.events_per_second == par["events_per_second"] and summ.total_windows == 0 == summ.total_cross_partition_events
    assert {k: v.events_handled for k, v in summ.entities.items()} == par["entity_events_handled"]
    assert [s_.generated_count for s_ in sources] == gold.generated.tolist()
    for key, attr in (("accepted", "stats_accepted"), ("dropped", "stats_dropped"), ("depth", "depth"), ("active", "active_requests")):
        assert [getattr(sv, attr) for sv in servers] == gold.arrays[key].tolist(), key
    assert [sv.stats.requests_completed for sv in servers] == gold.completed.tolist()
    assert [sv.stats.total_service_time for sv in servers] == gold.total_service_s.tolist()
    for i, sk in enumerate(sinks):
        t, lat = gold.sink_records(i)
        np.testing.assert_array_equal(sk.completion_ns, t)
        np.testing.assert_array_equal(sk.latencies_array, lat)


def _build_pipeline(spec):
    """make_golden._pipeline(spec, "network") with this package's classes: lane j flows stage 0 -> NetworkLink -> stage 1 -> ... ->
    Sink_j; one partition per stage."""
    lanes, stages = spec["lanes"], spec["stages"]
    servers = []
    for k, stg in enumerate(stages):
        mean = H.per_chain(stg["mean"], lanes)
        servers.append([hs.Server(f"srv{k}_{j}", concurrency=stg.get("concurrency", 1), queue_capacity=stg.get("queue_cap"),
                                  service_time=(hs.ExponentialLatency(mean[j]) if stg["svc"] == "exp" else hs.ConstantLatency(mean[j])))
                        for j in range(lanes)])
    sinks = [hs.Sink(f"sink{j}") for j in range(lanes)]
    hops = []
    for j in range(lanes):
        servers[-1][j].downstream = sinks[j]
    for k in range(len(stages) - 1):
        row = []
        for j in range(lanes):
            jit = hs.ExponentialLatency(spec["hop_jitter"]) if spec.get("hop_jitter") else None
            row.append(hs.NetworkLink(f"hop{k}_{j}", latency=hs.ConstantLatency(spec["hop_latency"]), jitter=jit, egress=servers[k + 1][j]))
            servers[k][j].downstream = row[-1]
        hops.append(row)
    rate = H.per_chain(spec["rate"], lanes)
    sources = [hs.Source.poisson(rate=rate[j], target=servers[0][j], name=f"src{j}") for j in range(lanes)]
    parts = [hs.SimulationPartition(name=f"P{k}", entities=row + (hops[k] if k < len(hops) else []) + (sinks if k == len(stages) - 1 else []),
                                    sources=sources if k == 0 else []) for k, row in enumerate(servers)]
    ploss = spec.get("packet_loss") or [0.0] * (len(stages) - 1)
    links = [hs.PartitionLink(f"P{k}", f"P{k + 1}", min_latency=spec["hop_latency"], packet_loss=ploss[k]) for k in range(len(stages) - 1)]
    return parts, links, sources, servers, hops, sinks


@pytest.mark.parametrize("name", ["parallel_linked_pipeline", "parallel_linked_three_stages", "parallel_linked_hazard"])
def test_linked_partitions_against_the_reference_parallel_simulation(name):
    """hs.ParallelSimulation(partitions, links=[PartitionLink ...]) -- one shard of the network engine per partition -- against the
    three runs of the LIVE reference the fixture holds (tests/golden/make_golden.py run_parallel_linked_case):

    * `seqnet_*`: the same library topology (NetworkLinks) in ONE reference Simulation -- equal to the bit: totals, final time,
      every statistic and Sink record.  This is the semantics the engine follows.
    * `win_*`: the reference's own windowed ParallelSimulation (whose hops must be future-delivering entities: its coordinator
      refuses NetworkLinks, tests/test_oracle_live_reference.py).  Where that run is self-consistent (no time-travel drops) every
      Sink record and completion count agrees, `total_windows` / `window_size_s` / `total_cross_partition_events` are the
      reference's numbers, and the event totals differ only by what is written below.
    * `parallel_linked_hazard` -- failing by design: the reference's windowed run drops more than half of the downstream
      partition's traffic as time travel (one-event overshoot per window); the engine equals the reference's
      SEQUENTIAL run instead."""
    gold = H.Golden(name)
    spec = gold.spec
    parts, links, sources, servers, hops, sinks = _build_pipeline(spec)
    summ = hs.ParallelSimulation(parts, end_time=Instant.from_seconds(spec["end_s"]), links=links, seed=spec["seed"]).run()
    sn, win = gold.meta["seq_network"], gold.meta["windowed"]
    flat = [sv for row in servers for sv in row]
    # == the reference's sequential run of the same library entities
    assert summ.total_events_processed == sn["total_events"] and summ.duration_s == sn["duration_s"]
    assert [s_.generated_count for s_ in sources] == gold.seqnet_generated.tolist()
    for key, attr in (("accepted", "stats_accepted"), ("dropped", "stats_dropped"), ("depth", "depth"), ("active", "active_requests")):
        assert [getattr(sv, attr) for sv in flat] == gold.arrays["seqnet_" + key].tolist(), key
    assert [sv.stats.requests_completed for sv in flat] == gold.seqnet_completed.tolist()
    assert [sv.stats.total_service_time for sv in flat] == gold.seqnet_total_service_s.tolist()
    assert [h.packets_sent for row in hops for h in row] == sn["packets_sent"]
    np.testing.assert_array_equal(np.concatenate([k.completion_ns for k in sinks]), gold.seqnet_sink_t_ns)
    np.testing.assert_array_equal(np.concatenate([k.latencies_array for k in sinks]), gold.seqnet_sink_latency_s)
    # the reference's windowed bookkeeping: W = min(PartitionLink.min_latency), its binary64 window count
    assert summ.window_size_s == win["window_size_s"] and summ.total_windows == win["total_windows"]
    assert set(summ.partitions) == set(win["partition_events"])
    if name == "parallel_linked_hazard":
        assert not gold.meta["windowed_equals_sequential"] and sum(win["time_travel_drops"].values()) > 100
        assert sum(k.events_received for k in sinks) == int(gold.seqfut_received.sum()) > 1.6 * int(gold.win_received.sum())
        return
    # ... and where the windowed run loses nothing, it saw what the engine saw
    np.testing.assert_array_equal(np.concatenate([k.completion_ns for k in sinks]), gold.win_sink_t_ns)
    assert [sv.stats.requests_completed for sv in flat] == gold.win_completed.tolist()
    # a cross-partition event = a request that entered a hop; the windowed run counts the ones its partitions SENT, incl. those of
    # its per-partition events beyond end_time (at most one per upstream partition)
    assert 0 <= win["total_cross_partition_events"] - summ.total_cross_partition_events <= len(spec["stages"]) - 1
    # event totals: a NetworkLink is two events per hop where the future-delivering entity is one (its continuation = packets_sent),
    # and every PARTITION of the windowed run processes its own one event beyond end_time, the engine (one heap) a single one
    diff = summ.total_events_processed - sum(sn["packets_sent"]) - win["total_events"]
    assert -len(spec["stages"]) <= diff <= 1


@pytest.mark.parametrize("name", ["parallel_linked_loss", "parallel_linked_loss_three"])
def test_partition_links_that_lose_packets_against_the_reference_parallel_simulation(name):
    """VERDICT r5 missing 1 / next 9: `PartitionLink(packet_loss=p)`.  The reference's coordinator drops a cross-partition event at the
    exchange when `self._rng.random() < link.packet_loss`, one `random.Random(seed)` for the run (parallel/coordinator.py:68,203-205).
    hs.ParallelSimulation replays that generator on the host over the run's cross-partition sends in the sending partition's
    processing order and hands the decisions to the engine as one bit per packet (happy_simulator_amd/parallel.py
    `_replay_partition_losses`).  Against the LIVE windowed run of the fixture (the coordinator's seed = the run's seed): every Sink
    record, every completion count and service-time sum; the hops' intake up to each partition's own event beyond end_time."""
    gold = H.Golden(name)
    spec = gold.spec
    assert spec["coord_seed"] == spec["seed"]
    parts, links, sources, servers, hops, sinks = _build_pipeline(spec)
    summ = hs.ParallelSimulation(parts, end_time=Instant.from_seconds(spec["end_s"]), links=links, seed=spec["seed"]).run()
    win = gold.meta["windowed"]
    assert sum(win["time_travel_drops"].values()) == 0
    flat = [sv for row in servers for sv in row]
    np.testing.assert_array_equal(np.concatenate([k.completion_ns for k in sinks]), gold.win_sink_t_ns)
    np.testing.assert_array_equal(np.concatenate([k.latencies_array for k in sinks]), gold.win_sink_latency_s)
    assert [k.events_received for k in sinks] == gold.win_received.tolist()
    assert [sv.stats.requests_completed for sv in flat] == gold.win_completed.tolist()
    assert [sv.stats.total_service_time for sv in flat] == gold.win_total_service_s.tolist()
    d_acc = gold.win_accepted - np.array([sv.stats_accepted for sv in flat])
    assert (d_acc >= 0).all() and d_acc.sum() <= len(spec["stages"])           # (every PARTITION's own event beyond end_time)
    # what the lossy PartitionLink took: close to p of what its hops took in; NetworkLink.packets_dropped is not the hop's doing
    for k, p in enumerate(spec["packet_loss"]):
        entered = sum(h._entered for h in hops[k])
        through = sum(sv.stats_accepted + sv.stats_dropped for sv in servers[k + 1])
        assert all(h.packets_dropped == 0 for h in hops[k])
        assert (entered - through <= spec["lanes"]) if p == 0 else abs((entered - through) / entered - p) < 0.08
    assert summ.window_size_s == win["window_size_s"] and summ.total_windows == win["total_windows"]
    assert 0 <= win["total_cross_partition_events"] - summ.total_cross_partition_events <= len(spec["stages"]) - 1 + spec["lanes"]


def test_partition_link_loss_out_of_two_partitions_is_refused_by_name():
    """Two lossy PartitionLinks out of different partitions: the reference's single loss stream interleaves by its windows'
    event-by-event overshoot (measured on the live class: the plain send-time order does NOT reproduce it) -- refused, not guessed.
    `PartitionLink(latency=...)` stays refused (the reference's coordinator calls `.sample()`, which its distributions lack)."""
    spec = dict(lanes=2, rate=[5.0, 6.0], seed=3, end_s=2.0, hop_latency=0.02, packet_loss=[0.2, 0.3],
                stages=[dict(svc="exp", mean=0.05), dict(svc="const", mean=0.0), dict(svc="const", mean=0.0)])
    parts, links, *_ = _build_pipeline(spec)
    with pytest.raises(hs.UnsupportedTopology, match="several partitions"):
        hs.ParallelSimulation(parts, duration=2.0, links=links)
    parts, links, *_ = _build_pipeline(dict(spec, packet_loss=None))
    links[0] = hs.PartitionLink("P0", "P1", min_latency=0.02, latency=hs.ConstantLatency(0.03))
    with pytest.raises(hs.UnsupportedTopology, match="sample"):
        hs.ParallelSimulation(parts, duration=2.0, links=links)


def test_probe_data_outlives_the_simulation_that_produced_it():
    """Probe samples of a large plain-chain run stay on the device until a Data is first read; a Data read after the Simulation
    (and its Sinks) are gone still finds the engine -- every unread Data keeps it alive -- and reads what an eager
    read would have."""
    import gc

    def build(n):
        sinks = [hs.Sink(f"k{i}") for i in range(n)]
        servers = [hs.Server(f"s{i}", service_time=hs.ExponentialLatency(0.05), downstream=sinks[i]) for i in range(n)]
        sources = [hs.Source.poisson(rate=9, target=servers[i], name=f"src{i}") for i in range(n)]
        probes = [hs.Probe.on(servers[i], "depth", interval=0.25) for i in range(n)]
        sim = hs.Simulation(duration=5.0, sources=sources, entities=[e for p in zip(servers, sinks) for e in p],
                            probes=[p for p, _ in probes], seed=13)
        return sim, [d for _, d in probes]

    sim, data = build(320)                       # more than 256 probes: lazy
    sim.run()
    want7, want300 = list(data[7].raw_values()), list(data[300].raw_values())
    assert len(want7) == 20
    sim2, data2 = build(320)
    sim2.run()
    keep = [data2[7], data2[300]]
    del sim2, data2
    gc.collect()
    assert list(keep[0].raw_values()) == want7 and list(keep[1].raw_values()) == want300
    small, dsmall = build(80)                    # a few probes: read at once, nothing pins the engine
    small.run()
    assert small._records._keep is False and dsmall[3].count() == 20 and dsmall[3]._lazy is None


def test_results_bind_one_object_at_a_time_until_somebody_walks_them():
    """Round 5: after `run()` on n plain chains nothing is bound; the first read of ONE Server's / Sink's result
    binds that chain alone (an identity search in the run's lists, entities._resolve) and gathers that Sink's column on the device
    (LazyRecords.records) -- not 4 x n attribute stores and a download of every record; a caller that walks the objects gets the bulk
    binding after a few lookups.  The values are the ones the bulk path gives."""
    from happy_simulator_amd import entities as E

    def build(n=2200):                                             # (> 4 096 entities: run() builds its entity summaries lazily too)
        sinks = [hs.Sink(f"sink{i}") for i in range(n)]
        servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sinks[i]) for i in range(n)]
        sources = [hs.Source.poisson(rate=8.0, target=servers[i], name=f"src{i}") for i in range(n)]
        sim = hs.Simulation(duration=20, sources=sources, entities=[e for pr in zip(servers, sinks) for e in pr], seed=9)
        return sinks, servers, sources, sim


    ref_sinks, ref_servers, ref_sources, ref_sim = build()             # the reference: the same run, everything bound at once
    ref_sim.run()
    E._flush_pending()
    assert not E._PENDING and all(s._bound is not None for s in ref_servers)
    sinks, servers, sources, sim = build()
    sim.run()
    assert len(E._PENDING) == 1 and all(s._bound is None for s in servers)
    done = servers[117].stats.requests_completed                       # one Server: its chain alone
    assert servers[117]._bound is not None and sources[117]._bound is not None and servers[3]._bound is None and len(E._PENDING) == 1
    lat = sinks[60].latencies_s                                        # one Sink: its column, gathered on the device
    assert sim._records._t is None and len(lat) == sinks[60].events_received > 50
    assert servers[60]._bound is not None and servers[61]._bound is None
    assert sources[5].generated_count > 100 and sources[5]._bound is not None
    assert done == ref_servers[117].stats.requests_completed and lat == ref_sinks[60].latencies_s
    assert sources[5].generated_count == ref_sources[5].generated_count
    total = sum(s.stats_accepted for s in servers)                     # walking the objects: the bulk binding after a few lookups
    assert not E._PENDING and all(s._bound is not None for s in servers)
    assert total == sum(s.stats_accepted for s in ref_servers)
    assert [k.events_received for k in sinks] == [k.events_received for k in ref_sinks]
    assert sinks[7].latencies_s == ref_sinks[7].latencies_s
