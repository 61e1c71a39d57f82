"""CPU: the host-side mirror of the reference API -- constructor contracts, error behaviour, graph lowering,
and the multi-rank host logic (world_size 2, gloo).  Modelled on the reference's own API tests
(tests/unit/test_ergonomic_api.py, tests/unit/test_source_factories.py, tests/unit/test_partition_link.py)."""
import os

import numpy as np
import pytest

import happy_simulator_amd as hs
from happy_simulator_amd import _native as N
from happy_simulator_amd import lowering as L
from happy_simulator_amd.core.temporal import Duration, Instant


class TestTemporal:
    def test_from_seconds_truncates(self):
        assert Instant.from_seconds(0.1).nanoseconds == 100_000_000
        assert Instant.from_seconds(1).nanoseconds == 1_000_000_000
        assert Instant.from_seconds(0.29).nanoseconds == int(0.29 * 1_000_000_000)
        assert Duration.from_seconds(1e-10).nanoseconds == 0
        with pytest.raises(TypeError):
            Instant.from_seconds("1")

    def test_arithmetic_and_ordering(self):
        t = Instant.from_seconds(5)
        assert (t + 0.5).nanoseconds == 5_500_000_000
        assert (t - Instant.from_seconds(2)) == Duration.from_seconds(3)
        assert (t + Duration(7)).nanoseconds == 5_000_000_007
        assert Instant.Epoch < t < Instant.Infinity
        assert Instant.Infinity >= Instant.Infinity and not (Instant.Infinity < t)
        assert Instant.from_seconds(1.5).to_seconds() == 1.5
        assert repr(Instant.from_seconds(3661.5)) == "T01:01:01.500000"


class TestSimulationContract:
    def test_duration_sets_end_time(self):
        sim = hs.Simulation(duration=50)
        assert sim._end_time == Instant.from_seconds(50)
        assert sim._start_time == Instant.Epoch

    def test_duration_conflicts_with_end_time(self):
        with pytest.raises(ValueError, match="Cannot specify both"):
            hs.Simulation(duration=50, end_time=Instant.from_seconds(100))

    def test_source_requires_target_or_provider(self):
        with pytest.raises(ValueError, match="Either 'target' or 'event_provider'"):
            hs.Source.constant(rate=10)
        with pytest.raises(ValueError, match="Either 'target' or 'event_provider'"):
            hs.Source.poisson(rate=10)

    def test_source_with_event_provider(self):
        sink = hs.Sink()
        src = hs.Source.poisson(rate=10, event_provider=hs.SimpleEventProvider(sink, stop_after=Instant.from_seconds(5)))
        assert src.downstream_entities() == [sink]

    def test_server_defaults_and_validation(self):
        s = hs.Server("S")
        assert s.concurrency == 1 and isinstance(s.service_time, hs.ConstantLatency) and s.service_time.mean == 0.01
        assert s.downstream is None and s.stats == hs.ServerStats(0, 0, 0.0)
        with pytest.raises(ValueError, match="max_concurrent must be >= 1"):
            hs.Server("S", concurrency=0)
        t = hs.LatencyTracker("T")
        s.downstream = t
        assert s.downstream is t and s.downstream_entities() == [t]

    def test_partition_link_requires_positive_min_latency(self):
        with pytest.raises(ValueError, match="min_latency must be > 0"):
            hs.PartitionLink("a", "b", min_latency=0.0)

    def test_no_gpu_is_a_loud_error(self):
        if N.lib().hs_device_count() > 0:
            pytest.skip("GPU present")
        sink = hs.Sink()
        srv = hs.Server("srv", service_time=hs.ExponentialLatency(0.1), downstream=sink)
        sim = hs.Simulation(duration=1, sources=[hs.Source.poisson(rate=8, target=srv)], entities=[srv, sink])
        with pytest.raises(hs.EngineUnavailable):
            sim.run()


class TestLowering:
    def _chain(self, i, **kw):
        sink = hs.Sink(f"sink{i}")
        srv = hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sink, **kw)
        return hs.Source.poisson(rate=8, target=srv, name=f"src{i}"), srv, sink

    def test_chains_become_stations_in_source_order(self):
        chains = [self._chain(i) for i in range(3)]
        sim = hs.Simulation(duration=10, sources=[c[0] for c in chains], entities=[e for c in chains for e in c[1:]])
        a = sim.lowered().arrays()
        assert a.n == 3
        assert list(a.src_kind) == [N.SRC_POISSON] * 3 and list(a.svc_kind) == [N.LAT_EXPONENTIAL] * 3
        assert list(a.queue_cap) == [-1] * 3 and list(a.egress) == [N.EGRESS_SINK] * 3

    def test_parameters_are_carried(self):
        sink = hs.Counter("c")
        srv = hs.Server("s", concurrency=3, service_time=hs.ConstantLatency(0.25), queue_capacity=7, downstream=sink)
        src = hs.Source.constant(rate=20, target=srv, stop_after=4.5)
        a = hs.Simulation(duration=10, sources=[src], entities=[srv, sink]).lowered().arrays()
        assert (a.src_kind[0], a.src_rate[0], a.src_stop_after_ns[0]) == (N.SRC_CONSTANT, 20.0, 4_500_000_000)
        assert (a.concurrency[0], a.svc_kind[0], a.svc_mean_s[0], a.queue_cap[0]) == (3, N.LAT_CONSTANT, 0.25, 7)

    def test_source_to_counter_and_server_without_downstream(self):
        counter = hs.Counter()
        src = hs.Source.constant(rate=1, target=counter)
        lone = hs.Server("lone")
        a = hs.Simulation(duration=60, sources=[src], entities=[counter, lone]).lowered().arrays()
        assert a.n == 2
        assert a.svc_kind[0] == N.LAT_NO_SERVER and a.egress[0] == N.EGRESS_SINK
        assert a.src_kind[1] == N.SRC_NONE and a.egress[1] == N.EGRESS_NONE

    def test_unsupported_graphs_are_refused_explicitly(self):
        class Custom(hs.Entity):
            pass

        sink = hs.Sink()
        s1 = hs.Server("a", downstream=sink)
        s2 = hs.Server("b", downstream=sink)
        # one collector behind several servers IS lowered: per-station logs, merged by time on the device at write-back
        g = hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=s1), hs.Source.poisson(1, target=s2)],
                          entities=[s1, s2, sink]).lowered()
        assert [st.sink for st in g.stations] == [sink, sink]
        pr, _ = hs.Probe.on(sink, "events_received")       # a probe on such a Sink ticks on the first station that feeds it
        g = hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=s1), hs.Source.poisson(1, target=s2)],
                          entities=[s1, s2, sink], probes=[pr]).lowered()
        assert g.stations[0].probes == (pr,) and g.stations[1].probes == () and g.shared_sink_probes == (pr,)
        with pytest.raises(hs.UnsupportedTopology, match="only Server / Sink"):
            hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=Custom("x"))]).lowered()
        with pytest.raises(hs.UnsupportedTopology, match="not lowered"):
            hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=hs.Sink())], entities=[Custom("x")]).lowered()
        t2 = hs.Server("t2")
        tandem = hs.Server("t1", downstream=t2)
        with pytest.raises(hs.UnsupportedTopology, match="not listed in `entities`"):        # (the reference would leave t2 without a clock)
            hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=tandem)]).lowered()
        # tandem queues ARE lowered: one station per Server, the upstream one's egress is the downstream one's station
        a = hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=tandem)], entities=[tandem, t2]).lowered().arrays()
        assert a.egress.tolist() == [N.EGRESS_SERVER, N.EGRESS_NONE] and a.downstream_lp.tolist() == [1, -1]
        assert a.src_kind.tolist() == [N.SRC_POISSON, N.SRC_NONE]
        t3 = hs.Server("t3", downstream=t2)                  # several upstream Servers per Server: lowered too (the engine's single heap)
        a = hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=tandem), hs.Source.poisson(1, target=t3)],
                          entities=[tandem, t3, t2]).lowered().arrays()
        assert a.downstream_lp.tolist() == [2, 2, -1]
        with pytest.raises(hs.UnsupportedTopology, match="not a lowered Probe"):
            hs.Simulation(duration=1, sources=[hs.Source.poisson(1, target=hs.Sink())], probes=[object()]).lowered()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [hs.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _rank_main(rank, world, port, q):
    """One rank of the N>1 host path on CPU: shard the replicas, run the local shard, reduce the totals.
    The C oracle stands in for the GPU engine here (tests may use it as the local runner)."""
    import torch.distributed as dist

    from oracle import hs_oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_replicas, base_seed, end_ns = 11, 1000, 5_000_000_000
    lo, hi = hs.shard_range(n_replicas, world, rank)
    events = 0
    final = 0
    for i in range(lo, hi):
        r = O.run(O.mm1_chains(1), end_ns, seed=base_seed + i)
        events += r.events_processed
        final = max(final, r.final_time_ns)
    tot = hs.reduce_summaries({"events": events, "max_final_ns": final, "replicas": hi - lo})
    if rank == 0:
        q.put(tot)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction_gloo():
    import torch.multiprocessing as mp

    from oracle import hs_oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tot = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_events, want_final = 0, 0
    for i in range(11):
        r = O.run(O.mm1_chains(1), 5_000_000_000, seed=1000 + i)
        want_events += r.events_processed
        want_final = max(want_final, r.final_time_ns)
    assert tot == {"events": want_events, "max_final_ns": want_final, "replicas": 11}


class TestLoadBalancerLowering:
    """Host side of BASELINE configs[4]: constructors mirror components/load_balancer (load_balancer.py:83-214,
    strategies.py:336-357) incl. their ValueErrors; the graph is lowered to `hs_lb_sources` / `hs_lb_backends` arrays."""

    def _topology(self, n_src=3, n_be=4, shared=True, **server_kw):
        sink = hs.Sink("sink")
        sinks = [sink] * n_be if shared else [hs.Sink(f"sink{j}") for j in range(n_be)]
        nodes = [hs.Server(f"srv{j}", service_time=hs.ExponentialLatency(0.1), downstream=sinks[j], **server_kw)
                 for j in range(n_be)]
        lb = hs.LoadBalancer("Router", backends=nodes, strategy=hs.ConsistentHash(virtual_nodes=150))
        srcs = [hs.Source.poisson(rate=10 + i, name=f"src{i}",
                                  event_provider=hs.ClientKeyEventProvider(lb, n_clients=200, stop_after=90.0))
                for i in range(n_src)]
        return srcs, lb, nodes, sinks

    def test_constructor_validation(self):
        with pytest.raises(ValueError, match="virtual_nodes must be >= 1"):
            hs.ConsistentHash(virtual_nodes=0)
        with pytest.raises(ValueError, match="on_no_backend must be"):
            hs.LoadBalancer("lb", strategy=hs.ConsistentHash(), on_no_backend="drop")
        lb = hs.LoadBalancer("lb", strategy=hs.ConsistentHash())
        with pytest.raises(ValueError, match="weight must be >= 1"):
            lb.add_backend(hs.Server("s"), weight=0)
        with pytest.raises(ValueError, match="n_clients must be >= 1"):
            hs.ClientKeyEventProvider(lb, n_clients=0)
        assert isinstance(hs.LoadBalancer("lb").strategy, hs.RoundRobin)          # the reference's default (load_balancer.py:112)
        with pytest.raises(NotImplementedError, match="is not lowered"):
            hs.LoadBalancer("lb", strategy=object())
        lb.add_backend(hs.Server("s"))
        lb.add_backend(hs.Server("t"), weight=3)
        assert lb.backend_count == lb.healthy_count == 2 and [b.name for b in lb.all_backends] == ["s", "t"]
        assert lb.get_backend_info_by_name("t").weight == 3 and lb.stats.requests_received == 0
        assert [e.name for e in lb.downstream_entities()] == ["s", "t"]

    def test_graph_is_lowered_to_engine_arrays(self):
        srcs, lb, nodes, sinks = self._topology(concurrency=3, queue_capacity=5)
        g = hs.Simulation(duration=100.0, sources=srcs, entities=[lb, *nodes, sinks[0]]).lowered()
        src, be = g.engine_arrays()
        assert g.shared_sink and src.n == 3 and be.n == 4 and be.names == ["srv0", "srv1", "srv2", "srv3"]
        assert list(src.src_rate) == [10.0, 11.0, 12.0] and list(src.n_clients) == [200] * 3
        assert list(src.src_stop_after_ns) == [90_000_000_000] * 3 and list(src.src_kind) == [N.SRC_POISSON] * 3
        assert list(be.concurrency) == [3] * 4 and list(be.queue_cap) == [5] * 4
        assert list(be.svc_kind) == [N.LAT_EXPONENTIAL] * 4 and list(be.egress) == [N.EGRESS_SINK] * 4
        srcs, lb, nodes, sinks = self._topology(shared=False)
        g = hs.Simulation(duration=1.0, sources=srcs, entities=[lb, *nodes, *sinks]).lowered()
        assert not g.shared_sink and len(g.sinks) == 4

    def test_unsupported_lb_graphs_are_refused_explicitly(self):
        srcs, lb, nodes, sinks = self._topology()
        plain = hs.Source.poisson(rate=5, target=lb)
        from happy_simulator_amd.graph_engine import GeneralGraph
        with pytest.raises(hs.UnsupportedTopology, match="ClientKeyEventProvider"):
            L.lower_lb([plain], [lb, *nodes, sinks[0]], lb)
        # (key-less Requests at a ConsistentHash LoadBalancer take the strategy's fallback RoundRobin: the single-heap path runs them)
        sim = hs.Simulation(duration=1, sources=[plain], entities=[lb, *nodes, sinks[0]])
        assert isinstance(sim.lowered(), GeneralGraph) and "ClientKeyEventProvider" in sim._station_refusal
        # outside the pipeline's shape: lower_lb refuses by name, Simulation takes the graph to the single-heap path (round 6)
        nodes[1].downstream = hs.Sink("other")
        with pytest.raises(hs.UnsupportedTopology, match="share ONE Sink"):
            L.lower_lb(srcs, [lb, *nodes], lb)
        sim = hs.Simulation(duration=1, sources=srcs, entities=[lb, *nodes])
        assert isinstance(sim.lowered(), GeneralGraph) and "share ONE Sink" in sim._station_refusal
        srcs, lb, nodes, sinks = self._topology()
        with pytest.raises(hs.UnsupportedTopology, match="not part of the load-balancer topology"):
            L.lower_lb(srcs, [lb, *nodes, sinks[0], hs.Server("stray")], lb)
        assert isinstance(hs.Simulation(duration=1, sources=srcs, entities=[lb, *nodes, sinks[0], hs.Server("stray")]).lowered(), GeneralGraph)
        lb2 = hs.LoadBalancer("lb2", backends=[hs.Sink("k")], strategy=hs.ConsistentHash())
        s2 = hs.Source.poisson(rate=1, event_provider=hs.ClientKeyEventProvider(lb2, n_clients=5))
        with pytest.raises(hs.UnsupportedTopology, match="only Server backends"):
            hs.Simulation(duration=1, sources=[s2], entities=[lb2]).lowered()

    def test_probes_on_backend_servers_and_sinks_are_lowered(self):
        """Probe.on(<backend Server> | <Sink>, ...) on a load-balancer graph -> hs_lb_set_probes arrays in `probes=[...]` order;
        other targets are refused explicitly."""
        srcs, lb, nodes, sinks = self._topology(shared=False, concurrency=2)
        p1, _ = hs.Probe.on(nodes[2], "depth", interval=0.5)
        p2, _ = hs.Probe.on(sinks[1], "events_received", interval=0.25)
        p3, _ = hs.Probe.on(nodes[0], "utilization", interval=1.0)
        g = hs.Simulation(duration=5.0, sources=srcs, entities=[lb, *nodes, *sinks], probes=[p1, p2, p3]).lowered()
        kinds, idx, met, iv = g.probe_arrays()
        assert kinds == [0, 1, 0] and idx == [2, 1, 0] and iv == [0.5, 0.25, 1.0]
        assert met == [N.PROBE_METRICS["depth"], N.PROBE_METRICS["events_received"], N.PROBE_METRICS["active_requests"]]
        from happy_simulator_amd.graph_engine import GeneralGraph
        pr, _ = hs.Probe.on(srcs[0], "generated_count")        # (a Source with stop_after keeps ticking: the pipeline's tick log does not
        sim = hs.Simulation(duration=1, sources=srcs, entities=[lb, *nodes, *sinks], probes=[pr])      # hold that; the single heap does)
        assert isinstance(sim.lowered(), GeneralGraph) and "a Source with stop_after keeps ticking" in sim._station_refusal
        for target, metric, msg in ((lb, "depth", "the Sources, the backend Servers and the Sinks are sampled"),
                                    (nodes[1], "events_received", "not an attribute of Server"),
                                    (sinks[0], "depth", "not an attribute of Sink")):
            pr, _ = hs.Probe.on(target, metric)
            with pytest.raises(hs.UnsupportedTopology, match=msg):
                hs.Simulation(duration=1, sources=srcs, entities=[lb, *nodes, *sinks], probes=[pr]).lowered()

    def test_product_md5_is_rfc1321(self):
        """The ring's hash function as libhs_hip.so computes it (host code of the library; no GPU involved)."""
        import hashlib

        from happy_simulator_amd.lb_engine import md5

        assert md5(b"").hex() == "d41d8cd98f00b204e9800998ecf8427e"
        assert md5(b"message digest").hex() == "f96b697d7cb7938d525a2f31aaf161d0"
        for n in (1, 9, 55, 56, 57, 63, 64, 65, 119, 120, 121, 250):
            msg = bytes((7 * i + n) & 0xFF for i in range(n))
            assert md5(msg) == hashlib.md5(msg).digest()


def test_profiles_are_lowered_and_unknown_profiles_refused():
    """Source.with_profile (load/source.py:271-320) with the reference's parametric profiles (load/profile.py:38-113)."""
    ramp = hs.LinearRampProfile(duration_s=10.0, start_rate=5.0, end_rate=30.0)
    spike = hs.SpikeProfile(baseline_rate=10.0, spike_rate=150.0, warmup_s=3.0, spike_duration_s=2.0)
    assert ramp.get_rate(Instant.from_seconds(5.0)) == 17.5 and ramp.get_rate(Instant.from_seconds(11.0)) == 30.0
    assert spike.get_rate(Instant.from_seconds(2.9)) == 10.0 and spike.get_rate(Instant.from_seconds(3.0)) == 150.0
    s1 = hs.Source.with_profile(ramp, target=hs.Server("a", downstream=hs.Sink("ka")), name="s1")
    s2 = hs.Source.with_profile(spike, target=hs.Server("b", downstream=hs.Sink("kb")), poisson=False, name="s2")
    s3 = hs.Source.poisson(rate=4, target=hs.Server("c", downstream=hs.Sink("kc")), name="s3")
    a = hs.Simulation(duration=20, sources=[s1, s2, s3]).lowered().arrays()
    assert list(a.src_profile_kind) == [N.PROF_LINEAR_RAMP, N.PROF_SPIKE, N.PROF_CONSTANT]
    assert a.src_profile_params[0].tolist() == [10.0, 5.0, 30.0, 0.0]
    assert a.src_profile_params[1].tolist() == [10.0, 150.0, 3.0, 2.0]
    assert list(a.src_kind) == [N.SRC_POISSON, N.SRC_CONSTANT, N.SRC_POISSON]
    assert list(a.src_rate) == [30.0, 150.0, 4.0]              # peak rates size the record logs
    with pytest.raises(ValueError, match="target"):
        hs.Source.with_profile(ramp)

    class Custom:
        def get_rate(self, t):
            return 1.0

    with pytest.raises(NotImplementedError, match="arbitrary Python"):
        hs.Source.with_profile(Custom(), target=hs.Sink())


def test_probes_are_lowered_onto_their_station():
    """Probe.on(target, metric, interval) (instrumentation/probe.py:81-164) -> hs_stations.probe_metric / probe_interval_s."""
    sink = hs.Sink("k")
    srv = hs.Server("s", concurrency=2, service_time=hs.ExponentialLatency(0.1), downstream=sink)
    src = hs.Source.poisson(rate=8, target=srv, name="src")
    other = hs.Server("t", downstream=hs.Sink("k2"))
    src2 = hs.Source.poisson(rate=3, target=other, name="src2")
    p1, d1 = hs.Probe.on(srv, "depth", interval=0.5)
    p2, d2 = hs.Probe.on(other.downstream, "events_received", interval=2.0)
    sim = hs.Simulation(duration=10, sources=[src, src2], entities=[srv, sink, other, other.downstream], probes=[p1, p2])
    a = sim.lowered().arrays()
    assert list(a.probe_metric) == [N.PROBE_METRICS["depth"], N.PROBE_METRICS["events_received"]]
    assert list(a.probe_interval_s) == [0.5, 2.0] and p1.name == "Probe_s_depth" and len(d1) == 0
    with pytest.raises(ValueError, match="interval must be positive"):
        hs.Probe.on(srv, "depth", interval=0.0)
    with pytest.raises(NotImplementedError, match="arbitrary attribute"):
        hs.Probe.on(srv, "some_custom_attr")
    # Probe(start_time=...) is accepted and changes nothing, as in the reference (Source.start() overwrites the provider's clock,
    # load/source.py:127; tests/test_oracle_live_reference.py::test_live_reference_ignores_a_probe_start_time)
    late = hs.Probe(srv, "depth", hs.Data(), interval=0.5, start_time=hs.Instant.from_seconds(4.0))
    c = hs.Simulation(duration=10, sources=[src, src2], entities=[srv, sink, other, other.downstream], probes=[late, p2]).lowered().arrays()
    assert list(c.probe_metric) == list(a.probe_metric) and list(c.probe_interval_s) == list(a.probe_interval_s)
    assert late.start_time == hs.Instant.from_seconds(4.0)
    # several probes on one station: engine slots 0..3, in `probes=[...]` order
    more = [hs.Probe.on(t, m, interval=iv)[0] for t, m, iv in ((srv, "active_requests", 1.0), (sink, "events_received", 0.25),
                                                              (src, "generated_count", 3.0))]
    b = hs.Simulation(duration=1, sources=[src], entities=[srv, sink], probes=[p1] + more).lowered().arrays()
    assert list(b.probe_metric) == [N.PROBE_METRICS["depth"]] and list(b.probe_interval_s) == [0.5]
    assert b.probe_metric_more[:, 0].tolist() == [N.PROBE_METRICS[m] for m in ("active_requests", "events_received", "generated_count")]
    assert b.probe_interval_more[:, 0].tolist() == [1.0, 0.25, 3.0]
    p5, _ = hs.Probe.on(srv, "stats_dropped")
    # a fifth probe on the station: beyond the station engines' four slots -- the single-heap path takes it (round 6: refused until then)
    from happy_simulator_amd.graph_engine import GeneralGraph
    five = hs.Simulation(duration=1, sources=[src], entities=[srv, sink], probes=[p1] + more + [p5])
    assert isinstance(five.lowered(), GeneralGraph) and "already has four probes" in five._station_refusal
    with pytest.raises(hs.UnsupportedTopology, match="already has four probes"):
        L.attach_probes(hs.Simulation(duration=1, sources=[src], entities=[srv, sink]).lowered(), [p1] + more + [p5])
    p4, _ = hs.Probe.on(sink, "depth")
    with pytest.raises(hs.UnsupportedTopology, match="not an attribute of Sink"):
        hs.Simulation(duration=1, sources=[src], entities=[srv, sink], probes=[p4]).lowered()
    probes, data = hs.Probe.on_many(srv, ["depth", "utilization"], interval=1.0)
    assert [p.metric for p in probes] == ["depth", "utilization"] and set(data) == {"depth", "utilization"}


def test_several_sources_of_one_server_are_lowered_onto_its_station():
    """Two or more Sources with the same target Server -> hs_stations.src_more_* (slots 1..3 of the station) and
    source_order / source_slot_order in `sources=[...]` order."""
    sink = hs.Sink("k")
    srv = hs.Server("s", concurrency=1, service_time=hs.ExponentialLatency(0.05), downstream=sink)
    other = hs.Server("t", downstream=hs.Sink("k2"))
    a = hs.Source.poisson(rate=6, target=srv, name="a")
    b = hs.Source.constant(rate=4, target=srv, name="b", stop_after=3.0)
    c = hs.Source.poisson(rate=2, target=srv, name="c")
    d = hs.Source.poisson(rate=3, target=other, name="d")
    sim = hs.Simulation(duration=5, sources=[b, d, a, c], entities=[srv, sink, other, other.downstream])
    g = sim.lowered()
    assert g.stations[0].source is b and list(g.stations[0].more_sources) == [a, c] and not g.stations[1].more_sources
    arr = g.arrays()
    assert arr.src_kind.tolist() == [N.SRC_CONSTANT, N.SRC_POISSON] and arr.src_rate.tolist() == [4.0, 3.0]
    assert arr.src_more_kind[:, 0].tolist() == [N.SRC_POISSON, N.SRC_POISSON, N.SRC_NONE]
    assert arr.src_more_rate[:2, 0].tolist() == [6.0, 2.0] and arr.src_more_kind[:, 1].tolist() == [N.SRC_NONE] * 3
    assert arr.src_stop_after_ns.tolist() == [3_000_000_000, -1] and arr.src_more_stop_after_ns[:, 0].tolist() == [-1, -1, -1]
    assert g.log_capacity(5.0) >= 12 * 5                     # the station admits what all three Sources generate
    more = [hs.Source.poisson(rate=1, target=srv, name=f"x{k}") for k in range(2)]
    # beyond the station's slots: lower() refuses by name, Simulation takes the graph to the single-heap path (round 6)
    from happy_simulator_amd.graph_engine import GeneralGraph
    with pytest.raises(hs.UnsupportedTopology, match="more than four Sources"):
        L.lower([a, b, c] + more, [srv, sink])
    five = hs.Simulation(duration=1, sources=[a, b, c] + more, entities=[srv, sink])
    assert isinstance(five.lowered(), GeneralGraph) and "more than four Sources" in five._station_refusal
    ramp = hs.Source.with_profile(hs.LinearRampProfile(duration_s=2.0, start_rate=1.0, end_rate=5.0), target=srv)
    with pytest.raises(hs.UnsupportedTopology, match="time-varying profile next to further Sources"):
        L.lower([a, ramp], [srv, sink])
    assert isinstance(hs.Simulation(duration=1, sources=[a, ramp], entities=[srv, sink]).lowered(), GeneralGraph)
    pr, _ = hs.Probe.on(a, "generated_count")
    sampled = hs.Simulation(duration=1, sources=[b, a], entities=[srv, sink], probes=[pr])
    assert isinstance(sampled.lowered(), GeneralGraph) and "only the first Source of a Server is sampled" in sampled._station_refusal


def test_schedule_is_lowered_to_per_station_time_lists():
    """Simulation.schedule() (core/simulation.py:195-206) -> hs_stations.sched_off / sched_time_ns: per station, ascending,
    ties in call order; cancelled events are kept out and counted; events before start_time are dropped with the
    reference's "time travel" warning."""
    import warnings

    k0, k1 = hs.Sink("k0"), hs.Sink("k1")
    s0 = hs.Server("s0", service_time=hs.ConstantLatency(0.1), downstream=k0)
    s1 = hs.Server("s1", service_time=hs.ExponentialLatency(0.1), downstream=k1)
    src = hs.Source.poisson(rate=5, target=s1, name="src")
    sim = hs.Simulation(start_time=hs.Instant.from_seconds(1.0), end_time=hs.Instant.from_seconds(9.0), sources=[src],
                        entities=[s0, k0, s1, k1])

    def ev(t, target):
        return hs.Event(time=hs.Instant.from_seconds(t), event_type="Request", target=target)

    evs = [ev(3.0, s0), ev(2.0, s1), ev(2.5, s0), ev(2.5, s0), ev(0.5, s0), ev(8.0, s1), ev(4.0, s0)]
    sim.schedule(evs[:4])
    sim.schedule(evs[4])
    sim.schedule(evs[5:])
    evs[6].cancel()
    assert evs[6].cancelled and not evs[0].cancelled and evs[0].context["created_at"] == evs[0].time
    g = sim.lowered()
    a = g.arrays()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cancelled = sim._schedule_arrays(g, a)
    assert any("Time travel" in str(x.message) for x in w)
    assert cancelled == [4_000_000_000]
    assert a.sched_off.tolist() == [0, 3, 5]
    assert a.sched_time_ns.tolist() == [2_500_000_000, 2_500_000_000, 3_000_000_000, 2_000_000_000, 8_000_000_000]
    with pytest.raises(TypeError, match="Event objects"):
        sim.schedule("Request")
    hooked = hs.Event(time=hs.Instant.from_seconds(2.0), event_type="Request", target=s0, on_complete=[lambda t: None])
    sim2 = hs.Simulation(duration=5, sources=[], entities=[s0, k0])
    sim2.schedule(hooked)
    with pytest.raises(hs.UnsupportedTopology, match="completion hooks"):
        sim2._schedule_arrays(sim2.lowered(), sim2.lowered().arrays())


def test_parallel_summary_formulas_and_layout():
    """ParallelSimulation._build_summary (parallel/simulation.py:225-284): speedup = sum of partition wall times / wall
    clock, efficiency = speedup / partitions, coordination efficiency = 1 - barrier / wall; entities are the merge of the
    partitions' entity summaries."""
    from happy_simulator_amd.parallel import _parallel_summary
    from happy_simulator_amd.summary import EntitySummary, QueueStats, SimulationSummary

    ea = EntitySummary("srv_a", "Server", 0, QueueStats(0, 5, 1))
    eb = EntitySummary("sink_b", "Sink", 9)
    parts = {"a": SimulationSummary(2.0, 10, entities={"srv_a": ea}), "b": SimulationSummary(3.0, 20, entities={"sink_b": eb})}
    s = _parallel_summary(parts, {"a": 0.5, "b": 0.25}, 2.0, duration_s=3.0, total_events=30, n_partitions=2, windows=4,
                          cross=7, window_s=0.001, barrier_s=0.5)
    assert (s.speedup, s.parallelism_efficiency, s.coordination_efficiency) == (0.375, 0.1875, 0.75)
    assert s.events_per_second == 10.0 and list(s.entities) == ["srv_a", "sink_b"]
    d = s.to_dict()
    assert list(d) == ["duration_s", "total_events_processed", "events_per_second", "wall_clock_seconds", "partitions",
                       "entities", "partition_wall_times", "speedup", "parallelism_efficiency", "total_windows",
                       "total_cross_partition_events", "window_size_s", "barrier_overhead_seconds", "coordination_efficiency"]
    assert d["entities"]["srv_a"]["queue"] == {"peak_depth": 0, "total_accepted": 5, "total_dropped": 1}
    assert "Barrier overhead: 0.500s" in str(s) and "Speedup: 0.38x" in str(s)
    quiet = _parallel_summary(parts, {"a": 1.0, "b": 1.0}, 1.0, duration_s=3.0, total_events=30, n_partitions=2)
    assert "Windows" not in str(quiet) and quiet.speedup == 2.0 and quiet.parallelism_efficiency == 1.0


@pytest.mark.live_reference
def test_parallel_summary_mirror_equals_the_live_reference_class():
    """Same constructor keywords, `to_dict()` and `__str__` as happysimulator/parallel/summary.py, field for field."""
    import os
    import sys

    if not os.path.isdir("/root/reference/happysimulator"):
        pytest.skip("needs /root/reference (build container only)")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import refshim

    refshim.install()
    from happysimulator.instrumentation.summary import SimulationSummary as RefSS
    from happysimulator.parallel.summary import ParallelSimulationSummary as Ref

    from happy_simulator_amd.parallel import ParallelSimulationSummary as Mine
    from happy_simulator_amd.summary import SimulationSummary as MySS

    kw = dict(duration_s=12.5, total_events_processed=1234, events_per_second=98.72, wall_clock_seconds=0.25,
              partition_wall_times={"p0": 0.1, "p1": 0.12}, speedup=0.88, parallelism_efficiency=0.44, total_windows=125,
              total_cross_partition_events=77, window_size_s=0.1, barrier_overhead_seconds=0.01, coordination_efficiency=0.96)
    ref = Ref(partitions={"p0": RefSS(12.5, 600, 0, 48.0, 0.1), "p1": RefSS(12.4, 634, 0, 51.1, 0.12)}, **kw)
    mine = Mine(partitions={"p0": MySS(12.5, 600, 0, 48.0, 0.1), "p1": MySS(12.4, 634, 0, 51.1, 0.12)}, **kw)
    assert mine.to_dict() == ref.to_dict()
    assert str(mine) == str(ref)
    assert str(Mine(duration_s=1.0, total_events_processed=3)) == str(Ref(1.0, 3, 0.0, 0.0))       # defaults, no windows
    # ... and the per-run summary with its entity / queue lines (instrumentation/summary.py:14-87)
    from happysimulator.instrumentation.summary import EntitySummary as RefES, QueueStats as RefQS

    from happy_simulator_amd.summary import EntitySummary as MyES, QueueStats as MyQS

    def build(SS, ES, QS):
        return SS(duration_s=60.000000001, total_events_processed=4321, events_cancelled=2, events_per_second=72.016,
                  wall_clock_seconds=0.0123, entities={"srv": ES("srv", "Server", 0, QS(0, 480, 3)), "Sink": ES("Sink", "Sink", 477)})
    a, b = build(MySS, MyES, MyQS), build(RefSS, RefES, RefQS)
    assert a.to_dict() == b.to_dict() and list(a.to_dict()) == list(b.to_dict())
    assert str(a) == str(b)
    assert str(MySS(1.5, 10)) == str(RefSS(1.5, 10)) and MySS(1.5, 10).to_dict() == RefSS(1.5, 10).to_dict()


def test_scheduled_requests_carry_their_construction_rank():
    """Simulation.schedule(): several Requests for one Server at exactly the start time used to be announced as a tie-break
    deviation; the engine now replays the reference's two sort counters (csrc/hs_exact.hpp) from the construction ranks --
    no warning, and a cancelled Event keeps its rank (it consumed a sort index) without being handed to the engine."""
    import warnings

    from happy_simulator_amd.engine import StationArrays

    sink = hs.Sink()
    srv = hs.Server("srv", service_time=hs.ConstantLatency(0.1), downstream=sink)
    sim = hs.Simulation(end_time=Instant.from_seconds(1.0), sources=[], entities=[srv, sink])
    evs = [hs.Event(time=Instant.from_seconds(t), event_type="Request", target=srv) for t in (0.0, 0.5, 0.5, 0.0, 0.25)]
    for ev in evs:
        sim.schedule(ev)
    evs[2].cancel()
    g = sim.lowered()
    arrays = g.arrays()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        cancelled = sim._schedule_arrays(g, arrays)
    assert isinstance(arrays, StationArrays) and cancelled == [500_000_000]
    assert arrays.sched_time_ns.tolist() == [0, 0, 250_000_000, 500_000_000]
    assert arrays.sched_rank.tolist() == [0, 3, 4, 1]


def test_log_capacity_follows_deep_tandems_and_cycles():
    """Round-1 advisor finding: the per-station record capacity looked only 4 hops upstream, so the last station of a 10-station
    tandem (every station with its own 100/s Source, everything forwarded) overflowed.  The inflow is now solved to its fixed
    point, capped by what each sender can serve."""
    n = 10
    servers = [hs.Server(f"s{i}", service_time=hs.ExponentialLatency(0.0005)) for i in range(n)]
    for i in range(n - 1):
        servers[i].downstream = hs.NetworkLink(f"l{i}", latency=hs.ConstantLatency(0.001), egress=servers[i + 1])
    sources = [hs.Source.poisson(rate=100, target=servers[i], name=f"src{i}") for i in range(n)]
    g = hs.Simulation(duration=10, sources=sources, entities=servers).lowered()
    assert g.log_capacity(10.0) > 10 * 100 * n                 # the last station admits ~ n x 100 x 10 s records
    # a ring where every router forwards half: inflow = 4 / (1 - 1/2) = 8 per station
    sinks = [hs.Sink(f"k{i}") for i in range(4)]
    ring = [hs.Server(f"r{i}", service_time=hs.ExponentialLatency(0.05)) for i in range(4)]
    for i in range(4):
        ring[i].downstream = hs.RandomRouter(f"rt{i}", targets=[sinks[i], hs.NetworkLink(f"rl{i}", latency=hs.ConstantLatency(0.001),
                                                                                          egress=ring[(i + 1) % 4])])
    g2 = hs.Simulation(duration=10, sources=[hs.Source.poisson(rate=4, target=r) for r in ring], entities=ring + sinks).lowered()
    assert 8 * 10 < g2.log_capacity(10.0) < 8 * 10 + 10 * 9 + 70 + 8
    sim = hs.Simulation(duration=10, sources=[], entities=[hs.Server("x")], log_capacity=123, bag_capacity=32, msg_capacity=512)
    assert (sim._log_capacity, sim._bag_capacity, sim._msg_capacity) == (123, 32, 512)


def test_probe_data_can_stay_on_the_device_until_first_read():
    """Data._set_lazy (lowering.write_back_plain_probes): one download on the first access, whichever accessor it is."""
    import numpy as np

    calls = []

    def fetch():
        calls.append(1)
        return np.array([1_000_000_000, 2_000_000_000], np.int64), np.array([3, 4], np.int64)

    d = hs.Data()
    d._set_lazy(fetch, scale=2)
    assert not calls
    assert d.count() == 2 and d.values == [(1.0, 1.5), (2.0, 2.0)] and d.max() == 2.0 and len(d) == 2
    assert calls == [1]
    d._set(np.array([5], np.int64), np.array([7], np.int64))
    assert d.values == [(5e-9, 7)]


def test_plain_probe_arrays_equal_the_per_station_lowering():
    """lowering.plain_probe_arrays (no Station object per chain) fills the same probe arrays, slots and `probes=[...]` order as
    attach_probes + LoweredGraph.arrays() do through Station objects."""
    import numpy as np

    from happy_simulator_amd import lowering as L

    def build():
        sinks = [hs.Sink(f"k{i}") for i in range(80)]
        servers = [hs.Server(f"s{i}", concurrency=1 + i % 3, service_time=hs.ExponentialLatency(0.05), downstream=sinks[i]) for i in range(80)]
        sources = [hs.Source.poisson(rate=5 + i % 4, target=servers[i], name=f"src{i}") for i in range(80)]
        probes = []
        for i in range(0, 80, 3):
            tgt, metric = [(servers[i], "depth"), (servers[i], "utilization"), (sinks[i], "events_received"),
                           (sources[i], "generated_count")][(i // 3) % 4]
            probes.append(hs.Probe.on(tgt, metric, interval=0.1 * (1 + i % 5))[0])
            if i % 2 == 0:
                probes += hs.Probe.on_many(servers[i], ["stats_accepted", "stats_dropped", "requests_completed"], interval=0.5)[0]
        return sources, servers + sinks, probes[::-1]

    sources, entities, probes = build()
    pc = L._plain_chains(sources, entities)
    assert pc is not None
    where = L.plain_probe_arrays(pc, probes, pc.arrays)
    fast = pc.arrays
    g = L.LoweredGraph(L._plain_chains(sources, entities))
    L.attach_probes(g, probes)
    slow = g.arrays()
    np.testing.assert_array_equal(fast.probe_metric, slow.probe_metric)
    np.testing.assert_array_equal(fast.probe_interval_s, slow.probe_interval_s)
    np.testing.assert_array_equal(fast.probe_metric_more, slow.probe_metric_more)
    np.testing.assert_array_equal(fast.probe_interval_more, slow.probe_interval_more)
    at = {id(pr): (i, slot) for i, st in enumerate(g.stations) for slot, pr in enumerate(st.probes)}
    assert [(w[0], w[1]) for w in where] == [at[id(p)] for p in probes]
    np.testing.assert_array_equal(fast.probe_order, [at[id(p)][0] for p in probes])
    np.testing.assert_array_equal(fast.probe_slot_order, [at[id(p)][1] for p in probes])
    assert [w[2] for w in where] == [p.target.concurrency if p.metric == "utilization" else None for p in probes]
    # Server attributes that are functions of active_requests: one engine counter, three value maps
    caps = hs.Probe.on_many(entities[0], ["available_capacity", "has_capacity", "utilization"], interval=0.5)[0]
    assert {hs.Probe.engine_metric(p.metric) for p in caps} == {"active_requests"}
    f = [hs.Probe.value_map(p.metric, hs.Server("c3", concurrency=3)) for p in caps]
    assert [f[0](a) for a in range(4)] == [3, 2, 1, 0] and [f[1](a) for a in range(4)] == [True, True, True, False] and f[2] == 3
    assert hs.Probe.value_map("depth", entities[0]) is None
    # what the fast path leaves to the general lowering / refuses like it
    other = hs.Server("elsewhere", service_time=hs.ExponentialLatency(0.1))
    assert L.plain_probe_arrays(pc, [hs.Probe.on(other, "depth")[0]], L.StationArrays.uniform(80)) is None
    with pytest.raises(hs.UnsupportedTopology, match="four probes"):
        L.plain_probe_arrays(pc, [hs.Probe.on(entities[0], "depth")[0] for _ in range(5)], L.StationArrays.uniform(80))
    with pytest.raises(hs.UnsupportedTopology, match="not an attribute"):
        L.plain_probe_arrays(pc, [hs.Probe.on(entities[0], "events_received")[0]], L.StationArrays.uniform(80))


def test_partition_link_is_the_reference_value_type():
    """parallel/link.py:18-79 and the reference's tests/unit/test_partition_link.py: defaults, the three ValueErrors, bidirectional()."""
    lk = hs.PartitionLink("A", "B", min_latency=0.01)
    assert lk.packet_loss == 0.0 and lk.latency is None
    with pytest.raises(ValueError, match="min_latency must be > 0"):
        hs.PartitionLink("A", "B", min_latency=0.0)
    with pytest.raises(ValueError, match="packet_loss must be in"):
        hs.PartitionLink("A", "B", min_latency=0.01, packet_loss=1.0)
    with pytest.raises(ValueError, match="packet_loss must be in"):
        hs.PartitionLink("A", "B", min_latency=0.01, packet_loss=-0.1)
    with pytest.raises(ValueError, match="source and dest must differ"):
        hs.PartitionLink("A", "A", min_latency=0.01)
    a_to_b, b_to_a = hs.PartitionLink.bidirectional("X", "Y", min_latency=0.1, packet_loss=0.05)
    assert (a_to_b.source_partition, a_to_b.dest_partition, b_to_a.source_partition, b_to_a.dest_partition) == ("X", "Y", "Y", "X")
    assert a_to_b.packet_loss == b_to_a.packet_loss == 0.05 and a_to_b.min_latency == b_to_a.min_latency == 0.1


def test_partition_link_losses_are_replayed_until_a_run_reproduces_them():
    """Host logic of `PartitionLink.packet_loss` without a GPU (happy_simulator_amd/parallel.py `_replay_partition_losses`): stand-in
    shards whose sends FOLLOW the loss decisions (a cycle back into the sending partition: what link 1 loses never comes back as a
    send of link 0) -- the loop must settle on the decisions of ONE `random.Random(seed)` drawn in send-time order, and the run it
    returns must be the one that ran with exactly those decisions."""
    import random
    import types

    from happy_simulator_amd.parallel import ParallelSimulation

    base = {0: [10 * k + 3 for k in range(40)], 1: [10 * k + 7 for k in range(40)]}     # link -> send times without any loss
    p_of = {0: 0.3, 1: 0.5}

    class Eng:
        def __init__(self, links):
            self.links, self.drops = links, {l: np.zeros(0, bool) for l in links}

        def set_link_drops(self, local, drops):
            self.drops[self.links[local]] = np.asarray(drops, bool).copy()

        def sends(self):
            d1 = self.drops.get(1, np.zeros(0, bool))
            out = {1: list(base[1])}
            # a packet of link 1 that survives comes back 4 ns later as an extra send of link 0
            extra = [t + 4 for e, t in enumerate(base[1]) if not (e < len(d1) and d1[e])]
            out[0] = sorted(base[0] + extra)
            return out

        def send_log(self):
            rows = [(t, l, e) for l, ts in self.sends().items() if l in self.links for e, t in enumerate(ts)]
            return np.array(rows, np.int64).reshape(-1, 3)

    class Comm:
        def gather_rows(self, arrays):
            return np.concatenate(arrays, axis=0)

    eng = Eng([0, 1])
    shard = types.SimpleNamespace(engine=eng, lo=0, hi=1, gids=np.array([0, 1]))
    runs = []
    sn = types.SimpleNamespace(shards=[shard], run_until=lambda end: runs.append({l: d.copy() for l, d in eng.drops.items()}) or len(runs))
    me = types.SimpleNamespace(_lossy_links=p_of, _seed=11, _graph=types.SimpleNamespace(links=[(None, 0, 5), (None, 0, 6)]))
    got = ParallelSimulation._replay_partition_losses(me, sn, Comm(), 0, 10**9)
    assert 2 <= got == len(runs) <= 45                       # the summary of the LAST run, which ran with the settled decisions
    # the fixed point: draw in send-time order over the sends of the settled run
    rng = random.Random(11)
    rows = sorted((t, l, e) for l, ts in eng.sends().items() for e, t in enumerate(ts))
    want = {0: [], 1: []}
    for t, l, e in rows:
        want[l].append(rng.random() < p_of[l])
    for l in (0, 1):
        d = np.zeros(len(want[l]), bool)
        d[:len(eng.drops[l])] = eng.drops[l][:len(want[l])]
        assert d.tolist() == want[l], l
    # two hops sending on one nanosecond: refused by name
    base[1][5] = base[0][5]
    with pytest.raises(hs.UnsupportedTopology, match="same nanosecond"):
        ParallelSimulation._replay_partition_losses(me, sn, Comm(), 0, 10**9)
