"""GPU: graphs outside the station shape on the device's single-heap loop (csrc/hs_graph.hip, happy_simulator_amd/graph_engine.py)
-- a RandomRouter with eight targets among Sinks, links and Servers, links with several senders, Servers behind Servers next to
links, seven Sources on one Server, routers with several upstreams -- through `hs.Simulation(...).run()`, against the LIVE-REFERENCE
goldens (tests/golden/graph_*.npz, make_golden.run_graph_case) and the C oracle on random graphs (random_specs.graph_spec, which
tests/test_oracle_live_reference.py pins on the live reference).  Bit-exact: totals, final time, every Source / Server / link /
router / Sink statistic, every Sink record."""
import numpy as np
import pytest

import graph_specs as GS
import happy_simulator_amd as hs
import helpers as H
from happy_simulator_amd import _native as N
from happy_simulator_amd.graph_engine import GeneralGraph, GraphEngine
from oracle import hs_oracle as O
from random_specs import graph_spec, lb_graph_spec

pytestmark = pytest.mark.gpu


def _compare_with_oracle(spec, sim, ents, r, nodes):
    assert sim.summary.total_events_processed == r.events_processed
    assert sim._current_time.nanoseconds == r.final_time_ns
    np.testing.assert_array_equal(list(sim._engine_summary.events_by_kind), r.events_by_kind)
    got = GS.results(ents)
    np.testing.assert_array_equal(got["generated"], r.generated[nodes["source"]])
    srv = nodes["server"]
    for k, arr in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed), ("rejected", r.rejected),
                   ("depth", r.depth), ("active", r.active), ("total_service_s", r.total_service_s)):
        np.testing.assert_array_equal(got[k], arr[srv], err_msg=k)
    if nodes["router"]:
        np.testing.assert_array_equal(got["routed"], r.routed[nodes["router"]])
    if nodes["link"]:
        np.testing.assert_array_equal(got["packets_sent"], r.packets_sent[nodes["link"]])
        np.testing.assert_array_equal(got["packets_dropped"], r.dropped[nodes["link"]])
    for j, nd in enumerate(nodes["sink"]):
        t, created = r.sinks[nd]
        np.testing.assert_array_equal(ents["sinks"][j].completion_ns, t, err_msg=f"sink {j}")
        np.testing.assert_array_equal(ents["sinks"][j]._created_ns, created, err_msg=f"sink {j} created_at")
    for rt in ents["routers"]:                                   # target_counts (random_router.py:37): consistent with what was routed
        assert sum(rt.target_counts.values()) == rt.stats_routed
    for j, nd in enumerate(nodes.get("lb", [])):                 # LoadBalancer.stats / BackendInfo.total_requests / RoundRobin._index
        lb, st = ents["lbs"][j], ents["lbs"][j].stats
        np.testing.assert_array_equal([st.requests_received, st.requests_forwarded, st.requests_failed, st.no_backend_available,
                                       lb._in_flight_count], r.lbs[nd]["stats"], err_msg=f"lb {j}")
        np.testing.assert_array_equal([lb.get_backend_info(b).total_requests for b in lb.all_backends], r.lbs[nd]["total_requests"],
                                      err_msg=f"lb {j} backends")
        kind = spec["lbs"][j]["strategy"]
        got_index = lb.strategy._index if kind == "round_robin" else lb.strategy._fallback._index if kind == "chash" else -1
        assert got_index == r.lbs[nd]["strategy_index"]


@pytest.mark.parametrize("name", H.golden_names("graph"))
def test_general_graphs_match_the_live_reference_goldens(name):
    gold = H.Golden(name)
    spec = gold.spec
    sim, ents = GS.build(spec)
    assert isinstance(sim.lowered(), GeneralGraph)               # (the station engines refuse these by name)
    summary = sim.run()
    assert [summary.total_events_processed] == gold.meta["total_events"]
    assert [sim._current_time.nanoseconds] == gold.meta["final_ns"]
    assert [summary.duration_s] == gold.meta["duration_s"]
    got = GS.results(ents)
    for k in ("generated", "accepted", "dropped", "completed", "rejected", "depth", "active", "total_service_s", "received", "routed",
              "packets_sent", "packets_dropped"):
        np.testing.assert_array_equal(got[k], gold.arrays[k], err_msg=k)
    for j, sk in enumerate(ents["sinks"]):
        gt, glat = gold.sink_records(j)
        np.testing.assert_array_equal(sk.completion_ns, gt, err_msg=f"sink {j}")
        np.testing.assert_array_equal(sk.latencies_array, glat, err_msg=f"sink {j} latencies")
        assert sk.latencies_s == list(glat)
    for j, lb in enumerate(ents["lbs"]):                          # (graph_three_load_balancers, graph_round_robin_schedule)
        st = lb.stats
        np.testing.assert_array_equal([st.requests_received, st.requests_forwarded, st.requests_failed, st.no_backend_available,
                                       lb._in_flight_count], gold.lb_stats[j], err_msg=f"lb {j}")
        lo, hi = gold.lb_backend_off[j], gold.lb_backend_off[j + 1]
        np.testing.assert_array_equal([lb.get_backend_info(b).total_requests for b in lb.all_backends],
                                      gold.lb_backend_total_requests[lo:hi], err_msg=f"lb {j} backends")
        assert (lb.strategy._fallback._index if isinstance(lb.strategy, hs.ConsistentHash) else getattr(lb.strategy, "_index", -1)) == gold.lb_rr_index[j]


@pytest.mark.parametrize("block", range(8))
def test_random_general_graphs_match_the_oracle(block):
    """40 graphs per block: up to 7 Servers, 5 shared links, 4 routers with up to 8 targets and several upstreams, up to 14 Sources."""
    ran = 0
    for k in range(block * 40, block * 40 + 40):
        spec = graph_spec(k)
        sim, ents = GS.build(spec)
        g = sim.lowered()
        if not isinstance(g, GeneralGraph):                      # (a draw the station engines take: theirs to test)
            continue
        g_o, nodes = H.oracle_graph(spec)
        r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"])
        sim.run()
        _compare_with_oracle(spec, sim, ents, r, nodes)
        ran += 1
    assert ran >= 30


@pytest.mark.parametrize("block", range(6))
def test_graphs_with_several_load_balancers_match_the_oracle(block):
    """random_specs.lb_graph_spec (pinned on the live reference by tests/test_oracle_live_reference.py): one to three LoadBalancers --
    ConsistentHash, RoundRobin, Random -- behind Sources, Servers and routers, Requests `schedule()`d on key-less graphs incl. for
    the LoadBalancers themselves (components/load_balancer/load_balancer.py:347-473, core/simulation.py:195-206)."""
    for k in range(block * 30, block * 30 + 30):
        spec = lb_graph_spec(k)
        sim, ents = GS.build(spec)
        assert isinstance(sim.lowered(), GeneralGraph)           # (lower_lb takes ONE LoadBalancer right behind every Source)
        g_o, nodes = H.oracle_graph(spec)
        r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"], schedule=H.oracle_graph_schedule(spec, nodes))
        sim.run()
        _compare_with_oracle(spec, sim, ents, r, nodes)
        assert sum(lb.stats.requests_forwarded for lb in ents["lbs"]) > 0


def test_key_less_requests_for_a_random_load_balancer_are_refused_by_name():
    """A Request without a client id at a Random LoadBalancer would ask the process-wide generator -- refused by name, for Sources
    that reach it through other entities and for `schedule()`d Requests.  (At a ConsistentHash LoadBalancer a key-less Request is
    the reference's own case -- the strategy's fallback RoundRobin: lb_graph_spec, the graph_keyless_consistent_hash fixture.)"""
    sink = hs.Sink("k")
    servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.01), downstream=sink) for i in range(3)]
    lb = hs.LoadBalancer("lb", backends=servers[1:], strategy=hs.Random())
    direct = hs.Source.poisson(rate=5, event_provider=hs.ClientKeyEventProvider(lb, n_clients=2), name="a")
    servers[0].downstream = lb
    around = hs.Source.poisson(rate=5, target=servers[0], name="b")
    with pytest.raises(hs.UnsupportedTopology, match="only Sources that aim at it directly"):
        hs.Simulation(duration=1, sources=[direct, around], entities=[*servers, lb, sink]).lowered()
    servers[0].downstream = sink
    sim = hs.Simulation(duration=1, sources=[direct, around], entities=[*servers, lb, sink])
    sim.schedule(hs.Event(time=hs.Instant.from_seconds(0.5), event_type="Request", target=lb))
    with pytest.raises(hs.UnsupportedTopology, match="carries no client id"):
        sim.run()
    sim = hs.Simulation(duration=1, sources=[direct, around], entities=[*servers, lb, sink])
    sim.schedule(hs.Event(time=hs.Instant.from_seconds(0.5), event_type="Request", target=servers[2]))     # (cannot reach the LoadBalancer)
    sim.run()
    assert 0 < lb.stats.requests_received <= direct.generated_count


def test_scheduled_requests_on_a_general_graph_take_the_pre_run_indices():
    """Simulation.schedule() (core/simulation.py:195-206) on a general graph: Requests for a Server, a router, a link and a Sink, some
    on the nanosecond of a Source's constant tick and of each other, one beyond the end -- the oracle's hso_schedule in call order."""
    spec = dict(H.Golden("graph_shared_links_tandem").spec)
    sched = [("server", 0, 0.5), ("server", 0, 0.5), ("router", 1, 0.5), ("link", 0, 1.0), ("sink", 2, 1.0), ("server", 3, 0.0),
             ("server", 0, 2.0), ("server", 4, 9.999999999), ("server", 1, 10.5), ("router", 0, 0.0)]
    sim, ents = GS.build(spec, extra_schedule=sched)
    g_o, nodes = H.oracle_graph(spec)
    r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"],
              schedule=[(nodes[k][i], H.ns_from_seconds(t)) for k, i, t in sched])
    sim.run()
    _compare_with_oracle(spec, sim, ents, r, nodes)


def test_windows_over_a_general_graph_equal_one_run():
    """`_run_window` = `_execute_until` again (core/simulation.py:527-541): growing, repeated and earlier ends on the engine handle."""
    spec = H.Golden("graph_fanout_8").spec
    sim, _ = GS.build(spec)
    g = sim.lowered()
    end_ns = H.ns_from_seconds(spec["end_s"])
    with GraphEngine(g.arrays, seed=spec["seed"]) as one:
        one.run_until(end_ns)
        s1, st1, rec1 = one.summary(), one.stats(), one.records()
    with GraphEngine(g.arrays, seed=spec["seed"], heap_capacity=1, request_capacity=1, record_capacity=16) as eng:   # (and every buffer grows)
        for w in (0.5, 0.5, 0.500000001, 3.25, 2.0, 7.75):
            eng.run_until(H.ns_from_seconds(w))
        eng.run_until(end_ns)
        s2, st2, rec2 = eng.summary(), eng.stats(), eng.records()
    assert s1.events_processed == s2.events_processed and s1.final_time_ns == s2.final_time_ns
    for k in st1:
        np.testing.assert_array_equal(st1[k], st2[k], err_msg=k)
    for a, b in zip(rec1, rec2):
        np.testing.assert_array_equal(a, b)


def test_replicas_of_a_general_graph_run_side_by_side():
    """ParallelRunner.run_replicas (parallel/runner.py:82-142: one worker process per replica, seeds base_seed + i) over a graph
    outside the station shape: 300 replicas in one launch of one workgroup each (hs_graph_run_many; more than the 256 the device
    holds at once) -- replica i == the oracle with seed base_seed + i, every statistic and Sink record."""
    spec = lb_graph_spec(7)
    built = []

    def build_fn():
        sim, ents = GS.build(spec)
        built.append((sim, ents))
        return sim

    results = hs.ParallelRunner().run_replicas(build_fn, 300, base_seed=1000)
    assert len(results) == len(built) == 300 and isinstance(built[0][0].lowered(), GeneralGraph)
    g_o, nodes = H.oracle_graph(spec)
    for i, ((sim, ents), res) in enumerate(zip(built, results)):
        r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=1000 + i, schedule=H.oracle_graph_schedule(spec, nodes))
        assert res.name == f"replica_{i}" and res.summary.total_events_processed == r.events_processed
        _compare_with_oracle(spec, sim, ents, r, nodes)
    assert len({res.summary.total_events_processed for res in results}) > 100        # (the seeds differ)


def test_a_sweep_over_different_kinds_of_simulations():
    """ParallelRunner.run_sweep over Simulations of every kind at once -- general graphs with different ends, a ring of stations (a
    network engine), a load-balancer pipeline, plain chains: each result == the same Simulation's own run() with that seed."""
    from random_specs import ring_spec
    from test_gpu_api import _build_ring

    def ring():
        spec = ring_spec(3)
        sources, servers, routers, links, sinks = _build_ring(spec)
        return hs.Simulation(end_time=hs.Instant.from_seconds(spec["end_s"]), sources=sources, entities=servers + routers + links + sinks)

    def general(k):
        return lambda: GS.build(lb_graph_spec(k) if k % 2 else graph_spec(k))[0]

    def chain(rate):
        def fn():
            sink = hs.Sink("k")
            srv = hs.Server("s", service_time=hs.ExponentialLatency(0.05), downstream=sink)
            return hs.Simulation(duration=5, sources=[hs.Source.poisson(rate=rate, target=srv)], entities=[srv, sink])
        return fn

    def lb_pipeline():
        sink = hs.Sink("k")
        servers = [hs.Server(f"srv{j}", service_time=hs.ExponentialLatency(0.05), downstream=sink) for j in range(4)]
        lb = hs.LoadBalancer("lb", backends=servers, strategy=hs.ConsistentHash(virtual_nodes=20))
        srcs = [hs.Source.poisson(rate=9, event_provider=hs.ClientKeyEventProvider(lb, n_clients=40), name=f"c{i}") for i in range(3)]
        return hs.Simulation(duration=4, sources=srcs, entities=[lb, *servers, sink])

    makers = [general(1), chain(7.0), general(4), ring, general(9), lb_pipeline, chain(3.0),
              general(12), general(1)]
    configs = [hs.RunConfig(name=f"c{i}", build_fn=fn, seed=500 + 3 * i) for i, fn in enumerate(makers)]
    results = hs.ParallelRunner().run_sweep(configs)
    assert [r.name for r in results] == [c.name for c in configs]
    for cfg, res in zip(configs, results):
        alone = cfg.build_fn()
        alone._seed = cfg.seed
        want = alone.run()
        assert res.summary.total_events_processed == want.total_events_processed > 0, cfg.name
        assert res.summary.duration_s == want.duration_s
        assert ({k: (e.entity_type, e.events_handled, e.queue_stats) for k, e in res.summary.entities.items()} ==
                {k: (e.entity_type, e.events_handled, e.queue_stats) for k, e in want.entities.items()}), cfg.name


def test_engines_run_in_a_batch_grow_their_buffers_and_continue():
    """hs_graph_run_many on handles created with the smallest capacities (every buffer grows, replicas finish in different launches),
    then again to a later end: each handle == the same graph run alone."""
    specs = [graph_spec(k) for k in (2, 3, 5)] + [lb_graph_spec(k) for k in (1, 2)]
    lowered = [GS.build(sp)[0].lowered() for sp in specs]
    assert all(isinstance(g, GeneralGraph) for g in lowered)
    ends = [1_500_000_000, 3_000_000_000]
    alone = []
    for sp, g in zip(specs, lowered):
        with GraphEngine(g.arrays, seed=sp["seed"]) as e:
            per_end = []
            for end in ends:
                e.run_until(end)
                per_end.append((e.summary().events_processed, e.summary().final_time_ns, e.stats(), e.records()))
            alone.append(per_end)
    engines = [GraphEngine(g.arrays, seed=sp["seed"], heap_capacity=1, request_capacity=1, record_capacity=16) for sp, g in zip(specs, lowered)]
    try:
        for w, end in enumerate(ends):
            GraphEngine.run_many(engines, end)
            for e, per_end in zip(engines, alone):
                ev, fin, st, rec = per_end[w]
                assert (e.summary().events_processed, e.summary().final_time_ns) == (ev, fin)
                got = e.stats()
                for k in st:
                    np.testing.assert_array_equal(got[k], st[k], err_msg=k)
                for a, b in zip(e.records(), rec):
                    np.testing.assert_array_equal(a, b)
        with pytest.raises(N.EngineError, match="listed twice"):
            GraphEngine.run_many([engines[0], engines[1], engines[0]], ends[-1])
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("block", range(4))
def test_a_simulation_of_several_disconnected_graphs_runs_as_parts(block):
    """Two to six graphs that share nothing in ONE Simulation: no Request crosses between them, so each part runs on a heap of its
    own, side by side (hs_graph_run_parts) -- and together they leave what the reference's ONE heap leaves (the oracle on the union):
    every statistic and record, the total, the single event beyond end_time.  A run whose order the parts cannot decide (a first tick
    or a schedule()d Request on the nanosecond of an event the run created) is repeated on one heap: same answer either way."""
    from random_specs import union_spec

    split = 0
    for k in range(block * 12, block * 12 + 12):
        rng = np.random.default_rng(91_000 + k)
        members = [(lb_graph_spec if rng.random() < 0.5 else graph_spec)(int(rng.integers(0, 5000))) for _ in range(int(rng.integers(2, 7)))]
        spec = union_spec(members, name=f"union_{k}")
        if k % 2 == 0:                                           # Poisson Sources only, nothing schedule()d: no tick shares a nanosecond
            spec["schedule"] = []
            for sc in spec["sources"]:
                sc["kind"] = "poisson"
        sim, ents = GS.build(spec)
        assert isinstance(sim.lowered(), GeneralGraph)
        g_o, nodes = H.oracle_graph(spec)
        r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"], schedule=H.oracle_graph_schedule(spec, nodes))
        sim.run()
        _compare_with_oracle(spec, sim, ents, r, nodes)
        split += sim._graph_parts >= 2
        assert k % 2 or sim._graph_parts >= 2
    assert split >= 6


def test_many_chains_beyond_the_station_shape_run_on_many_heaps():
    """6 000 independent chains of FIVE Poisson Sources -> Server(c = 40) -> Sink (the station engines stop at four Sources and
    c = 32): one Simulation, its 6 000 components on 2 048 heaps side by side, == the oracle's one heap."""
    n, per = 6000, 5
    g = O.Graph()
    src = [g.source(O.ARR_POISSON, 1.0 + (k % 5), stream_base=k) for k in range(n * per)]
    snk, srv = [], []
    for i in range(n):
        snk.append(g.sink())
    for i in range(n):
        srv.append(g.server(O.LAT_EXP, 0.3, concurrency=40, queue_cap=-1, stream_base=i))
        g.target[srv[i]] = snk[i]
    for k, s_ in enumerate(src):
        g.target[s_] = srv[k // per]
    end_ns = 2_000_000_000
    r = O.run(g, end_ns, seed=11)
    sinks = [hs.Sink(f"k{i}") for i in range(n)]
    servers = [hs.Server(f"s{i}", concurrency=40, service_time=hs.ExponentialLatency(0.3), downstream=sinks[i]) for i in range(n)]
    sources = [hs.Source.poisson(rate=1.0 + (k % 5), target=servers[k // per], name=f"src{k}") for k in range(n * per)]
    sim = hs.Simulation(end_time=hs.Instant.from_seconds(2.0), sources=sources, entities=servers + sinks, seed=11)
    summary = sim.run()
    assert sim._graph_parts == 2048                          # (6 000 components share 2 048 heaps)
    assert summary.total_events_processed == r.events_processed and sim._current_time.nanoseconds == r.final_time_ns
    np.testing.assert_array_equal([s.generated_count for s in sources], r.generated[src])
    np.testing.assert_array_equal([s._requests_completed for s in servers], r.completed[srv])
    np.testing.assert_array_equal([s._total_service_time for s in servers], r.total_service_s[srv])
    for i in (0, 1, n // 2, n - 1):
        np.testing.assert_array_equal(sinks[i].completion_ns, r.sinks[snk[i]][0])
        np.testing.assert_array_equal(sinks[i]._created_ns, r.sinks[snk[i]][1])


def test_parts_with_probes_and_time_varying_sources_match_the_oracle():
    """300 chains of five Sources -> Server(c = 40) -> Sink in ONE Simulation, a Probe on the Server or the Sink of two chains in three, a ramp instead of
    one Poisson Source in every tenth chain: every part builds its own tick tables (hs_tables.hip) and runs on a heap of its own --
    every sample, statistic and record == the oracle's one heap."""
    n, per = 300, 5
    g = O.Graph()
    src = []
    for k in range(n * per):
        ramp = (k % per == 0) and ((k // per) % 10 == 0)
        src.append(g.source(O.ARR_POISSON, 1.0 + (k % 4), stream_base=k, profile=("ramp", 2.0, 1.0, 6.0) if ramp else None))
    # (ONE Probe per chain: two Probes of one interval in a part put the second one's first tick -- numbered before the run -- on the
    #  nanosecond of the first one's sample -- numbered by the run: undecided, the one heap's case)
    probe_plan = [(i, "server" if i % 3 == 0 else "sink") for i in range(n) if i % 3 != 2]
    o_probes = [None] * len(probe_plan)
    snk = [None] * n
    srv = [None] * n
    # oracle node order = the product's: Sources, Probes, then entities (servers + sinks)
    placeholders = [g.probe(0, 0, 0.25) for _ in probe_plan]
    for i in range(n):
        srv[i] = g.server(O.LAT_EXP, 0.2, concurrency=40, queue_cap=-1, stream_base=i)
    for i in range(n):
        snk[i] = g.sink()
        g.target[srv[i]] = snk[i]
    for k, s_ in enumerate(src):
        g.target[s_] = srv[k // per]
    for j, (i, which) in enumerate(probe_plan):
        nd = placeholders[j]
        g.target[nd] = srv[i] if which == "server" else snk[i]
        g.probe_metric[nd] = 0 if which == "server" else 5                    # depth / events_received
        o_probes[j] = nd
    end_ns = 3_000_000_000
    r = O.run(g, end_ns, seed=23)
    sinks = [hs.Sink(f"k{i}") for i in range(n)]
    servers = [hs.Server(f"s{i}", concurrency=40, service_time=hs.ExponentialLatency(0.2), downstream=sinks[i]) for i in range(n)]
    sources = []
    for k in range(n * per):
        if (k % per == 0) and ((k // per) % 10 == 0):
            sources.append(hs.Source.with_profile(hs.LinearRampProfile(duration_s=2.0, start_rate=1.0, end_rate=6.0), target=servers[k // per],
                                                  name=f"src{k}"))
        else:
            sources.append(hs.Source.poisson(rate=1.0 + (k % 4), target=servers[k // per], name=f"src{k}"))
    probes = [hs.Probe.on(servers[i] if which == "server" else sinks[i], "depth" if which == "server" else "events_received", interval=0.25)
              for i, which in probe_plan]
    sim = hs.Simulation(end_time=hs.Instant.from_seconds(3.0), sources=sources, entities=servers + sinks, probes=[p for p, _ in probes], seed=23)
    assert isinstance(sim.lowered(), GeneralGraph)
    summary = sim.run()
    assert sim._graph_parts >= 2
    assert summary.total_events_processed == r.events_processed and sim._current_time.nanoseconds == r.final_time_ns
    np.testing.assert_array_equal([s.generated_count for s in sources], r.generated[src])
    np.testing.assert_array_equal([s._requests_completed for s in servers], r.completed[srv])
    np.testing.assert_array_equal([s._total_service_time for s in servers], r.total_service_s[srv])
    for i in range(0, n, 7):
        np.testing.assert_array_equal(sinks[i].completion_ns, r.sinks[snk[i]][0])
    for (pr, data), nd in zip(probes, o_probes):
        t, v = r.sinks[nd]
        np.testing.assert_array_equal(data._t_ns, t, err_msg=pr.name)
        np.testing.assert_array_equal(data._v, v, err_msg=pr.name)
        assert len(t) >= 11


def test_parts_that_cannot_decide_an_order_hand_the_run_to_one_heap():
    """Two constant Sources of one rate on one Server: the second one's first tick (numbered before the run) shares its nanosecond
    with the payload of the first (numbered by the run) -- which comes first depends on every event the Simulation created so far,
    other parts' included.  The part run reports it undecided; the one heap answers; == the oracle."""
    spec = dict(name="undecided", topology="graph", n_sinks=2, end_s=4.0, seed=5, links=[], routers=[],
                servers=[dict(mean=0.05, c=40, cap=None, out=["sink", 0]), dict(mean=0.05, c=40, cap=None, out=["sink", 1])],     # (c > 32: no station)
                sources=[dict(kind="poisson", rate=7.0, to=1), dict(kind="constant", rate=4.0, to=0), dict(kind="constant", rate=4.0, to=0)])
    sim, ents = GS.build(spec)
    assert isinstance(sim.lowered(), GeneralGraph)
    g_o, nodes = H.oracle_graph(spec)
    r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"])
    sim.run()
    assert sim._graph_parts == 1
    _compare_with_oracle(spec, sim, ents, r, nodes)


def test_a_heap_beyond_the_lds_window_and_a_large_concurrency():
    """6 000 Sources on one Server with concurrency 5 000 (the station engines stop at four Sources and c = 32): the heap holds more
    pending events than its 4 096 LDS entries, so sifts cross from LDS into HBM; == the oracle."""
    n_src = 6000
    g = O.Graph()
    src = [g.source(O.ARR_POISSON if k % 3 else O.ARR_CONSTANT, 2.0 + (k % 7), stream_base=k) for k in range(n_src)]
    snk = g.sink()
    srv = g.server(O.LAT_EXP, 0.4, concurrency=5000, queue_cap=-1, stream_base=0)
    for s in src:
        g.target[s] = srv
    g.target[srv] = snk
    end_ns = 1_000_000_000
    r = O.run(g, end_ns, seed=99)
    assert r.heap_peak > 4096 + 2000
    sink = hs.Sink("k")
    server = hs.Server("s", concurrency=5000, service_time=hs.ExponentialLatency(0.4), downstream=sink)
    sources = [(hs.Source.poisson if k % 3 else hs.Source.constant)(rate=2.0 + (k % 7), target=server, name=f"src{k}") for k in range(n_src)]
    sim = hs.Simulation(end_time=hs.Instant.from_seconds(1.0), sources=sources, entities=[server, sink], seed=99)
    summary = sim.run()
    assert summary.total_events_processed == r.events_processed
    assert sim._current_time.nanoseconds == r.final_time_ns
    np.testing.assert_array_equal([s.generated_count for s in sources], r.generated[src])
    assert server._requests_completed == r.completed[srv] and server._total_service_time == r.total_service_s[srv]
    assert server.active_requests == r.active[srv] and server.stats_accepted == r.accepted[srv]
    np.testing.assert_array_equal(sink.completion_ns, r.sinks[snk][0])
    np.testing.assert_array_equal(sink._created_ns, r.sinks[snk][1])


def test_a_run_beyond_max_graph_events_is_refused_by_name():
    spec = H.Golden("graph_fanout_8").spec
    sim, _ = GS.build(spec)
    sim._max_graph_events = 10
    with pytest.raises(hs.UnsupportedTopology, match="single-heap path"):
        sim.run()
    sim2, _ = GS.build(spec)
    g = sim2.lowered()
    with GraphEngine(g.arrays, seed=spec["seed"], max_events=500) as eng:
        with pytest.raises(N.EngineError, match="max_events"):
            eng.run_until(H.ns_from_seconds(spec["end_s"]))


def test_an_auto_terminating_station_network_runs_on_the_single_heap():
    """`Simulation(end_time=None)` over a ring of stations driven by schedule()d Requests only (core/simulation.py:304-322): the
    network engines run to a horizon and refuse it; the single-heap loop drains its heap like the reference's -- every Request
    circles until its router draws the Sink.  The graph itself is station-shaped (lower() takes it)."""
    n = 5
    spec = dict(n_sinks=n, end_s=None, seed=321,
                servers=[dict(mean=0.02 * (1 + i % 3), c=1 + i % 2, cap=None if i % 2 else 6, out=["router", i]) for i in range(n)],
                links=[dict(lat=0.001, jk="exp", jm=0.002, loss=0.05 if i == 2 else 0.0, to=(i + 1) % n) for i in range(n)],
                routers=[dict(targets=[["sink", i], ["link", i]]) for i in range(n)], sources=[])
    rng = np.random.default_rng(8)
    sched = [("server", int(rng.integers(0, n)), float(np.round(rng.uniform(0.0, 2.0), 3))) for _ in range(200)]
    sim, ents = GS.build(spec, extra_schedule=sched)
    from happy_simulator_amd.lowering import LoweredGraph
    assert isinstance(sim.lowered(), LoweredGraph) and sim.lowered().is_network
    g_o, nodes = H.oracle_graph(spec)
    r = O.run(g_o, 1 << 61, seed=spec["seed"], schedule=[(nodes[k][i], H.ns_from_seconds(t)) for k, i, t in sched])
    summary = sim.run()
    assert summary.total_events_processed == r.events_processed > 2000
    _compare_with_oracle(spec, sim, ents, r, nodes)
    assert all(s.depth == 0 and s.active_requests == 0 for s in ents["servers"])          # everything drained
    assert sum(k.events_received for k in ents["sinks"]) + sum(l.packets_dropped for l in ents["links"]) + sum(
        s.stats_dropped for s in ents["servers"]) == len(sched)


def _with_probes_and_profiles(k):
    """graph_spec(k) + Probes on its Servers / Sinks / Sources (several per target, shared and distinct intervals) and, on odd k,
    time-varying profiles on some Sources.  Returns (sim, ents, probes, oracle graph, oracle nodes, oracle probe nodes)."""
    spec = graph_spec(k)
    rng = np.random.default_rng(88_000 + k)
    profiles = {}
    if k % 2:
        for j in range(len(spec["sources"])):
            if rng.random() < 0.5:
                profiles[j] = (("ramp", float(rng.choice([1.0, 2.5])), float(rng.choice([0.0, 2.0])), float(rng.choice([6.0, 12.0])))
                               if rng.random() < 0.5 else
                               ("spike", float(rng.choice([1.0, 3.0])), float(rng.choice([10.0, 25.0])), float(rng.choice([0.5, 1.25])),
                                float(rng.choice([0.25, 0.5]))))
    sim, ents = GS.build(spec)
    for j, pf in profiles.items():                              # swap the Source for one with the profile (same name, target, kind)
        old = ents["sources"][j]
        prof = (hs.LinearRampProfile(duration_s=pf[1], start_rate=pf[2], end_rate=pf[3]) if pf[0] == "ramp" else
                hs.SpikeProfile(baseline_rate=pf[1], spike_rate=pf[2], warmup_s=pf[3], spike_duration_s=pf[4]))
        new = hs.Source.with_profile(prof, target=old._event_provider._target, poisson=spec["sources"][j]["kind"] == "poisson", name=old.name)
        ents["sources"][j] = new
    plan = []
    for _ in range(int(rng.integers(1, 7))):
        kind = str(rng.choice(["server", "server", "sink", "source"]))
        idx = int(rng.integers(0, len(ents[kind + "s"])))
        metric = (str(rng.choice(["depth", "active_requests", "stats_accepted", "stats_dropped", "requests_completed"])) if kind == "server"
                  else "events_received" if kind == "sink" else "generated_count")
        plan.append((kind, idx, metric, float(rng.choice([0.1, 0.25, 0.25, 0.5, 1.0]))))
    probes = [hs.Probe.on(ents[kind + "s"][idx], metric, interval=iv) for kind, idx, metric, iv in plan]
    sim = hs.Simulation(end_time=hs.Instant.from_seconds(spec["end_s"]), sources=ents["sources"],
                        entities=ents["servers"] + ents["routers"] + ents["links"] + ents["sinks"], probes=[p for p, _ in probes],
                        seed=spec["seed"])
    g_o, nodes = H.oracle_graph(spec)
    for j, pf in profiles.items():
        nd = nodes["source"][j]
        g_o.prof_kind[nd] = O.PROF_LINEAR_RAMP if pf[0] == "ramp" else O.PROF_SPIKE
        g_o.prof_p[nd] = tuple(pf[1:]) + (0.0,) * (5 - len(pf))
    codes = {"depth": 0, "active_requests": 1, "stats_accepted": 2, "stats_dropped": 3, "requests_completed": 4, "events_received": 5,
             "generated_count": 6}
    o_probes = [g_o.probe(nodes[kind][idx], codes[metric], iv) for kind, idx, metric, iv in plan]
    return spec, sim, ents, probes, g_o, nodes, o_probes


@pytest.mark.parametrize("block", range(4))
def test_probes_and_time_varying_sources_on_general_graphs_match_the_oracle(block):
    """Probes (instrumentation/probe.py:81-164: a tick chain through next_arrival_time's general path + daemon samples) and
    Source.with_profile ramps / spikes (load/profile.py:52-113) on graphs outside the station shape: tick tables from the
    cooperative kernel, events and samples on the single heap -- every sample (time, value), statistic and Sink record."""
    ran = 0
    for k in range(block * 20, block * 20 + 20):
        spec, sim, ents, probes, g_o, nodes, o_probes = _with_probes_and_profiles(k)
        if not isinstance(sim.lowered(), GeneralGraph):
            continue
        r = O.run(g_o, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"])
        sim.run()
        _compare_with_oracle(spec, sim, ents, r, nodes)
        for (pr, data), nd in zip(probes, o_probes):
            t, v = r.sinks[nd]
            np.testing.assert_array_equal(data._t_ns, t, err_msg=pr.name)
            np.testing.assert_array_equal(data._v, v, err_msg=pr.name)
            assert len(t) > 0
        ran += 1
    assert ran >= 15
