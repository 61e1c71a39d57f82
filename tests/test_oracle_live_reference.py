"""CPU, build container only (needs /root/reference): the oracle against the LIVE reference on randomly drawn
configurations -- beyond the 55 committed fixtures, which were produced by exactly the same code path
(tests/golden/make_golden.py: the reference's own event loop, queue protocol, Server generator and sort-index ledger with
per-entity Philox streams plugged in through its extension points).  Every count, statistic, Sink record and, for the
small cases, the full processed-event trace incl. `_sort_index`.  Skipped where the reference is absent (the GPU box)."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from test_oracle_golden import (check_oracle_against_graph_golden, check_oracle_against_lb_golden, check_oracle_against_ring_golden,
                                check_oracle_against_station_golden, check_oracle_against_tandem_golden)

pytestmark = pytest.mark.live_reference

if not os.path.isdir("/root/reference/happysimulator"):
    pytest.skip("needs /root/reference (build container only)", allow_module_level=True)

sys.path.insert(0, H.GOLDEN_DIR)
import make_golden as MG  # noqa: E402  (imports the reference through refshim)
from random_specs import (graph_spec, jitter_ring_spec, lb_graph_spec, lb_probe_spec, lb_profile_spec, lb_spec as _lb_spec, lb_strategy_spec, lb_workers_spec, multi_source_ring_spec, multi_source_spec, ring_spec as _ring_spec, station_spec as _station_spec,  # noqa: E402
                          tie_spec)


@pytest.mark.parametrize("k", range(40))
def test_oracle_equals_live_reference_on_random_station_specs(k):
    spec = _station_spec(k)
    if spec["mode"] == "replicas":
        spec["trace"] = False                      # traces are recorded for one Simulation
    out, meta = MG.run_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert sum(gold.meta["total_events"]) > 0
    check_oracle_against_station_golden(gold)


@pytest.mark.parametrize("k", range(30))
def test_oracle_equals_live_reference_on_random_ring_specs(k):
    spec = _ring_spec(k)
    out, meta = MG.run_ring_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 100
    check_oracle_against_ring_golden(gold)


@pytest.mark.parametrize("k", range(20))
def test_oracle_equals_live_reference_on_random_load_balancer_specs(k):
    spec = _lb_spec(k)
    out, meta = MG.run_lb_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 50
    check_oracle_against_lb_golden(gold)


@pytest.mark.parametrize("k", range(40))
def test_oracle_equals_live_reference_on_tie_storms(k):
    """Same-nanosecond orders (lock-step constant sources, Requests injected at the start instant, c up to 16): the oracle's
    sort-index ledger against the reference's, full traces."""
    out, meta = MG.run_case(tie_spec(k))
    check_oracle_against_station_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(60)) + [1374])          # (1374 ...: the cases tools/gpu_random_sweep.py found, tests/test_gpu_random.py)
def test_oracle_equals_live_reference_with_several_sources_per_server(k):
    """Up to four Sources feeding one Server, in two `sources=[...]` orders, on tie storms and random configurations."""
    spec = multi_source_spec(k)
    if spec["mode"] == "replicas":
        spec["trace"] = False
    out, meta = MG.run_case(spec)
    check_oracle_against_station_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(30)) + [1068, 1084, 1282])
def test_oracle_equals_live_reference_with_several_sources_per_server_on_rings(k):
    out, meta = MG.run_ring_case(multi_source_ring_spec(k))
    check_oracle_against_ring_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", range(20))
def test_oracle_equals_live_reference_with_probes_on_load_balancer_graphs(k):
    out, meta = MG.run_lb_case(lb_probe_spec(k))
    check_oracle_against_lb_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(12)) + [1351])
def test_oracle_equals_live_reference_with_profiles_on_load_balancer_sources(k):
    out, meta = MG.run_lb_case(lb_profile_spec(k))
    check_oracle_against_lb_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", range(60))
def test_oracle_equals_live_reference_on_random_tandem_queues(k):
    """Server(downstream=<Server>), up to four in a row (tests/tandem_specs.py; every fourth case a lock-step tie storm): the
    cases tests/test_gpu_tandem.py runs engine == oracle on MI355X."""
    import tandem_specs as TS

    out, meta = MG.run_tandem_case(TS.tandem_spec(k))
    check_oracle_against_tandem_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(40)) + [7932, 8264])
def test_oracle_equals_live_reference_on_tandem_queues_with_probes_and_injected_requests(k):
    """tests/tandem_specs.py tandem_probe_case: Probes on Servers of the chains and Simulation.schedule() Requests next to tandem
    queues -- statistics, Sink records, every Probe sample and the full trace with sort indices.  7932 / 8264: two Requests reach
    an idle one-worker Server in one nanosecond and the reference rejects the second at the worker (server.py:223-234)."""
    import tandem_specs as TS

    out, meta = MG.run_tandem_case(TS.tandem_probe_case(k))
    check_oracle_against_tandem_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(0, 60)) + [1788, 4095, 4329, 4857, 6237])
def test_oracle_equals_live_reference_on_forests_of_servers(k):
    """tests/tandem_specs.py fan_in_case: several Servers forwarding to one, Sources of their own on downstream Servers; every third
    case lock-step constants.  The five extra cases are the ones the GPU sweep differed on while fan-in was being built (the worker's
    rejection of a second same-nanosecond Request; the Sources' first ticks and the restarted sort counter)."""
    import tandem_specs as TS
    from test_oracle_golden import check_oracle_against_fan_in_reference

    case = TS.fan_in_case(k)
    check_oracle_against_fan_in_reference(case, MG.run_fan_in_case(case))


# ---- the reference's own ParallelSimulation ---------------------------------------------------------------
def test_live_parallel_simulation_fixtures_are_current():
    """The committed parallel_* fixtures are what the live `ParallelSimulation(...).run()` computes now (threads and all)."""
    for spec in MG.PARALLEL_CASES:
        fn = MG.run_parallel_linked_case if spec["kind"] == "linked" else MG.run_parallel_independent_case
        out, meta = fn(dict(spec))
        gold = H.Golden(spec["name"])
        fresh = H.Golden.from_results(out, meta)
        drop = lambda m: {k: v for k, v in m.items() if k != "spec"}                                    # noqa: E731
        assert drop(fresh.meta) == drop(gold.meta), spec["name"]
        for k, v in gold.arrays.items():
            np.testing.assert_array_equal(fresh.arrays[k], v, err_msg=f"{spec['name']}: {k}")


def _two_partition_chain(hop):
    """Source -> Server_a -> <hop> -> Server_b -> Sink with library components; `hop(srv_b)` builds the hop."""
    from happysimulator import ConstantLatency, Server, Sink, Source

    sink = Sink("sink_b")
    srv_b = Server("srv_b", service_time=ConstantLatency(0.02), downstream=sink)
    h = hop(srv_b)
    srv_a = Server("srv_a", service_time=ConstantLatency(0.03), downstream=h if h is not None else srv_b)
    return Source.constant(rate=7, target=srv_a, event_type="Request"), srv_a, h, srv_b, sink


def test_what_the_live_linked_parallel_simulation_refuses():
    """Failing by design -- three facts about the reference's LINKED mode that shape what `hs.ParallelSimulation(links=...)` mirrors
    (DESIGN section 7): (1) a library Server inside a linked partition needs its private queue / driver / worker listed as entities
    (parallel/routing.py:52-60); (2) a library NetworkLink cannot cross partitions, whichever side owns it: it forwards at its own
    `now`, the coordinator demands `delay >= min_latency` (parallel/coordinator.py:213-219); (3) `PartitionLink(latency=<library
    distribution>)` is unusable: the coordinator calls `.sample()` (`coordinator.py:209`), which no LatencyDistribution has.  What
    does cross is an entity that returns `Event(time=self.now + delay)` -- the pattern of the reference's own tests, which the
    parallel_linked_* fixtures use."""
    import warnings

    from happysimulator import ConstantLatency
    from happysimulator.components.network.link import NetworkLink
    from happysimulator.parallel import ParallelSimulation, PartitionLink, SimulationPartition

    warnings.simplefilter("ignore")
    link = lambda srv_b: NetworkLink("hop", latency=ConstantLatency(0.05), egress=srv_b)           # noqa: E731
    src, a, hop, b, sink = _two_partition_chain(link)
    ps = ParallelSimulation([SimulationPartition(name="A", entities=[a, hop], sources=[src]),
                             SimulationPartition(name="B", entities=[b, sink])], duration=2.0,
                            links=[PartitionLink("A", "B", min_latency=0.05)])
    with pytest.raises(RuntimeError, match="srv_a.driver.*not in this partition"):
        ps.run()
    for owner in ("A", "B"):
        src, a, hop, b, sink = _two_partition_chain(link)
        pa = SimulationPartition(name="A", entities=MG._server_parts(a) + ([hop] if owner == "A" else []), sources=[src])
        pb = SimulationPartition(name="B", entities=([hop] if owner == "B" else []) + MG._server_parts(b) + [sink])
        with pytest.raises(RuntimeError, match="violates min_latency: delay=0.000000s"):
            ParallelSimulation([pa, pb], duration=2.0, links=[PartitionLink("A", "B", min_latency=0.05)]).run()
    src, a, hop, b, sink = _two_partition_chain(lambda srv_b: None)
    with pytest.raises(AttributeError, match="no attribute 'sample'"):
        ParallelSimulation([SimulationPartition(name="A", entities=MG._server_parts(a), sources=[src]),
                            SimulationPartition(name="B", entities=MG._server_parts(b) + [sink])], duration=2.0,
                           links=[PartitionLink("A", "B", min_latency=0.05, latency=ConstantLatency(0.05))]).run()


@pytest.mark.parametrize("k", range(24))
def test_oracle_equals_live_reference_on_random_round_robin_and_random_load_balancers(k):
    """The LoadBalancer's default RoundRobin strategy and Random (its `random.choice` plugged per Request) on random topologies: the
    live reference against the oracle -- totals, per-backend statistics, every Sink record, probes, the full trace."""
    out, meta = MG.run_lb_case(lb_strategy_spec(k))
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 50
    check_oracle_against_lb_golden(gold)


@pytest.mark.parametrize("k", range(6))
def test_live_reference_ignores_a_probe_start_time(k):
    """`Probe(start_time=...)` seeds the probe's ConstantArrivalTimeProvider, but `Source.start()` overwrites the provider's
    clock with the Simulation's start time before it draws the first tick (load/source.py:120-127): the live reference samples
    at start + k * interval whatever `start_time` says.  The mirror accepts the argument and does the same."""
    spec = next(s for s in (_station_spec(j) for j in range(7 * k, 400)) if s.get("probes") and s["mode"] != "replicas")
    spec["trace"] = False
    plain, meta = MG.run_case(dict(spec))
    shifted, meta2 = MG.run_case(dict(spec, probe_start_s=0.37 * spec["end_s"]))
    assert len(plain["probe_t_ns"]) > 0
    for key in plain:
        if key != "meta":                         # (the spec itself, with the extra field)
            assert np.array_equal(plain[key], shifted[key]), key
    check_oracle_against_station_golden(H.Golden.from_results(shifted, meta2))


@pytest.mark.parametrize("k", range(6))
def test_live_reference_capacity_probes_are_functions_of_active_requests(k):
    """`Server.available_capacity` (= limit - active, concurrency.py:129-131) and the callable `Server.has_capacity` (the probe
    calls it: probe.py:52-55; active < limit, concurrency.py:117-127) sampled by the live reference == the mirror's value maps
    applied to the live `active_requests` samples of the same run (probes draw no random numbers: same trajectory)."""
    import happy_simulator_amd as hs

    spec = next(s for s in (_station_spec(j) for j in range(5 * k, 400)) if s["mode"] != "replicas" and s["downstream"])
    spec["trace"] = False
    n = spec["n_chains"]
    conc = spec["concurrency"] if isinstance(spec["concurrency"], (list, tuple)) else [spec["concurrency"]] * n
    spec["probes"] = [["active_requests", 0.05 * spec["end_s"]]] * n
    base, _ = MG.run_case(dict(spec))
    for metric in ("available_capacity", "has_capacity"):
        out, _ = MG.run_case(dict(spec, probes=[[metric, 0.05 * spec["end_s"]]] * n))
        assert np.array_equal(out["probe_t_ns"], base["probe_t_ns"]) and np.array_equal(out["probe_off"], base["probe_off"])
        for c in range(n):
            lo, hi = int(base["probe_off"][c]), int(base["probe_off"][c + 1])
            srv = hs.Server(f"s{c}", concurrency=int(conc[c]))
            f = hs.Probe.value_map(metric, srv)
            assert [int(f(int(a))) for a in base["probe_v"][lo:hi]] == [int(v) for v in out["probe_v"][lo:hi]], (metric, c)
            assert hi > lo


@pytest.mark.parametrize("k", range(8))
def test_oracle_equals_live_reference_on_load_balancers_with_up_to_32_workers_per_backend(k):
    out, meta = MG.run_lb_case(lb_workers_spec(k))
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 500
    check_oracle_against_lb_golden(gold)


@pytest.mark.parametrize("k", range(24))
def test_oracle_equals_live_reference_on_rings_with_constant_exponential_and_no_jitter(k):
    """NetworkLink(jitter=ConstantLatency(x)) -- the reference's datacenter_network preset (components/network/conditions.py:60-63) --
    next to exponential jitter and none, per link: a constant on top of the base latency, no random number (link.py:195-200)."""
    out, meta = MG.run_ring_case(jitter_ring_spec(k))
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 100
    check_oracle_against_ring_golden(gold)


def test_network_condition_presets_equal_the_live_references():
    """components/network/conditions.py:13-258 -- nine NetworkLink presets -- against the mirror's table: every parameter."""
    import happy_simulator_amd as hs
    from happysimulator.components.network import conditions as RC

    def params(link):
        jit = link.jitter
        jm = None if jit is None else getattr(jit, "mean", None)
        if jm is None and jit is not None:
            jm = jit._mean_latency
        lat = getattr(link.latency, "mean", None)
        if lat is None:
            lat = link.latency._mean_latency
        return (link.name, float(lat), link.bandwidth_bps, link.packet_loss_rate, None if jit is None else type(jit).__name__, None if jm is None else float(jm))

    for fn in ("local_network", "datacenter_network", "cross_region_network", "internet_network", "satellite_network",
               "mobile_3g_network", "mobile_4g_network"):
        assert params(getattr(hs, fn)()) == params(getattr(RC, fn)()), fn
        assert params(getattr(hs, fn)("x")) == params(getattr(RC, fn)("x")), fn
    assert params(hs.lossy_network(0.07)) == params(RC.lossy_network(0.07))
    assert params(hs.lossy_network(0.5, "l2", 0.2)) == params(RC.lossy_network(0.5, "l2", 0.2))
    assert params(hs.slow_network(0.3)) == params(RC.slow_network(0.3))
    assert params(hs.slow_network(0.3, "s2", 5e5)) == params(RC.slow_network(0.3, "s2", 5e5))
    with pytest.raises(ValueError):
        hs.lossy_network(1.5)
    with pytest.raises(ValueError):
        RC.lossy_network(1.5)


@pytest.mark.parametrize("k", range(40))
def test_oracle_equals_live_reference_on_random_graphs_beyond_the_engines(k):
    """random_specs.graph_spec: routers with up to 8 targets incl. Servers and several upstreams, links with several senders, Servers
    behind Servers inside link networks, more than four Sources per Server -- live reference == oracle."""
    out, meta = MG.run_graph_case(graph_spec(k))
    check_oracle_against_graph_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", range(60))
def test_oracle_equals_live_reference_on_graphs_with_several_load_balancers(k):
    """random_specs.lb_graph_spec: one to three LoadBalancers (ConsistentHash / RoundRobin / Random) behind Sources, Servers and
    routers of a general graph, `schedule()`d Requests on the key-less ones -- LoadBalancer.stats, BackendInfo.total_requests,
    RoundRobin._index and everything test_..._on_random_graphs_beyond_the_engines compares: live reference == oracle."""
    out, meta = MG.run_graph_case(lb_graph_spec(k))
    check_oracle_against_graph_golden(H.Golden.from_results(out, meta))


def _same_results(a, b, what):
    assert a.keys() == b.keys(), what
    for k in a:
        if k == "meta":
            continue
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{what}: {k}")


@pytest.mark.parametrize("kind,k", [("station", 3), ("station", 11), ("tie", 5), ("ring", 2), ("ring", 9), ("lb", 1), ("lb", 7),
                                    ("multi_ring", 4)])
def test_the_reference_in_windows_equals_the_reference_in_one_run(kind, k):
    """VERDICT r4 missing 4: what `hs_engine_run_until` / `hs_lb_run` rely on when they serve a later window end by repeating the
    run (include/hs_engine.h).  The LIVE reference driven window by window (core/simulation.py:527-541 `_run_window`: growing ends,
    an end inside the gap before the event beyond the previous end, a repeated and an EARLIER end) leaves exactly what ONE run to
    the last end leaves: every count, statistic, Sink record, the total and the final time."""
    spec, run = {"station": (_station_spec(k), MG.run_case), "tie": (tie_spec(k), MG.run_case), "ring": (_ring_spec(k), MG.run_ring_case),
                 "lb": (_lb_spec(k), MG.run_lb_case), "multi_ring": (multi_source_ring_spec(k), MG.run_ring_case)}[kind]
    spec = dict(spec, trace=False)
    if spec.get("mode") == "replicas":
        spec["mode"] = "single"
    one, meta1 = run(dict(spec))
    e = spec["end_s"]
    win, meta2 = run(dict(spec, windows=[e / 7, e / 7 + 1e-9, e / 3, e / 3, e / 5, 2 * e / 3]))
    assert meta1["total_events"] == meta2["total_events"] and meta1["final_ns"] == meta2["final_ns"]
    assert sum(meta1["total_events"]) > 20
    _same_results(one, win, f"{kind} {k}")
