"""Host side of the general-graph path (happy_simulator_amd/graph_engine.py), no GPU: which graphs go there, the node arrays and
stream numbering lower_general() builds, the refusals it keeps, the write-back, and the ABI structs of include/hs_engine.h."""
import ctypes as C

import numpy as np
import pytest

import graph_specs as GS
import happy_simulator_amd as hs
import helpers as H
from happy_simulator_amd import _native as N
from happy_simulator_amd.graph_engine import GeneralGraph, lower_general, write_back_general
from happy_simulator_amd.lowering import LoweredGraph
from random_specs import graph_spec


def test_graphs_the_station_engines_refuse_lower_to_the_single_heap_path():
    for name in H.golden_names("graph"):
        sim, _ = GS.build(H.Golden(name).spec)
        assert isinstance(sim.lowered(), GeneralGraph)
        assert sim._station_refusal                            # (what lowering.lower() said: kept for the messages)


def test_station_shaped_graphs_stay_on_the_station_engines():
    sink = hs.Sink("k")
    servers = [hs.Server(f"s{i}", service_time=hs.ExponentialLatency(0.05)) for i in range(3)]
    for i, sv in enumerate(servers):
        link = hs.NetworkLink(f"l{i}", latency=hs.ConstantLatency(0.001), jitter=hs.ExponentialLatency(0.002), egress=servers[(i + 1) % 3])
        sv.downstream = hs.RandomRouter(f"r{i}", targets=[sink, link])
    sources = [hs.Source.poisson(rate=5.0, target=sv, name=f"src{i}") for i, sv in enumerate(servers)]
    sim = hs.Simulation(duration=1.0, sources=sources, entities=servers + [sink])
    assert isinstance(sim.lowered(), LoweredGraph)


def test_node_arrays_and_stream_numbering_of_a_golden_graph():
    spec = H.Golden("graph_shared_links_tandem").spec
    sim, ents = GS.build(spec)
    g = sim.lowered()
    a = g.arrays
    S, V, R, L, K = (len(spec[k]) if k != "n_sinks" else spec[k] for k in ("sources", "servers", "routers", "links", "n_sinks"))
    assert a.n == S + V + R + L + K
    np.testing.assert_array_equal(a.kind, [N.NODE_SOURCE] * S + [N.NODE_SERVER] * V + [N.NODE_ROUTER] * R + [N.NODE_LINK] * L + [N.NODE_SINK] * K)
    # Source k / Server s / router r / link l draw from stream base k / s / r / l (the fixtures' numbering)
    np.testing.assert_array_equal(a.stream_base[:S + V + R + L], list(range(S)) + list(range(V)) + list(range(R)) + list(range(L)))
    node = lambda kind, i: {"server": S, "router": S + V, "link": S + V + R, "sink": S + V + R + L}[kind] + i   # noqa: E731
    np.testing.assert_array_equal(a.target[:S], [node("server", sc["to"]) for sc in spec["sources"]])
    np.testing.assert_array_equal(a.target[S:S + V], [node(*sv["out"]) for sv in spec["servers"]])
    np.testing.assert_array_equal(a.target[S + V + R:S + V + R + L], [node("server", lk["to"]) for lk in spec["links"]])
    flat = [node(k, i) for rt in spec["routers"] for k, i in rt["targets"]]
    np.testing.assert_array_equal(a.rt_targets, flat)
    np.testing.assert_array_equal(a.rt_cnt[S + V:S + V + R], [len(rt["targets"]) for rt in spec["routers"]])
    np.testing.assert_array_equal(a.link_loss_rate[S + V + R:S + V + R + L], [lk["loss"] for lk in spec["links"]])
    np.testing.assert_array_equal(a.queue_cap[S:S + V], [-1 if sv["cap"] is None else sv["cap"] for sv in spec["servers"]])
    np.testing.assert_array_equal(a.src_kind[:S], [N.SRC_POISSON if sc["kind"] == "poisson" else N.SRC_CONSTANT for sc in spec["sources"]])
    st = a.struct()
    assert st.n_nodes == a.n and st.n_rt == len(flat)


def test_entities_only_reachable_downstream_become_nodes_in_discovery_order():
    sink = hs.Sink("k")
    b = hs.Server("b", service_time=hs.ConstantLatency(0.01), downstream=sink)
    link = hs.NetworkLink("l", latency=hs.ConstantLatency(0.001), egress=b)
    a = hs.Server("a", service_time=hs.ConstantLatency(0.01), downstream=link)
    c = hs.Server("c", service_time=hs.ConstantLatency(0.01), downstream=link)        # the link's second sender
    src = [hs.Source.constant(rate=3.0, target=a, name="s0"), hs.Source.constant(rate=2.0, target=c, name="s1")]
    g = lower_general(src, [a, c, b])              # the link and the Sink are not listed
    assert [getattr(x, "name") for x in g.nodes] == ["s0", "s1", "a", "c", "b", "l", "k"]
    np.testing.assert_array_equal(g.arrays.stream_base, [0, 1, 0, 1, 2, 0, 0])
    with pytest.raises(hs.UnsupportedTopology, match="not listed in `entities`"):     # (a Server needs the clock `entities` hands out)
        lower_general(src, [a, c])


@pytest.mark.parametrize("k", range(40))
def test_random_graph_specs_lower(k):
    sim, ents = GS.build(graph_spec(k))
    g = sim.lowered()
    if isinstance(g, GeneralGraph):
        a = g.arrays
        assert (a.target[a.kind == N.NODE_SOURCE] >= 0).all()
        assert all(0 <= t < a.n and a.kind[t] != N.NODE_SOURCE for t in a.rt_targets)


def test_probes_and_profiles_on_a_general_graph_become_nodes_behind_the_sources():
    sink = hs.Sink("k")
    sv = hs.Server("s", concurrency=64, service_time=hs.ExponentialLatency(0.05), downstream=sink)      # c > 32: not a station
    src = hs.Source.poisson(rate=5.0, target=sv, name="src")
    ramp = hs.Source.with_profile(hs.LinearRampProfile(duration_s=2.0, start_rate=1.0, end_rate=9.0), target=sv, name="ramp")
    spike = hs.Source.with_profile(hs.SpikeProfile(baseline_rate=2.0, spike_rate=20.0, warmup_s=0.5, spike_duration_s=0.25), target=sv,
                                   poisson=False, name="spike")
    p_depth, _ = hs.Probe.on(sv, "depth", interval=0.5)
    p_util, _ = hs.Probe.on(sv, "utilization", interval=0.25)
    p_recv, _ = hs.Probe.on(sink, "events_received", interval=0.5)
    p_gen, _ = hs.Probe.on(ramp, "generated_count", interval=1.0)
    g = hs.Simulation(duration=1.0, sources=[src, ramp, spike], entities=[sv, sink], probes=[p_depth, p_util, p_recv, p_gen]).lowered()
    assert isinstance(g, GeneralGraph)
    a = g.arrays
    np.testing.assert_array_equal(a.kind, [N.NODE_SOURCE] * 3 + [N.NODE_PROBE] * 4 + [N.NODE_SERVER, N.NODE_SINK])
    np.testing.assert_array_equal(a.target[3:7], [7, 7, 8, 1])
    np.testing.assert_array_equal(a.probe_metric[3:7], [N.PROBE_METRICS["depth"], N.PROBE_METRICS["active_requests"],
                                                        N.PROBE_METRICS["events_received"], N.PROBE_METRICS["generated_count"]])
    np.testing.assert_array_equal(a.probe_interval_s[3:7], [0.5, 0.25, 0.5, 1.0])
    np.testing.assert_array_equal(a.src_profile_kind[:3], [N.PROF_CONSTANT, N.PROF_LINEAR_RAMP, N.PROF_SPIKE])
    np.testing.assert_array_equal(a.src_profile_params[1], [2.0, 1.0, 9.0, 0.0])
    np.testing.assert_array_equal(a.src_profile_params[2], [2.0, 20.0, 0.5, 0.25])
    np.testing.assert_array_equal(a.src_rate[:3], [5.0, 9.0, 20.0])                      # (a profile's peak)
    np.testing.assert_array_equal(a.src_kind[:3], [N.SRC_POISSON, N.SRC_POISSON, N.SRC_CONSTANT])


def test_what_the_single_heap_path_refuses_too():
    sink = hs.Sink("k")
    sv = hs.Server("s", concurrency=64, service_time=hs.ExponentialLatency(0.05), downstream=sink)      # c > 32: not a station
    src = hs.Source.poisson(rate=5.0, target=sv, name="src")
    other = hs.Server("elsewhere", service_time=hs.ExponentialLatency(0.05))
    probe, _ = hs.Probe.on(other, "depth", interval=0.5)
    with pytest.raises(hs.UnsupportedTopology, match="not an entity of this Simulation"):
        hs.Simulation(duration=1.0, sources=[src], entities=[sv, sink], probes=[probe]).lowered()
    probe, _ = hs.Probe.on(sink, "events_received", interval=0.5)
    with pytest.raises(hs.UnsupportedTopology, match="never terminates either"):         # (a Probe's ticks are primary events)
        sim = hs.Simulation(sources=[], entities=[sv, sink], probes=[probe])
        sim.schedule(hs.Event(time=hs.Instant.from_seconds(0.5), event_type="Request", target=sv))
        sim.run()
    assert isinstance(hs.Simulation(duration=1.0, sources=[src], entities=[sv, sink]).lowered(), GeneralGraph)
    with pytest.raises(hs.UnsupportedTopology, match="would need"):       # 64 such chains (64 heaps) at 10^7 / s for a minute: refused before anything runs
        sinks = [hs.Sink(f"k{i}") for i in range(64)]
        svs = [hs.Server(f"s{i}", concurrency=64, service_time=hs.ExponentialLatency(0.05), downstream=sinks[i]) for i in range(64)]
        srcs = [hs.Source.poisson(rate=1e7, target=x, name=f"src{i}") for i, x in enumerate(svs)]
        hs.Simulation(duration=60.0, sources=srcs, entities=svs + sinks).run()


def test_write_back_puts_the_node_results_on_the_objects():
    spec = H.Golden("graph_fanout_8").spec
    sim, ents = GS.build(spec)
    g = sim.lowered()
    a = g.arrays
    stats = {k: (np.arange(a.n, dtype=np.float64) * 0.5 if k == "total_service_s" else
                 np.arange(len(a.rt_targets), dtype=np.int64) + 1 if k == "rt_taken" else np.arange(a.n, dtype=np.int64) * 10 + j)
             for j, k in enumerate(N.GRAPH_STATS)}
    sink_nodes = [g.node_of[id(x)] for x in ents["sinks"]]
    rec_node = np.array([sink_nodes[1], sink_nodes[0], sink_nodes[1], sink_nodes[1]], np.int32)
    rec_t, rec_cr = np.array([5, 6, 7, 9], np.int64), np.array([1, 2, 3, 4], np.int64)
    write_back_general(g, stats, rec_node, rec_t, rec_cr)
    i = g.node_of[id(ents["servers"][2])]
    sv = ents["servers"][2]
    assert (sv.stats_accepted, sv.stats_dropped, sv._requests_completed, sv._requests_rejected) == (
        i * 10 + 2, i * 10 + 3, i * 10 + 4, i * 10 + 5)
    assert sv._total_service_time == i * 0.5 and sv.depth == i * 10 + 7 and sv.active_requests == i * 10 + 8
    s0 = ents["sources"][0]
    assert s0.generated_count == 0 and s0._event_provider._generated == 1
    np.testing.assert_array_equal(ents["sinks"][1].completion_ns, [5, 7, 9])
    np.testing.assert_array_equal(ents["sinks"][0]._created_ns, [2])
    rt = ents["routers"][0]
    assert rt.stats_routed == g.node_of[id(rt)] * 10 + 13
    # eight targets, two of them listed twice (link0, sink1 ...): counts add up per target NAME (random_router.py:37)
    assert sum(rt.target_counts.values()) == sum(range(1, 9))
    assert rt.target_counts["link0"] == 2 + 6


def test_abi_structs_of_the_graph_entry_points():
    assert C.sizeof(N.GraphConfig) == 64
    assert C.sizeof(N.GraphNodes) == 8 + 15 * 8 + 8 + 9 * 8
    assert C.sizeof(N.GraphStats) == 16 * 8
    L = N.lib()
    for sym in ("hs_graph_create", "hs_graph_schedule", "hs_graph_run_until", "hs_graph_run_many", "hs_graph_run_parts", "hs_graph_get_summary", "hs_graph_get_stats",
                "hs_graph_read_records", "hs_graph_last_error", "hs_graph_destroy"):
        assert sym in N.EXPORTED_SYMBOLS and getattr(L, sym)


@pytest.mark.parametrize("k", range(12))
def test_split_parts_partitions_the_graph_without_cutting_an_edge(k):
    """graph_engine.split_parts (hs_graph_run_parts' host side): unions of random graphs fall into parts that (i) partition the
    nodes, ascending inside a part (Sources first: hs_graph_nodes' contract), (ii) keep every edge -- target, router targets,
    LoadBalancer backends -- inside one part under the part's own numbering, (iii) carry every per-node array and the backends'
    names unchanged, (iv) never exceed `max_parts`; a connected graph is not split."""
    from happy_simulator_amd.graph_engine import split_parts
    from random_specs import lb_graph_spec, union_spec

    rng = np.random.default_rng(95_000 + k)
    members = [(lb_graph_spec if rng.random() < 0.5 else graph_spec)(int(rng.integers(0, 5000))) for _ in range(int(rng.integers(2, 7)))]
    sim, _ = GS.build(union_spec(members))
    g = sim.lowered()
    assert isinstance(g, GeneralGraph)
    a = g.arrays
    for max_parts in (2048, 3):
        parts = split_parts(a, max_parts)
        assert parts is not None and 2 <= len(parts) <= max_parts and len(parts) >= min(len(members), max_parts)
        seen = np.concatenate([ids for ids, _pos, _b in parts])
        np.testing.assert_array_equal(np.sort(seen), np.arange(a.n))                  # (i) a partition
        rt_seen = np.concatenate([pos for _ids, pos, _b in parts])
        np.testing.assert_array_equal(np.sort(rt_seen), np.arange(len(a.rt_targets)))
        for ids, pos, b in parts:
            assert (np.diff(ids) > 0).all() and b.n == len(ids)
            kinds = a.kind[ids]
            first_other = int(np.argmax(kinds != N.NODE_SOURCE)) if (kinds != N.NODE_SOURCE).any() else len(kinds)
            assert (kinds[first_other:] != N.NODE_SOURCE).all()                       # Sources first
            for nm in ("kind", "stream_base", "src_kind", "src_rate", "concurrency", "lat_kind", "lat_mean_s", "link_loss_rate", "queue_cap"):
                np.testing.assert_array_equal(getattr(b, nm), getattr(a, nm)[ids], err_msg=nm)
            has = a.target[ids] >= 0                                                  # (ii) edges stay inside, renumbered
            np.testing.assert_array_equal(ids[b.target[has]], a.target[ids][has])
            assert (b.target[~has] == -1).all()
            np.testing.assert_array_equal(b.rt_cnt, a.rt_cnt[ids])
            np.testing.assert_array_equal(ids[b.rt_targets], a.rt_targets[pos])
            off = 0
            for j, i in enumerate(ids):
                assert b.rt_off[j] == off or b.rt_cnt[j] == 0
                np.testing.assert_array_equal(pos[off:off + b.rt_cnt[j]], np.arange(a.rt_off[i], a.rt_off[i] + a.rt_cnt[i]))
                off += int(b.rt_cnt[j])
            if a.names is not None:                                                   # (iii) names travel with their nodes
                for j, i in enumerate(ids):
                    assert b.names[b.name_off[j]:b.name_off[j + 1]] == a.names[a.name_off[i]:a.name_off[i + 1]]
            b.struct()
    chain = hs.Simulation(duration=1.0, sources=[hs.Source.poisson(rate=5.0, target=(sv := hs.Server("s", concurrency=64, downstream=hs.Sink("k"))))],
                          entities=[sv, sv.downstream])
    assert split_parts(chain.lowered().arrays) is None                                # one component: the one heap
