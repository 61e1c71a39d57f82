"""GPU: tandem queues -- `Server(..., downstream=<another Server>)` (components/server/server.py:64-122,271-272; the forwarded Event
keeps its context, core/entity.py:83-105) -- engine == oracle on everything the ABI reports: totals, the per-kind histogram, the
time of the one event beyond end_time, every Server's statistics (binary64 total_service_time included), every Sink record.
The engine runs a chain in passes, upstream Servers first (csrc/hs_station.hpp "tandem queues"); the oracle is the reference's
single heap (oracle/hs_oracle.c on_continuation: forward(event, downstream)).  The oracle's tandem path itself is pinned by the
live-reference golden `tandem_*` (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import tandem_specs as TS
from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu


def _run_case(spec, windows=(), flags=0):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    g, srcs, servers, sinks = TS.oracle_graph(spec)
    end = int(spec["end_s"] * 1e9)
    r = O.run(g, end, seed=spec["seed"], windows=[int(w * 1e9) for w in windows])
    with StationEngine(TS.engine_arrays(spec), mode=N.MODE_SINGLE, horizon_ns=end, seed=spec["seed"]) as eng:
        if flags:
            eng.set_debug_flags(flags)
        for w in windows:
            eng.run_until(int(w * 1e9))
        eng.run_until(end)
        TS.compare(spec, eng, r, srcs, servers, sinks)
        r.engine_path = eng.tandem_path()
    return r


SINGLE_HEAP = 1 << 17      # debug flag: the whole run on the single-heap loop (csrc/hs_exact.hpp), as after an undecided tie


def test_two_servers_in_a_row():
    spec = dict(chains=[dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="exp", mean=0.1, conc=1, qcap=None), dict(svc="exp", mean=0.08, conc=1, qcap=None)])],
                end_s=20.0, seed=42)
    r = _run_case(spec)
    assert r.events_processed > 1500


def test_lock_step_constants_share_every_nanosecond():
    """Arrivals every 100 ms, both Servers take exactly 100 ms: from the second arrival on, the tick, the first Server's
    completion, its forward and the second Server's completion all happen on the same nanosecond."""
    for means in ((0.1, 0.1), (0.1, 0.1, 0.1, 0.1), (0.05, 0.1), (0.0, 0.1), (0.1, 0.0, 0.05), (0.2, 0.1)):
        spec = dict(chains=[dict(arr="constant", rate=10.0, stop_after_s=None, sink=True,
                                 stages=[dict(svc="const", mean=m, conc=1, qcap=None) for m in means])],
                    end_s=3.0, seed=7)
        _run_case(spec)
        spec["chains"][0]["stages"][-1]["conc"] = 2
        spec["chains"].append(dict(arr="constant", rate=10.0, stop_after_s=1.0, sink=False,
                                   stages=[dict(svc="const", mean=m, conc=2, qcap=1) for m in means]))
        _run_case(spec)


@pytest.mark.parametrize("first", range(0, 200, 50))
def test_random_tandems_match_the_oracle(first):
    bad, paths = [], {0: 0, 1: 0, 2: 0}
    for k in range(first, first + 50):
        try:
            paths[_run_case(TS.tandem_spec(k)).engine_path] += 1
            if k % 5 == 0:
                assert _run_case(TS.tandem_spec(k), flags=SINGLE_HEAP).engine_path == (2 if TS.has_tandem(TS.tandem_spec(k)) else 0)
        except AssertionError as e:
            bad.append((k, str(e).strip().splitlines()[:3]))
    assert not bad, bad
    print("paths (none / passes / single heap):", paths)
    assert paths[1] > paths[2]            # the single heap is the exception: lock-step ties


def test_several_windows_continue_the_chain():
    """run_until twice: the event the first window elected beyond its end (a completion whose forward is still pending, a
    forward's enqueue, ...) is part of the state the second window starts from."""
    for k in (1, 4, 8, 12, 17, 20, 33, 1196, 1756, 3892):
        spec = TS.tandem_spec(k)
        _run_case(spec, windows=(0.37 * spec["end_s"], 0.81 * spec["end_s"]))
        _run_case(spec, windows=(0.37 * spec["end_s"], 0.81 * spec["end_s"]), flags=SINGLE_HEAP)


def test_ties_the_lineage_key_does_not_decide_go_to_the_single_heap():
    """The three cases of 3 000 (profiles/r03_gpu_sweep_tandem.log, first sweep) on which the passes alone differed from the
    reference: lock-step constant chains whose Servers' events tie on (time, creation time, steps from the root, the root's creation
    time) -- the reference decides by the roots' ancestry, arbitrarily far back (tools/election_rules.py --family tandem)."""
    for k in (1196, 1756, 3892):
        r = _run_case(TS.tandem_spec(k))
        assert r.engine_path == 2


def test_what_is_not_lowered_is_refused_by_name():
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    spec = dict(chains=[dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="exp", mean=0.1, conc=1, qcap=None)] * 2)] * 2, end_s=1.0, seed=1)
    st = TS.engine_arrays(spec)
    st.egress[1], st.downstream_lp[1] = N.EGRESS_SERVER, 0          # station 0 -> 1 -> 0
    with pytest.raises(N.EngineError, match="cycle of Servers"):
        StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=10**9)
    st = TS.engine_arrays(spec)
    with pytest.raises(N.EngineError, match="HS_MODE_SINGLE"):
        StationEngine(st, mode=N.MODE_REPLICAS, horizon_ns=10**9)


# ---- through the reference-shaped API, against the live-reference goldens (tests/golden/tandem_*.npz) -----------------------------
def _build_api(spec):
    import happy_simulator_amd as hs

    sources, entities, servers, sinks = [], [], {}, []
    for c, ch in enumerate(spec["chains"]):
        sink = hs.Sink(f"sink{c}") if ch["sink"] else None
        nxt = sink
        for st in reversed(range(len(ch["stages"]))):
            sg = ch["stages"][st]
            lat = hs.ExponentialLatency(sg["mean"]) if sg["svc"] == "exp" else hs.ConstantLatency(sg["mean"])
            nxt = servers[(c, st)] = hs.Server(f"srv{c}_{st}", concurrency=sg["conc"], service_time=lat, queue_capacity=sg["qcap"],
                                               downstream=nxt)
        make = hs.Source.poisson if ch["arr"] == "poisson" else hs.Source.constant
        sources.append(make(rate=ch["rate"], target=servers[(c, 0)], name=f"src{c}", stop_after=ch["stop_after_s"]))
        entities += [servers[(c, st)] for st in range(len(ch["stages"]))]
        if sink is not None:
            entities.append(sink)
        sinks.append(sink)
    return sources, entities, servers, sinks


@pytest.mark.parametrize("name", ["tandem_2stage_philox", "tandem_lock_step_consts", "tandem_4stage_mixed"])
def test_tandem_queues_through_the_api_match_the_reference_golden(name):
    import happy_simulator_amd as hs
    import helpers as H

    gold = H.Golden(name)
    spec = gold.spec
    sources, entities, servers, sinks = _build_api(spec)
    summary = hs.Simulation(duration=spec["end_s"], sources=sources, entities=entities, seed=spec["seed"]).run()
    assert summary.total_events_processed == gold.meta["total_events"][0]
    assert summary.duration_s == gold.meta["duration_s"][0]
    order, first = TS.station_index(spec)
    for i, (c, st) in enumerate(order):
        sv = servers[(c, st)]
        assert sv.stats_accepted == gold.accepted[i] and sv.stats_dropped == gold.dropped[i], (c, st)
        assert sv.stats.requests_completed == gold.completed[i] and sv.depth == gold.depth[i], (c, st)
        assert sv.stats.total_service_time == gold.total_service_s[i], (c, st)
        assert sv.active_requests == gold.active[i], (c, st)
    for c, src in enumerate(sources):
        assert src.generated_count == gold.generated[c]
        t, lat = gold.sink_records(c)
        if sinks[c] is None:
            continue
        assert sinks[c].events_received == len(t)
        np.testing.assert_array_equal(sinks[c].completion_ns, t)
        assert sinks[c].latencies_s == lat.tolist()


def test_api_refusals_name_what_is_missing():
    import happy_simulator_amd as hs

    sink = hs.Sink()
    b = hs.Server("b", service_time=hs.ExponentialLatency(0.05), downstream=sink)
    a1 = hs.Server("a1", service_time=hs.ExponentialLatency(0.05), downstream=b)
    a2 = hs.Server("a2", service_time=hs.ExponentialLatency(0.05), downstream=b)
    s1, s2 = hs.Source.poisson(rate=5, target=a1, name="s1"), hs.Source.poisson(rate=5, target=a2, name="s2")
    summary = hs.Simulation(duration=1.0, sources=[s1, s2], entities=[a1, a2, b, sink]).run()      # fan-in: lowered (single heap)
    assert b.stats_accepted == a1.stats.requests_completed + a2.stats.requests_completed > 0
    assert summary.total_events_processed > 50
    b = hs.Server("b", service_time=hs.ExponentialLatency(0.05), downstream=hs.Sink())
    a = hs.Server("a", service_time=hs.ExponentialLatency(0.05), downstream=b)
    with pytest.raises(hs.UnsupportedTopology, match="not listed in `entities`"):
        hs.Simulation(duration=1.0, sources=[hs.Source.poisson(rate=5, target=a)], entities=[a]).run()


def test_probes_and_scheduled_requests_next_to_tandem_queues():
    """Probes on Servers of a chain and Requests injected with Simulation.schedule() into its second Server: pre-run events whose
    sort indices run-time events can overtake (csrc/hs_exact.hpp).  Two of the Requests are injected at one instant: the passes
    report that and the run is repeated on the single heap."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    spec = dict(chains=[dict(arr="poisson", rate=12.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="exp", mean=0.05, conc=1, qcap=None), dict(svc="exp", mean=0.07, conc=2, qcap=3),
                                     dict(svc="const", mean=0.02, conc=1, qcap=None)]),
                        dict(arr="constant", rate=10.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="const", mean=0.1, conc=1, qcap=None), dict(svc="const", mean=0.1, conc=1, qcap=None)])],
                end_s=6.0, seed=11)
    g, srcs, servers, sinks = TS.oracle_graph(spec)
    probes = [((0, 1), "depth", 0.25), ((0, 0), "completed", 0.5), ((1, 1), "active", 0.1)]
    metric_id = {"depth": N.PROBE_METRICS["depth"], "completed": N.PROBE_METRICS["requests_completed"], "active": N.PROBE_METRICS["active_requests"]}
    pnodes = [g.probe(servers[cs], metric_id[m], iv) for cs, m, iv in probes]
    sched = [((0, 1), 1.0), ((0, 1), 1.0), ((1, 1), 2.5), ((0, 2), 0.0)]
    end = int(spec["end_s"] * 1e9)
    r = O.run(g, end, seed=spec["seed"], schedule=[(servers[cs], int(t * 1e9)) for cs, t in sched])
    st = TS.engine_arrays(spec)
    order, first = TS.station_index(spec)
    lp_of = {cs: i for i, cs in enumerate(order)}
    st.probe_metric = np.full(st.n, N.PROBE_NONE, np.uint8)
    st.probe_interval_s = np.ones(st.n)
    for cs, m, iv in probes:
        st.probe_metric[lp_of[cs]], st.probe_interval_s[lp_of[cs]] = metric_id[m], iv
    st.probe_order = np.array([lp_of[cs] for cs, _, _ in probes], np.int32)
    per = [[] for _ in range(st.n)]
    for rank, (cs, t) in enumerate(sched):
        per[lp_of[cs]].append((int(t * 1e9), rank))
    for lst in per:
        lst.sort(key=lambda x: x[0])
    st.sched_off = np.concatenate([[0], np.cumsum([len(x) for x in per])]).astype(np.int64)
    st.sched_time_ns = np.array([t for lst in per for t, _ in lst], np.int64)
    st.sched_rank = np.array([rk for lst in per for _, rk in lst], np.int64)
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end, seed=spec["seed"]) as eng:
        eng.run_until(end)
        assert eng.tandem_path() == 2
        TS.compare(spec, eng, r, srcs, servers, sinks)
        for (cs, _m, _iv), nd in zip(probes, pnodes):
            t, v = r.sinks[nd]
            pt, pv = eng.read_probe(lp_of[cs])
            np.testing.assert_array_equal(pt, t)
            np.testing.assert_array_equal(pv, v)


def test_random_tandem_queues_with_probes_and_injected_requests_equal_the_oracle():
    """300 tandem_probe_case configurations: on the passes without the prologue where no pre-run event shares its nanosecond with
    another event of its LP (hs_engine_prologue_path() == 1), on the single heap otherwise -- exact either way."""
    from collections import Counter

    paths = Counter()
    for k in range(300):
        spec = TS.tandem_probe_case(k)
        try:
            paths[TS.run_tandem_probe_case(spec)] += 1
        except AssertionError as e:
            raise AssertionError(f"tandem_probe_case({k}): {e}") from e
    print("(tandem path, prologue path):", dict(paths))
    assert paths[(1, 1)] > paths[(2, 2)] > 0


_fan_in_case = TS.fan_in_case


def _run_fan_in(case):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    sv, down = case["servers"], case["down"]
    n = len(sv)
    g = O.Graph()
    srcs = {i: g.source(O.ARR_POISSON if s["src"][0] == "poisson" else O.ARR_CONSTANT, s["src"][1], stream_base=i)
            for i, s in enumerate(sv) if s["src"] is not None}
    nodes = [g.server(O.LAT_EXP if s["svc"] == "exp" else O.LAT_CONST, s["mean"], concurrency=s["conc"],
                      queue_cap=-1 if s["qcap"] is None else s["qcap"], stream_base=i) for i, s in enumerate(sv)]
    sinks = {i: g.sink() for i, s in enumerate(sv) if s["sink"]}
    for i, nd in srcs.items():
        g.target[nd] = nodes[i]
    for i in range(n):
        g.target[nodes[i]] = nodes[down[i]] if down[i] >= 0 else sinks.get(i, -1)
    end = int(case["end_s"] * 1e9)
    r = O.run(g, end, seed=case["seed"])
    st = StationArrays(n=n, src_kind=np.array([N.SRC_NONE if s["src"] is None else N.SRC_POISSON if s["src"][0] == "poisson" else N.SRC_CONSTANT
                                                for s in sv], np.uint8),
                       src_rate=np.array([20.0 if s["src"] is None else s["src"][1] for s in sv]), src_stop_after_ns=np.full(n, -1, np.int64),
                       concurrency=np.array([s["conc"] for s in sv], np.int32),
                       svc_kind=np.array([N.LAT_EXPONENTIAL if s["svc"] == "exp" else N.LAT_CONSTANT for s in sv], np.uint8),
                       svc_mean_s=np.array([s["mean"] for s in sv]), queue_cap=np.array([-1 if s["qcap"] is None else s["qcap"] for s in sv], np.int64),
                       egress=np.array([N.EGRESS_SERVER if down[i] >= 0 else N.EGRESS_SINK if sv[i]["sink"] else N.EGRESS_NONE for i in range(n)], np.uint8))
    st.downstream_lp = np.array(down, np.int32)
    fan_in = len([d for d in down if d >= 0]) != len({d for d in down if d >= 0})
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end, seed=case["seed"]) as eng:
        eng.run_until(end)
        s = eng.summary()
        assert s.events_processed == r.events_processed
        np.testing.assert_array_equal(s.events_by_kind, r.events_by_kind)
        assert s.final_time_ns == r.final_time_ns
        stats = eng.lp_stats()
        for i in range(n):
            for k, ok in (("accepted", "accepted"), ("dropped", "dropped"), ("completed", "completed"), ("queue_depth", "depth"), ("active", "active")):
                assert stats[k][i] == getattr(r, ok)[nodes[i]], (k, i)
            assert stats["total_service_s"][i] == r.total_service_s[nodes[i]], i
            if i in srcs:
                assert stats["generated"][i] == r.generated[srcs[i]]
        counts, t, cr = eng.read_sinks()
        off = np.concatenate([[0], np.cumsum(counts)])
        for i, nd in sinks.items():
            ot, ocr = r.sinks[nd]
            np.testing.assert_array_equal(t[off[i]:off[i + 1]], ot)
            np.testing.assert_array_equal(cr[off[i]:off[i + 1]], ocr)
        return fan_in, eng.tandem_path()


def test_several_upstream_servers_per_server_match_the_oracle():
    """Fan-in (`Server(downstream=s)` for several Servers with the same `s`): up to four upstream Servers per Server on the passes
    (the downstream LP merges their forward logs by the roots' keys), more -- or an undecided tie -- on the single-heap loop."""
    seen, paths = {True: 0, False: 0}, {1: 0, 2: 0}
    for k in range(300):
        try:
            fan_in, path = _run_fan_in(_fan_in_case(k))
        except AssertionError as e:
            raise AssertionError(f"_fan_in_case({k}): {e}") from e
        seen[fan_in] += 1
        if fan_in:
            paths[path] += 1
    print("fan-in cases on the passes / on the single heap:", paths)
    assert seen[True] >= 100 and paths[1] > paths[2]


def test_more_upstream_servers_than_the_passes_merge_run_on_the_single_heap():
    sv = [dict(svc="exp", mean=0.05, conc=1, qcap=None, src=("poisson", 6.0), sink=False) for _ in range(6)]
    sv.append(dict(svc="exp", mean=0.01, conc=2, qcap=None, src=None, sink=True))
    fan_in, path = _run_fan_in(dict(servers=sv, down=[6] * 6 + [-1], end_s=3.0, seed=5))
    assert fan_in and path == 2
