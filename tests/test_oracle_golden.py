"""CPU: the C oracle (oracle/hs_oracle.c) against the live-reference goldens.

This is what PINS the oracle: every fixture under tests/golden/ was produced by the
upstream reference itself (tests/golden/make_golden.py).  Compared bit-exactly:
total events, per-kind histogram, final time, every per-chain statistic, every
Sink record (ns + latency_s), and -- where recorded -- the full processed-event
trace including the reference's `_sort_index` of each event.
"""
import numpy as np
import pytest

import helpers as H
from oracle import hs_oracle as O


@pytest.mark.parametrize("name", H.golden_names())
def test_oracle_matches_reference_golden(name):
    check_oracle_against_station_golden(H.Golden(name))


def check_oracle_against_station_golden(gold):
    """Oracle == the reference's results for a chains spec (also used with freshly generated results by
    tests/test_oracle_live_reference.py)."""
    spec = gold.spec
    want_trace = "trace" in gold.arrays
    runs = H.run_oracle_for_spec(spec, trace_cap=(len(gold.trace) + 16) if want_trace else 0)
    assert [r.events_processed for _, _, r in runs] == gold.meta["total_events"]
    assert [r.final_time_ns for _, _, r in runs] == gold.meta["final_ns"]
    for (chain_ids, nodes, r), dur in zip(runs, gold.meta["duration_s"]):
        # duration_s = (t_last - t_start).to_seconds()  (core/simulation.py:546)
        assert float(r.final_time_ns) / 1e9 == dur
        for c in chain_ids:
            src, srv, snk = nodes[c]
            assert (r.generated[src] if src >= 0 else 0) == gold.generated[c]
            for (cc, slot), nd in r.xsrc_nodes.items():               # the Server's further Sources
                if cc == c:
                    assert r.generated[nd] == gold.generated_more[slot - 1, c]
            assert r.accepted[srv] == gold.accepted[c]
            assert r.dropped[srv] == gold.dropped[c]
            assert r.completed[srv] == gold.completed[c]
            assert r.rejected[srv] == gold.rejected[c]
            assert r.depth[srv] == gold.depth[c]
            assert r.active[srv] == gold.active[c]
            assert r.total_service_s[srv] == gold.total_service_s[c]  # same fp64 sum order
            if snk >= 0:
                t, created = r.sinks[snk]
                gt, glat = gold.sink_records(c)
                assert r.received[snk] == gold.received[c]
                np.testing.assert_array_equal(t, gt)
                # Sink latency rule: (t - created_at).to_seconds() (components/common.py:39-40)
                np.testing.assert_array_equal((t - created).astype(np.float64) / 1e9, glat)
    if "probe_t_ns" in gold.arrays:          # Probe samples: (time ns, getattr(target, metric)) in sampling order
        for chain_ids, nodes, r in runs:
            for c in chain_ids:
                for j in range(gold.n_probe_slots):
                    gt, gv = gold.probe_samples(c, j)
                    if (c, j) in r.probe_nodes_all:
                        t, v = r.sinks[r.probe_nodes_all[(c, j)]]
                        np.testing.assert_array_equal(t, gt)
                        np.testing.assert_array_equal(v, gv)
                    else:
                        assert len(gt) == 0
    if want_trace:
        assert spec["mode"] == "single"
        (chain_ids, nodes, r), = runs
        node_chain = {}
        for c, trio in nodes.items():
            for nd in trio:
                if nd >= 0:
                    node_chain[nd] = c
        for (c, _slot), nd in r.probe_nodes_all.items():
            node_chain[nd] = c
        for (c, _slot), nd in r.xsrc_nodes.items():
            node_chain[nd] = c
        if spec.get("shared_sink"):          # make_golden's node table labels the one shared Sink with the LAST chain
            node_chain[nodes[chain_ids[0]][2]] = chain_ids[-1]
        t, k, nd, ix = r.trace
        got = np.stack([t, k.astype(np.int64), np.array([node_chain[x] for x in nd], np.int64), ix], axis=1)
        np.testing.assert_array_equal(got, gold.trace)


def test_known_reference_numbers():
    """Numbers quoted in SURVEY.md 8(c) (captured from the reference during the survey)."""
    g = H.Golden("quickstart_mt42")
    assert g.meta["total_events"] == [3621] and g.generated[0] == 483 and g.received[0] == 482
    assert list(g.sink_t_ns[:3]) == [160664539, 437456572, 631679305]
    assert H.Golden("const_r8").meta["total_events"] == [4318]
    assert H.Golden("const_r10").meta["total_events"] == [4805]
    assert H.Golden("const_r12_overload").meta["total_events"] == [4445]


@pytest.mark.parametrize("name", H.golden_names("ring"))
def test_oracle_matches_reference_ring_golden(name):
    """Ring of stations built from reference components only (Server -> RandomRouter -> [Sink | NetworkLink ->
    next Server]); the live reference ran with Philox-plugged streams (make_golden.py run_ring_case)."""
    check_oracle_against_ring_golden(H.Golden(name))


def check_oracle_against_graph_golden(gold):
    """make_golden.run_graph_case vs the oracle on the same graph: every Source / Server / router / link / Sink statistic and every
    Sink record (time and latency), bit for bit."""
    spec = gold.spec
    g, nodes = H.oracle_graph(spec)
    r = O.run(g, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"], schedule=H.oracle_graph_schedule(spec, nodes))
    assert [r.events_processed] == gold.meta["total_events"]
    assert [r.final_time_ns] == gold.meta["final_ns"]
    np.testing.assert_array_equal(r.generated[nodes["source"]], gold.generated)
    for j, nd in enumerate(nodes["lb"]):                     # LoadBalancer.stats, BackendInfo.total_requests, RoundRobin._index
        np.testing.assert_array_equal(r.lbs[nd]["stats"], gold.lb_stats[j], err_msg=f"lb {j}")
        lo, hi = gold.lb_backend_off[j], gold.lb_backend_off[j + 1]
        np.testing.assert_array_equal(r.lbs[nd]["total_requests"], gold.lb_backend_total_requests[lo:hi], err_msg=f"lb {j} backends")
        assert r.lbs[nd]["strategy_index"] == gold.lb_rr_index[j]          # RoundRobin._index / ConsistentHash's key-less fallback's / -1
    srv = nodes["server"]
    for k, arr in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed), ("rejected", r.rejected),
                   ("depth", r.depth), ("active", r.active), ("total_service_s", r.total_service_s)):
        np.testing.assert_array_equal(arr[srv], gold.arrays[k], err_msg=k)
    if nodes["router"]:
        np.testing.assert_array_equal(r.routed[nodes["router"]], gold.routed)
    if nodes["link"]:
        np.testing.assert_array_equal(r.packets_sent[nodes["link"]], gold.packets_sent)
        np.testing.assert_array_equal(r.dropped[nodes["link"]], gold.packets_dropped)
    for j, nd in enumerate(nodes["sink"]):
        t, created = r.sinks[nd]
        gt, glat = gold.sink_records(j)
        np.testing.assert_array_equal(t, gt, err_msg=f"sink {j}")
        np.testing.assert_array_equal((t - created).astype(np.float64) / 1e9, glat, err_msg=f"sink {j} latencies")


def check_oracle_against_ring_golden(gold):
    spec = gold.spec
    want_trace = "trace" in gold.arrays
    g, nodes = H.oracle_ring_graph(spec)
    r = O.run(g, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"],
              trace_cap=(len(gold.trace) + 16) if want_trace else 0,
              schedule=[(nodes[c]["srv"], t) for c, t in H.ring_params(spec)["schedule"]])
    assert [r.events_processed] == gold.meta["total_events"]
    assert [r.final_time_ns] == gold.meta["final_ns"]
    for i in range(spec["n"]):
        nd = nodes[i]
        if nd["src"] >= 0:
            assert r.generated[nd["src"]] == gold.generated[i]
        for j in (1, 2, 3):                                  # further Sources of the station's Server
            if f"src{j}" in nd:
                assert r.generated[nd[f"src{j}"]] == gold.generated_more[j - 1, i]
        for k, arr in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed),
                       ("rejected", r.rejected), ("depth", r.depth), ("active", r.active),
                       ("total_service_s", r.total_service_s)):
            assert arr[nd["srv"]] == gold.arrays[k][i], (k, i)
        assert r.routed[nd["rtr"]] == gold.routed[i]
        assert r.packets_sent[nd["lnk"]] == gold.packets_sent[i]
        if "packets_dropped" in gold.arrays:
            assert r.dropped[nd["lnk"]] == gold.packets_dropped[i]
            assert gold.bytes_transmitted[i] == 0          # no payload_size in the metadata: bandwidth has no effect
        t, created = r.sinks[nd["snk"]]
        gt, glat = gold.sink_records(i)
        np.testing.assert_array_equal(t, gt)
        np.testing.assert_array_equal((t - created).astype(np.float64) / 1e9, glat)
        if "probe_t_ns" in gold.arrays:                     # probes on networked stations
            for j in range(gold.n_probe_slots):
                gt, gv = gold.probe_samples(i, j)
                key = "prb" if j == 0 else f"prb{j}"
                if key in nd:
                    pt, pv = r.sinks[nd[key]]
                    np.testing.assert_array_equal(pt, gt)
                    np.testing.assert_array_equal(pv, gv)
                else:
                    assert len(gt) == 0
    if want_trace:
        node_station = {v: i for i, d in nodes.items() for v in d.values() if v >= 0}
        t, k, nd, ix = r.trace
        got = np.stack([t, k.astype(np.int64), np.array([node_station[x] for x in nd], np.int64), ix], axis=1)
        np.testing.assert_array_equal(got, gold.trace)


@pytest.mark.parametrize("name", H.golden_names("lb"))
def test_oracle_matches_reference_lb_golden(name):
    """BASELINE configs[4] in miniature: Sources -> LoadBalancer(ConsistentHash) -> Server backends -> Sink(s), reference
    components only, client ids / arrivals / services from Philox-plugged streams (make_golden.py run_lb_case).  Pins the
    oracle's md5 ring, its key -> backend selection, the LB's two extra events per request (Request@LB, `_lb_response`
    fired by the completion hook when the backend's enqueue handler returns) and the whole trace with sort indices."""
    check_oracle_against_lb_golden(H.Golden(name))


def check_oracle_against_lb_golden(gold):
    spec = gold.spec
    want_trace = "trace" in gold.arrays
    g, p = H.oracle_lb_graph(spec)
    S, B = p["S"], p["B"]
    chash = spec.get("strategy", "chash") == "chash"          # (RoundRobin / Random: no ring to compare)
    r = O.run(g, p["end_ns"], seed=spec["seed"], trace_cap=(len(gold.trace) + 16) if want_trace else 0,
              lb_probe=len(gold.client_backend) if chash else 0)
    assert [r.events_processed] == gold.meta["total_events"]
    assert [r.final_time_ns] == gold.meta["final_ns"]
    lb = r.lbs[S]
    np.testing.assert_array_equal(lb["stats"], gold.lb_stats)
    np.testing.assert_array_equal(lb["total_requests"], gold.backend_total_requests)
    if chash:
        np.testing.assert_array_equal(lb["ring_backend"] - (S + 1), gold.ring_backend)       # the sorted md5 ring
        np.testing.assert_array_equal(np.array(lb["select"]) - (S + 1), gold.client_backend)  # ConsistentHash.select
    elif "rr_index" in gold.arrays:
        assert int(gold.rr_index[0]) == lb["stats"][1]                                        # RoundRobin._index = selects made
    np.testing.assert_array_equal(r.generated[:S], gold.generated)
    be = slice(S + 1, S + 1 + B)
    for k in ("accepted", "dropped", "completed", "rejected", "depth", "active", "total_service_s"):
        np.testing.assert_array_equal(getattr(r, k)[be], gold.arrays[k], err_msg=k)
    for j, nd in enumerate(g.lb_probe_nodes):                # Probe samples (time ns, value), then the probe's "sink" goes
        pt, pv = r.sinks.pop(nd)
        a, b = gold.probe_off[j], gold.probe_off[j + 1]
        np.testing.assert_array_equal(pt, gold.probe_t_ns[a:b])
        np.testing.assert_array_equal(pv, gold.probe_v[a:b])
    sinks = sorted(r.sinks)
    np.testing.assert_array_equal([r.received[i] for i in sinks], gold.received)
    t = np.concatenate([r.sinks[i][0] for i in sinks])
    cr = np.concatenate([r.sinks[i][1] for i in sinks])
    np.testing.assert_array_equal(t, gold.sink_t_ns)
    np.testing.assert_array_equal((t - cr).astype(np.float64) / 1e9, gold.sink_latency_s)
    if want_trace:
        tt, k, nd, ix = r.trace
        got = np.stack([tt, k.astype(np.int64), nd.astype(np.int64), ix], axis=1)
        np.testing.assert_array_equal(got, gold.trace)


@pytest.mark.parametrize("name", H.golden_names("tandem"))
def test_oracle_matches_reference_tandem_golden(name):
    check_oracle_against_tandem_golden(H.Golden(name))


def check_oracle_against_tandem_golden(gold):
    """Tandem queues, `Server(downstream=<Server>)` (tests/tandem_specs.py; make_golden.run_tandem_case): the oracle against the
    reference on every Server's statistics, every Sink record and the whole processed-event trace with sort indices."""
    import tandem_specs as TS

    spec = gold.spec
    order, first = TS.station_index(spec)
    g, srcs, servers, sinks = TS.oracle_graph(spec)
    # (tandem_probe_case: Probes on Servers and injected Requests next to the tandem queues; JSON turned the keys into lists)
    from happy_simulator_amd import _native as N
    probes = [(tuple(cs), m, iv) for cs, m, iv in spec.get("probes") or []]
    sched = [(tuple(cs), t) for cs, t in spec.get("sched") or []]
    pnodes = [g.probe(servers[cs], N.PROBE_METRICS[m], iv) for cs, m, iv in probes]
    r = O.run(g, int(spec["end_s"] * 1e9), seed=spec["seed"], trace_cap=len(gold.trace) + 16,
              schedule=[(servers[cs], int(t * 1e9)) for cs, t in sched])
    for j, nd in enumerate(pnodes):
        off = gold.arrays["probe_off"]
        pt, pv = r.sinks[nd]
        np.testing.assert_array_equal(pt, gold.arrays["probe_t_ns"][off[j]:off[j + 1]], err_msg=f"probe {j} times")
        np.testing.assert_array_equal(pv, gold.arrays["probe_v"][off[j]:off[j + 1]], err_msg=f"probe {j} values")
    assert [r.events_processed] == gold.meta["total_events"]
    assert [r.final_time_ns] == gold.meta["final_ns"]
    for i, (c, st) in enumerate(order):
        nd = servers[(c, st)]
        for k in ("accepted", "dropped", "completed", "rejected", "depth", "active"):
            assert getattr(r, k)[nd] == gold.arrays[k][i], (k, c, st)
        assert r.total_service_s[nd] == gold.total_service_s[i], ("total_service_s", c, st)
    np.testing.assert_array_equal([r.generated[s] for s in srcs], gold.generated)
    for c, ch in enumerate(spec["chains"]):
        gt, glat = gold.sink_records(c)
        if sinks[c] < 0:
            assert len(gt) == 0
            continue
        t, cr = r.sinks[sinks[c]]
        np.testing.assert_array_equal(t, gt)
        np.testing.assert_array_equal((t - cr) / 1e9, glat)            # latency_s = (t - created_at).to_seconds()
    node_station = {srcs[c]: first[c] for c in range(len(srcs))}
    node_station.update({nd: first[c] + st for (c, st), nd in servers.items()})
    node_station.update({sinks[c]: first[c] + len(ch["stages"]) - 1 for c, ch in enumerate(spec["chains"]) if sinks[c] >= 0})
    node_station.update({nd: first[cs[0]] + cs[1] for nd, (cs, _m, _iv) in zip(pnodes, probes)})
    t, k, nd, ix = r.trace
    got = np.stack([t, k.astype(np.int64), np.array([node_station[x] for x in nd], np.int64), ix], axis=1)
    np.testing.assert_array_equal(got, gold.trace)


def check_oracle_against_fan_in_reference(case, ref):
    """fan_in_case forests (make_golden.run_fan_in_case): the oracle against the live reference on every Server's statistics,
    every Sink record and the whole processed-event trace with sort indices."""
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
    r = O.run(g, int(case["end_s"] * 1e9), seed=case["seed"], trace_cap=len(ref["trace"]) + 16)
    assert r.events_processed == ref["total_events"]
    assert r.final_time_ns == ref["final_ns"]
    for i in range(n):
        for k, ok in (("accepted", "accepted"), ("dropped", "dropped"), ("completed", "completed"), ("rejected", "rejected"),
                      ("depth", "depth"), ("active", "active")):
            assert getattr(r, ok)[nodes[i]] == ref[k][i], (k, i)
        assert r.total_service_s[nodes[i]] == ref["total_service_s"][i], ("total_service_s", i)
        if i in srcs:
            assert r.generated[srcs[i]] == ref["generated"][i], ("generated", i)
    for i, (gt, glat) in ref["sinks"].items():
        if down[i] >= 0:
            continue                                # (a Sink flag on a Server that forwards: no Sink was built)
        t, cr = r.sinks[sinks[i]]
        np.testing.assert_array_equal(t, gt)
        np.testing.assert_array_equal((t - cr) / 1e9, glat)
    station = {nd: i for i, nd in srcs.items()}
    station.update({nd: i for i, nd in enumerate(nodes)})
    station.update({nd: i for i, nd in sinks.items()})
    t, k, nd, ix = r.trace
    got = np.stack([t, k.astype(np.int64), np.array([station[x] for x in nd], np.int64), ix], axis=1)
    np.testing.assert_array_equal(got, ref["trace"])


def test_oracle_md5_known_answers():
    """RFC 1321 appendix A.5 test suite + lengths around the 56/64-byte padding boundary."""
    import hashlib

    rfc = {b"": "d41d8cd98f00b204e9800998ecf8427e", b"a": "0cc175b9c0f1b6a831c399e269772661",
           b"abc": "900150983cd24fb0d6963f7d28e17f72", b"message digest": "f96b697d7cb7938d525a2f31aaf161d0",
           b"abcdefghijklmnopqrstuvwxyz": "c3fcd3d76192e4007dfb496cca67e13b"}
    for msg, hexd in rfc.items():
        assert O.md5(msg).hex() == hexd
    for n in (55, 56, 57, 63, 64, 65, 119, 120, 200):
        msg = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert O.md5(msg) == hashlib.md5(msg).digest()


# ---- the reference's own ParallelSimulation (fixtures: make_golden.PARALLEL_CASES) -----------------------------
def test_oracle_matches_reference_parallel_simulation_without_links():
    """`ParallelSimulation(partitions).run()` of the live reference, one Philox-plugged M/M/c chain per partition: every partition
    is a Simulation of its own (own heap, own one event beyond end_time) with the model-wide entity numbering."""
    gold = H.Golden("parallel_philox_independent_6")
    assert gold.spec["mode"] == "partitions"
    check_oracle_against_station_golden(gold)
    par = gold.meta["parallel"]
    assert par["total_events"] == sum(gold.meta["total_events"]) and par["duration_s"] == max(gold.meta["duration_s"])
    assert par["total_windows"] == 0 and par["total_cross_partition_events"] == 0


def test_reference_parallel_simulation_known_answers():
    """The reference's own known-answer tests (tests/integration/test_parallel_simulation.py:75-109,239-289), as its
    ParallelSimulation computed them when the fixture was made: 100 / 100, 100, 50 with no windows, parallel == sequential."""
    subs = H.Golden("parallel_ref_counters").meta["counters"]
    assert subs[0]["totals"] == [100, 100] and subs[1]["totals"] == [100] and subs[2]["totals"] == [50]
    assert subs[2]["total_windows"] == 0 and subs[2]["total_cross_partition_events"] == 0
    for sub in subs:
        assert sub["totals"] == sub["sequential_totals"]
        assert sub["total_events"] == sum(sub["partition_events"]) and sub["duration_s"] == max(sub["partition_duration_s"])
    # the C oracle on the same partitions: a constant Source straight into a collector, one heap per partition
    spec = H.Golden("parallel_ref_counters").spec
    for sub, want in zip(spec["counters"], subs):
        for i, rate in enumerate(sub["rates"]):
            g = O.Graph()
            src, snk = g.source(O.ARR_CONSTANT, float(rate)), g.sink()
            g.target[src] = snk
            r = O.run(g, H.ns_from_seconds(sub["duration"]))
            assert r.received[snk] == want["totals"][i] and r.generated[src] == want["generated"][i]
            assert r.events_processed == want["partition_events"][i]
            assert float(r.final_time_ns) / 1e9 == want["partition_duration_s"][i]


@pytest.mark.parametrize("name", ["parallel_linked_pipeline", "parallel_linked_three_stages", "parallel_linked_hazard"])
def test_oracle_matches_linked_partition_pipelines(name):
    """Linked partitions.  The fixture holds three runs of the live reference: its own windowed ParallelSimulation (hops = entities
    that deliver in the future, the only thing its coordinator accepts), the same entities in ONE Simulation, and the topology with
    library NetworkLinks in ONE Simulation (`seqnet_*`) -- the last is what the engine's linked partitions compute, and the
    oracle's single heap must equal it to the bit.  Where the reference's windowed run is self-consistent it agrees with that on
    every Sink record; `parallel_linked_hazard` is the failing-by-design case where it is not."""
    gold = H.Golden(name)
    spec = gold.spec
    g, srv, lnk, snk, src = H.pipeline_oracle_graph(spec)
    r = O.run(g, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"])
    sn = gold.meta["seq_network"]
    assert r.events_processed == sn["total_events"] and r.final_time_ns == sn["final_ns"]
    flat = [x for row in srv for x in row]
    np.testing.assert_array_equal(r.generated[src], gold.seqnet_generated)
    for key in ("accepted", "dropped", "completed", "depth", "active", "total_service_s"):
        np.testing.assert_array_equal(getattr(r, key)[flat], gold.arrays["seqnet_" + key], err_msg=key)
    np.testing.assert_array_equal(r.received[snk], gold.seqnet_received)
    np.testing.assert_array_equal(r.packets_sent[[x for row in lnk for x in row]], np.asarray(sn["packets_sent"]))
    np.testing.assert_array_equal(np.concatenate([r.sinks[k][0] for k in snk]), gold.seqnet_sink_t_ns)
    lat = np.concatenate([(r.sinks[k][0] - r.sinks[k][1]).astype(np.float64) / 1e9 for k in snk])
    np.testing.assert_array_equal(lat, gold.seqnet_sink_latency_s)
    # the reference against itself
    win = gold.meta["windowed"]
    drops = sum(win["time_travel_drops"].values())
    if name == "parallel_linked_hazard":
        # failing by design: the windowed coordinator loses requests its own sequential run delivers
        assert not gold.meta["windowed_equals_sequential"] and drops > 100
        assert gold.win_received.sum() < 0.6 * gold.seqfut_received.sum()
        lanes = spec["lanes"]                  # what the downstream partition accepted + what it dropped as time travel = what was sent to it
        assert abs(int(gold.win_accepted[lanes:].sum()) + drops - int(gold.win_hop_entered.sum())) <= lanes
    else:
        assert drops == 0
        np.testing.assert_array_equal(gold.win_sink_t_ns, gold.seqnet_sink_t_ns)          # every Sink record, to the nanosecond
        np.testing.assert_array_equal(gold.win_sink_latency_s, gold.seqnet_sink_latency_s)
        np.testing.assert_array_equal(gold.win_completed, gold.seqnet_completed)
        # what differs is bounded by the events beyond end_time: every PARTITION runs one of its own in the windowed run
        assert 0 <= (gold.win_accepted - gold.seqfut_accepted).sum() <= len(spec["stages"])
        assert 0 <= (gold.win_generated - gold.seqfut_generated).sum() <= 1
        assert 0 <= win["total_events"] - gold.meta["seq_future"]["total_events"] <= len(spec["stages"]) - 1
    # a NetworkLink is two events per hop (Request@Link + its continuation), the future-delivering entity one
    # (+- 1: the one event beyond end_time is of a different kind in the two runs)
    assert abs(sn["total_events"] - gold.meta["seq_future"]["total_events"] - sum(sn["packets_sent"])) <= 1


@pytest.mark.parametrize("name", ["parallel_linked_loss", "parallel_linked_loss_three"])
def test_oracle_matches_partition_links_that_lose_packets(name):
    """`PartitionLink(packet_loss=p)` (parallel/link.py:31-39): the reference's coordinator drops a cross-partition event at the
    exchange when `self._rng.random() < link.packet_loss` -- ONE `random.Random(seed)` for the whole run
    (parallel/coordinator.py:68,203-205), drawn in (window, source partition, outbox) order.  While every lossy PartitionLink leaves
    one partition that is the partition's own processing order of the sending events, i.e. the order of ONE heap -- the oracle draws
    CPython's MT19937 `random()` there (hso_graph.ploss, hso_params.coord_seed).  Against the LIVE windowed run of the fixture
    (tests/golden/make_golden.py run_parallel_linked_case): every Server statistic, every Sink record, what each hop took in."""
    gold = H.Golden(name)
    spec = gold.spec
    assert sum(1 for p in spec["packet_loss"] if p > 0) == 1
    g, srv, lnk, snk, src = H.pipeline_oracle_graph(spec)
    r = O.run(g, H.ns_from_seconds(spec["end_s"]), seed=spec["seed"], coord_seed=spec["coord_seed"])
    win = gold.meta["windowed"]
    assert sum(win["time_travel_drops"].values()) == 0                 # the windowed run is self-consistent: it lost nothing else
    flat = [x for row in srv for x in row]
    links = [x for row in lnk for x in row]
    np.testing.assert_array_equal(np.concatenate([r.sinks[k][0] for k in snk]), gold.win_sink_t_ns)
    lat = np.concatenate([(r.sinks[k][0] - r.sinks[k][1]).astype(np.float64) / 1e9 for k in snk])
    np.testing.assert_array_equal(lat, gold.win_sink_latency_s)
    np.testing.assert_array_equal(r.received[snk], gold.win_received)
    np.testing.assert_array_equal(r.completed[flat], gold.win_completed)
    np.testing.assert_array_equal(r.total_service_s[flat], gold.win_total_service_s)
    # the partitions of the windowed run each process ONE event of their own beyond end_time (the single heap: one in all), so an
    # upstream Server may have taken in one more request there, and a hop one more packet
    lanes = spec["lanes"]
    d_acc = gold.win_accepted - r.accepted[flat]
    assert (d_acc >= 0).all() and d_acc.sum() <= len(spec["stages"])
    d_hop = gold.win_hop_entered - (r.packets_sent[links] + r.dropped[links] + _in_flight(r, links, g))
    assert (d_hop >= 0).all() and d_hop.sum() <= len(spec["stages"]) - 1
    # what was lost: the lossy hops dropped a share of what entered them close to p, the others nothing
    for k, p in enumerate(spec["packet_loss"]):
        dropped = int(r.dropped[lnk[k]].sum())
        entered = int((r.packets_sent[lnk[k]] + r.dropped[lnk[k]]).sum())
        assert (dropped == 0) if p == 0 else abs(dropped / entered - p) < 0.08
    # delivered cross-partition events: the coordinator counts what it injected (incl. events beyond end_time)
    delivered = int(r.packets_sent[links].sum())
    assert 0 <= win["total_cross_partition_events"] - delivered <= len(links)


def _in_flight(r, links, g):
    """Packets that entered a link before end_time and arrive behind it (neither sent-through nor dropped in the oracle's counters):
    the difference between what the link's source Server forwarded and what the link counted."""
    out = np.zeros(len(links), np.int64)
    for i, l in enumerate(links):
        up = [n for n in range(len(g)) if g.target[n] == l]
        out[i] = int(r.completed[up].sum()) - int(r.packets_sent[l] + r.dropped[l])
    return out


@pytest.mark.parametrize("name", H.golden_names("graph"))
def test_oracle_matches_reference_on_arbitrary_graphs(name):
    """Graphs the engines still refuse -- a RandomRouter with eight targets (Sinks, links, Servers), NetworkLinks with several senders,
    Servers behind Servers next to links, seven Sources on one Server -- run by the live reference (make_golden.py run_graph_case):
    the oracle reproduces every statistic and Sink record.  The ground the next lifted refusals are checked against."""
    check_oracle_against_graph_golden(H.Golden(name))
