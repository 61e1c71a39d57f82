"""Tandem queues -- `Server(..., downstream=<another Server>)` (components/server/server.py:64-122,271-272) -- as test cases:
seeded random specs, the oracle graph of a spec, the engine arrays of a spec (one station per Server, chain-major; station i's
entities draw from stream base i), and the comparison of everything the ABI reports.

A spec: {"chains": [{"arr": "poisson" | "constant", "rate": r, "stop_after_s": s | None, "sink": bool,
                     "stages": [{"svc": "exp" | "const", "mean": m, "conc": c, "qcap": q | None}, ...]}, ...],
         "end_s": T, "seed": k}"""
import numpy as np

from oracle import hs_oracle as O


def tandem_spec(k: int) -> dict:
    """Case k.  Every fourth case is a tie storm: constant arrivals and constant services on a 10 ms grid (zero-length services
    included), so that completions, forwards, ticks and the downstream Servers' own events share nanoseconds all the time."""
    rng = np.random.default_rng(900_000 + k)
    storm = k % 4 == 0
    chains = []
    for _ in range(int(rng.integers(1, 6))):
        n_stage = int(rng.integers(1, 5)) if rng.random() < 0.85 else 1
        stages = []
        for _s in range(n_stage):
            if storm:
                svc, mean = "const", float(rng.choice([0.0, 0.01, 0.02, 0.05, 0.1, 0.1, 0.2]))
            elif rng.random() < 0.3:
                svc, mean = "const", float(rng.choice([0.05, 0.1, 0.125, 0.3]))
            else:
                svc, mean = "exp", float(rng.choice([0.02, 0.05, 0.1, 0.15, 0.3]))
            stages.append(dict(svc=svc, mean=mean, conc=int(rng.choice([1, 1, 1, 2, 3, 4])),
                               qcap=None if rng.random() < 0.7 else int(rng.integers(0, 5))))
        if storm:
            arr, rate = "constant", float(rng.choice([5.0, 10.0, 10.0, 20.0, 50.0]))
        else:
            arr, rate = ("poisson", float(rng.choice([4.0, 8.0, 12.0, 20.0]))) if rng.random() < 0.7 else \
                        ("constant", float(rng.choice([5.0, 10.0, 16.0])))
        chains.append(dict(arr=arr, rate=rate, stop_after_s=None if rng.random() < 0.8 else float(rng.choice([0.5, 1.0, 2.0])),
                           sink=bool(rng.random() < 0.85), stages=stages))
    return dict(chains=chains, end_s=float(rng.choice([1.0, 2.0, 3.0, 5.0])), seed=int(rng.integers(1, 1 << 30)))


def has_tandem(spec) -> bool:
    return any(len(ch["stages"]) > 1 for ch in spec["chains"])


def station_index(spec):
    """[(chain, stage)] in station order and chain -> first station."""
    order, first = [], []
    for c, ch in enumerate(spec["chains"]):
        first.append(len(order))
        order += [(c, s) for s in range(len(ch["stages"]))]
    return order, first


def _ns(x: float) -> int:
    return int(x * 1e9)


def oracle_graph(spec):
    """Sources first (chain order = `sources=[...]`), then every chain's Servers head to tail and its Sink."""
    g = O.Graph()
    order, first = station_index(spec)
    srcs = []
    for c, ch in enumerate(spec["chains"]):
        srcs.append(g.source(O.ARR_POISSON if ch["arr"] == "poisson" else O.ARR_CONSTANT, ch["rate"],
                             stop_after_ns=-1 if ch["stop_after_s"] is None else _ns(ch["stop_after_s"]), stream_base=first[c]))
    servers, sinks = {}, {}
    for c, ch in enumerate(spec["chains"]):
        for s, sg in enumerate(ch["stages"]):
            servers[(c, s)] = g.server(O.LAT_EXP if sg["svc"] == "exp" else O.LAT_CONST, sg["mean"], concurrency=sg["conc"],
                                       queue_cap=-1 if sg["qcap"] is None else sg["qcap"], stream_base=first[c] + s)
        sinks[c] = g.sink() if ch["sink"] else -1
        g.target[srcs[c]] = servers[(c, 0)]
        for s in range(len(ch["stages"])):
            g.target[servers[(c, s)]] = servers[(c, s + 1)] if s + 1 < len(ch["stages"]) else sinks[c]
    return g, srcs, servers, sinks


def engine_arrays(spec):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays

    order, first = station_index(spec)
    n = len(order)
    st = StationArrays(n=n, src_kind=np.full(n, N.SRC_NONE, np.uint8), src_rate=np.ones(n), src_stop_after_ns=np.full(n, -1, np.int64),
                       concurrency=np.ones(n, np.int32), svc_kind=np.zeros(n, np.uint8), svc_mean_s=np.zeros(n),
                       queue_cap=np.full(n, -1, np.int64), egress=np.zeros(n, np.uint8))
    st.downstream_lp = np.full(n, -1, np.int32)
    for i, (c, s) in enumerate(order):
        ch = spec["chains"][c]
        sg = ch["stages"][s]
        if s == 0:
            st.src_kind[i] = N.SRC_POISSON if ch["arr"] == "poisson" else N.SRC_CONSTANT
            st.src_rate[i] = ch["rate"]
            st.src_stop_after_ns[i] = -1 if ch["stop_after_s"] is None else _ns(ch["stop_after_s"])
        else:
            st.src_rate[i] = ch["rate"]         # (no Source here; the rate still sizes the record logs)
        st.concurrency[i] = sg["conc"]
        st.svc_kind[i] = N.LAT_EXPONENTIAL if sg["svc"] == "exp" else N.LAT_CONSTANT
        st.svc_mean_s[i] = sg["mean"]
        st.queue_cap[i] = -1 if sg["qcap"] is None else sg["qcap"]
        last = s + 1 == len(ch["stages"])
        st.egress[i] = (N.EGRESS_SINK if ch["sink"] else N.EGRESS_NONE) if last else N.EGRESS_SERVER
        if not last:
            st.downstream_lp[i] = i + 1
    return st


def compare(spec, eng, r, srcs, servers, sinks):
    """Everything the ABI reports against the oracle run `r` of oracle_graph(spec)."""
    order, first = station_index(spec)
    s = eng.summary()
    assert s.events_processed == r.events_processed, (s.events_processed, r.events_processed)
    np.testing.assert_array_equal(s.events_by_kind, r.events_by_kind)
    assert s.final_time_ns == r.final_time_ns
    stats = eng.lp_stats()
    for i, (c, st) in enumerate(order):
        nd = servers[(c, st)]
        for k, ok in (("accepted", "accepted"), ("dropped", "dropped"), ("completed", "completed"), ("rejected", "rejected"),
                      ("queue_depth", "depth"), ("active", "active")):
            assert stats[k][i] == getattr(r, ok)[nd], (k, c, st, stats[k][i], getattr(r, ok)[nd])
        assert stats["total_service_s"][i] == r.total_service_s[nd], ("total_service_s", c, st)
        if st == 0:
            assert stats["generated"][i] == r.generated[srcs[c]], ("generated", c)
    counts, t, cr = eng.read_sinks()
    off = np.concatenate([[0], np.cumsum(counts)])
    for c, ch in enumerate(spec["chains"]):
        last = first[c] + len(ch["stages"]) - 1
        if sinks[c] >= 0:
            ot, ocr = r.sinks[sinks[c]]
            assert counts[last] == len(ot), ("sink count", c, counts[last], len(ot))
            np.testing.assert_array_equal(t[off[last]:off[last + 1]], ot, err_msg=f"sink t chain {c}")
            np.testing.assert_array_equal(cr[off[last]:off[last + 1]], ocr, err_msg=f"sink created chain {c}")


def fan_in_case(k):
    """Several chains' heads merging into shared Servers: a random forest of Servers (every Server has at most one downstream,
    any number of upstreams), Sources on some of them."""
    rng = np.random.default_rng(70_000 + k)
    n = int(rng.integers(3, 9))
    storm = k % 3 == 0
    down = [-1] * n
    for i in range(n - 1):
        if rng.random() < 0.8:
            down[i] = int(rng.integers(i + 1, n))            # forwards to a later Server: acyclic
    servers = []
    for i in range(n):
        if storm:
            svc, mean = "const", float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.1]))
        else:
            svc, mean = ("exp", float(rng.choice([0.02, 0.05, 0.1]))) if rng.random() < 0.7 else ("const", 0.05)
        servers.append(dict(svc=svc, mean=mean, conc=int(rng.choice([1, 1, 2, 3])), qcap=None if rng.random() < 0.7 else int(rng.integers(0, 4)),
                            src=None if (rng.random() < 0.35 and any(d == i for d in down)) else
                            (("constant", float(rng.choice([5.0, 10.0, 20.0]))) if storm else ("poisson", float(rng.choice([4.0, 8.0, 12.0])))),
                            sink=down[i] < 0 and rng.random() < 0.85))
    return dict(servers=servers, down=down, end_s=float(rng.choice([1.0, 2.0, 3.0])), seed=int(rng.integers(1, 1 << 30)))


def tandem_probe_case(k: int) -> dict:
    """tandem_spec(k) with Probes on some of its Servers and Requests injected with Simulation.schedule(): pre-run events next to
    tandem queues.  Returns the spec plus {"probes": [((chain, stage), metric name, interval_s)], "sched": [((chain, stage), t_s)]}."""
    rng = np.random.default_rng(4_400_000 + k)
    spec = tandem_spec(k)
    order, _first = station_index(spec)
    probes, seen = [], set()
    for _ in range(int(rng.integers(0, 4))):
        cs = order[int(rng.integers(0, len(order)))]
        if cs in seen:
            continue                                     # (one Probe per Server here: slot 0)
        seen.add(cs)
        probes.append((cs, str(rng.choice(["depth", "active_requests", "stats_accepted", "stats_dropped", "requests_completed"])),
                       float(rng.choice([0.05, 0.1, 0.25, 0.3, 0.5, 1.0]))))
    sched = []
    for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.6 else 0):
        cs = order[int(rng.integers(0, len(order)))]
        sched.append((cs, float(rng.choice([0.0, 0.1, 0.25, 0.5, 0.7, 1.0, round(float(rng.random()) * spec["end_s"], 6)]))))
    spec["probes"], spec["sched"] = probes, sched
    return spec


def run_tandem_probe_case(spec):
    """Oracle and engine on a tandem_probe_case: everything compare() checks plus every Probe's samples.  Returns the engine's
    (tandem path, prologue path)."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    g, srcs, servers, sinks = oracle_graph(spec)
    pnodes = [g.probe(servers[cs], N.PROBE_METRICS[m], iv) for cs, m, iv in spec["probes"]]
    end = _ns(spec["end_s"])
    r = O.run(g, end, seed=spec["seed"], schedule=[(servers[cs], _ns(t)) for cs, t in spec["sched"]])
    st = engine_arrays(spec)
    order, _first = station_index(spec)
    lp_of = {cs: i for i, cs in enumerate(order)}
    if spec["probes"]:
        st.probe_metric = np.full(st.n, N.PROBE_NONE, np.uint8)
        st.probe_interval_s = np.ones(st.n)
        for cs, m, iv in spec["probes"]:
            st.probe_metric[lp_of[cs]], st.probe_interval_s[lp_of[cs]] = N.PROBE_METRICS[m], iv
        st.probe_order = np.array([lp_of[cs] for cs, _, _ in spec["probes"]], np.int32)
    if spec["sched"]:
        per = [[] for _ in range(st.n)]
        for rank, (cs, t) in enumerate(spec["sched"]):
            per[lp_of[cs]].append((_ns(t), rank))
        for lst in per:
            lst.sort(key=lambda x: x[0])
        st.sched_off = np.concatenate([[0], np.cumsum([len(x) for x in per])]).astype(np.int64)
        st.sched_time_ns = np.array([t for lst in per for t, _ in lst], np.int64)
        st.sched_rank = np.array([rk for lst in per for _, rk in lst], np.int64)
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end, seed=spec["seed"]) as eng:
        eng.run_until(end)
        compare(spec, eng, r, srcs, servers, sinks)
        for (cs, _m, _iv), nd in zip(spec["probes"], pnodes):
            t, v = r.sinks[nd]
            pt, pv = eng.read_probe(lp_of[cs])
            np.testing.assert_array_equal(pt, t, err_msg=f"probe times {cs}")
            np.testing.assert_array_equal(pv, v, err_msg=f"probe values {cs}")
        return eng.tandem_path(), eng.prologue_path()
