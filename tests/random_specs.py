"""Seeded random configurations (chains, rings, load-balancer topologies) with every lowered feature mixed in: bounded and
unbounded queues, concurrency, stop_after, no downstream, replicas, probes, time-varying profiles, scheduled Requests, shared
Sinks, lossy links.  tests/test_oracle_live_reference.py runs the LIVE reference on them against the oracle (build container);
tests/test_gpu_random.py runs the engines against the oracle (MI355X)."""
import numpy as np


def profile(rng):
    """A LinearRamp or Spike profile whose brackets stay narrow (start rates >= 4: see DESIGN.md section 1.2 for what a ramp
    from ~0 costs in the reference's own integrator)."""
    if rng.random() < 0.5:
        return ["ramp", float(np.round(rng.uniform(2.0, 10.0), 2)), float(rng.choice([4.0, 8.0, 25.0])),
                float(np.round(rng.uniform(2.0, 30.0), 2))]
    return ["spike", float(np.round(rng.uniform(3.0, 10.0), 2)), float(np.round(rng.uniform(20.0, 80.0), 1)),
            float(np.round(rng.uniform(0.5, 3.0), 2)), float(np.round(rng.uniform(0.5, 2.0), 2))]


def schedule(rng, n, end_s):
    ev = [[int(rng.integers(0, n)), float(np.round(rng.uniform(0.0, end_s * 1.05), int(rng.choice([1, 3, 9]))))]
          for _ in range(int(rng.integers(1, 9)))]
    return ev + ev[:int(rng.integers(0, 3))]                        # duplicates: same target, same instant


def station_spec(k):
    rng = np.random.default_rng(1000 + k)
    n = int(rng.integers(1, 7))
    conc = [int(rng.choice([1, 1, 2, 3])) for _ in range(n)]
    mean = [float(rng.choice([0.02, 0.05, 0.1, 0.25])) for _ in range(n)]
    rho = rng.uniform(0.3, 1.6, n)                                   # under- and overloaded chains
    spec = dict(
        name=f"live_station_{k}", n_chains=n, arr=[str(rng.choice(["poisson", "constant"])) for _ in range(n)],
        rate=[float(np.round(r * c / m, 3)) for r, c, m in zip(rho, conc, mean)],
        svc=[str(rng.choice(["exp", "exp", "const"])) for _ in range(n)], mean=mean, concurrency=conc,
        queue_cap=[None if rng.random() < 0.5 else int(rng.integers(0, 5)) for _ in range(n)],
        stop_after_s=None if rng.random() < 0.7 else float(np.round(rng.uniform(1.0, 5.0), 3)),
        downstream=bool(rng.random() < 0.85), end_s=float(np.round(rng.uniform(2.0, 8.0), 3)), rng="philox",
        seed=int(rng.integers(1, 10_000)), mode=str(rng.choice(["single", "single", "replicas"])), trace=True)
    if spec["mode"] == "single":                                     # the features below are per Simulation
        if rng.random() < 0.3:
            metrics = ["depth", "active_requests", "stats_accepted", "stats_dropped", "requests_completed", "generated_count"]
            if spec["downstream"]:
                metrics.append("events_received")
            spec["probes"] = [None if rng.random() < 0.4 else [str(rng.choice(metrics)), float(rng.choice([0.1, 0.25, 0.3, 0.5]))]
                              for _ in range(n)]
        if rng.random() < 0.3 and spec["stop_after_s"] is None:
            spec["profile"] = [None if rng.random() < 0.5 else profile(rng) for _ in range(n)]
        if rng.random() < 0.3:
            spec["schedule"] = schedule(rng, n, spec["end_s"])
        if rng.random() < 0.15 and spec["downstream"] and n > 1 and "probes" not in spec:
            spec["shared_sink"] = True
    return spec


def ring_spec(k):
    rng = np.random.default_rng(2000 + k)
    n = int(rng.integers(2, 8))
    spec = dict(
        name=f"live_ring_{k}", topology="ring", n=n, ext_rate=[float(rng.choice([0.0, 3.0, 5.0, 9.0])) for _ in range(n)],
        mean=float(rng.choice([0.05, 0.1])), concurrency=int(rng.choice([1, 1, 2])),
        queue_cap=None if rng.random() < 0.6 else int(rng.integers(1, 5)),
        lat_min=float(rng.choice([0.0005, 0.002, 0.01])), jitter_mean=None if rng.random() < 0.3 else float(rng.choice([0.002, 0.01])),
        end_s=float(np.round(rng.uniform(2.0, 6.0), 3)), seed=int(rng.integers(1, 10_000)), trace=True)
    if sum(spec["ext_rate"]) == 0.0:
        spec["ext_rate"][0] = 4.0
    if rng.random() < 0.4:
        spec["loss"] = [float(rng.choice([0.0, 0.1, 0.5, 1.0])) for _ in range(n)]
    if rng.random() < 0.3:
        spec["probes"] = [None if rng.random() < 0.4 else
                          [str(rng.choice(["depth", "active_requests", "stats_accepted", "requests_completed", "events_received"])),
                           float(rng.choice([0.1, 0.25, 0.3, 0.5]))] for _ in range(n)]
    if rng.random() < 0.3:
        spec["profile"] = [None if (rng.random() < 0.5 or spec["ext_rate"][i] == 0.0) else profile(rng) for i in range(n)]
    if rng.random() < 0.3:
        spec["schedule"] = schedule(rng, n, spec["end_s"])
    return spec


def lb_spec(k):
    rng = np.random.default_rng(3000 + k)
    S, B = int(rng.integers(1, 7)), int(rng.integers(1, 13))
    conc = [int(rng.choice([1, 1, 2, 3])) for _ in range(B)]
    return dict(
        name=f"live_lb_{k}", topology="lb", n_sources=S, n_backends=B,
        rate=[float(np.round(rng.uniform(2.0, 14.0 * B / S), 2)) for _ in range(S)], mean=float(rng.choice([0.05, 0.1, 0.2])),
        concurrency=conc, queue_cap=None if rng.random() < 0.5 else int(rng.integers(0, 4)),
        vnodes=int(rng.choice([1, 7, 50, 150])), n_clients=int(rng.choice([1, 13, 1000, 100000])),
        stop_after_s=None if rng.random() < 0.7 else float(np.round(rng.uniform(1.0, 4.0), 3)),
        shared_sink=bool(rng.random() < 0.6), end_s=float(np.round(rng.uniform(2.0, 6.0), 3)),
        seed=int(rng.integers(1, 10_000)), trace=True)


def lb_workers_spec(k):
    """A load-balancer configuration whose backends have up to 32 workers, loaded so that most of them are busy."""
    rng = np.random.default_rng(8800 + k)
    S, B = int(rng.integers(2, 6)), int(rng.integers(1, 5))
    conc = [int(rng.choice([1, 7, 16, 17, 24, 32])) for _ in range(B)]
    mean = float(rng.choice([0.4, 0.8]))
    load = float(rng.uniform(0.7, 1.3)) * sum(conc) / mean
    return dict(
        name=f"live_lb_workers_{k}", topology="lb", n_sources=S, n_backends=B, rate=[float(np.round(load / S, 2))] * S, mean=mean,
        concurrency=conc, queue_cap=None if rng.random() < 0.5 else int(rng.integers(0, 5)),
        vnodes=int(rng.choice([3, 40])), n_clients=int(rng.choice([50, 5000])), shared_sink=bool(rng.random() < 0.5),
        end_s=float(np.round(rng.uniform(3.0, 5.0), 3)), seed=int(rng.integers(1, 10_000)), trace=True)


def jitter_ring_spec(k):
    """ring_spec(k) with every NetworkLink's jitter drawn among ExponentialLatency / ConstantLatency (incl. sub-nanosecond and
    zero constants) / None -- the reference's presets use all three (components/network/conditions.py)."""
    spec = ring_spec(k)
    rng = np.random.default_rng(91_000 + k)
    n = spec["n"]
    kinds = [str(rng.choice(["exp", "const", "const", "none"])) for _ in range(n)]
    spec["jitter_kind"] = [None if q == "none" else q for q in kinds]
    spec["jitter_mean"] = [None if q == "none" else float(rng.choice([0.0001, 0.0013, 0.006, 0.0000004, 0.0])) if q == "const"
                           else float(rng.choice([0.002, 0.006, 0.01])) for q in kinds]
    spec["name"] = f"jitter_ring_{k}"
    return spec


def graph_spec(k):
    """A random graph of the lowered entity set beyond what the engines take today: routers with up to 8 targets (Sinks, links,
    Servers, other routers) and several upstreams, links shared by several senders, Servers behind Servers next to links, up to six
    Sources per Server -- the oracle's ground (live reference: tests/test_oracle_live_reference.py)."""
    rng = np.random.default_rng(77_000 + k)
    n_srv, n_snk = int(rng.integers(2, 8)), int(rng.integers(1, 4))
    n_lnk, n_rtr = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    links = [dict(lat=float(rng.choice([0.0005, 0.001, 0.004])), jk=[None, "exp", "const"][int(rng.integers(0, 3))],
                  jm=float(rng.choice([0.0002, 0.003, 0.008])), loss=float(rng.choice([0.0, 0.0, 0.1, 0.4])),
                  to=int(rng.integers(0, n_srv))) for _ in range(n_lnk)]
    for lk in links:
        if lk["jk"] is None:
            lk["jm"] = None
    routers = []
    for r in range(n_rtr):                                   # router r may target routers < r only (no router cycles)
        tg = [["sink", int(rng.integers(0, n_snk))]]         # every router can end a Request
        for _ in range(int(rng.integers(0, 8))):
            kind = str(rng.choice(["sink", "link", "link", "server"] + (["router"] if r > 0 else [])))
            tg.append([kind, int(rng.integers(0, {"sink": n_snk, "link": n_lnk, "server": n_srv, "router": max(r, 1)}[kind]))])
        order = rng.permutation(len(tg))
        routers.append(dict(targets=[tg[i] for i in order]))
    servers = []
    for i in range(n_srv):
        kind = str(rng.choice(["router", "router", "link", "sink", "server", "none"]))
        out = (None if kind == "none" else ["router", int(rng.integers(0, n_rtr))] if kind == "router" else
               ["link", int(rng.integers(0, n_lnk))] if kind == "link" else ["sink", int(rng.integers(0, n_snk))] if kind == "sink" else
               ["server", int(rng.integers(i + 1, n_srv))] if i + 1 < n_srv else ["sink", 0])   # Server -> later Server: no zero-delay cycles
        servers.append(dict(mean=float(rng.choice([0.02, 0.05, 0.1])), c=int(rng.choice([1, 1, 2, 4])),
                            cap=None if rng.random() < 0.6 else int(rng.integers(0, 5)), out=out))
    # a zero-delay cycle Server -> router -> Server would never end a nanosecond: routers may target Servers only with a larger index
    # than every Server that feeds them directly
    for r, rt in enumerate(routers):
        feeders = [i for i, sv in enumerate(servers) if sv["out"] == ["router", r]]
        lo = (max(feeders) + 1) if feeders else 0
        for t in rt["targets"]:
            if t[0] == "server" and t[1] < lo:
                t[0], t[1] = ("sink", int(rng.integers(0, n_snk))) if lo >= n_srv else ("server", int(rng.integers(lo, n_srv)))
            if t[0] == "router":                             # ... and a router reached through a router inherits the constraint: keep it simple
                t[0], t[1] = "sink", int(rng.integers(0, n_snk))
    n_src = int(rng.integers(1, 2 * n_srv + 1))
    sources = [dict(kind=str(rng.choice(["poisson", "poisson", "constant"])), rate=float(rng.choice([2.0, 4.0, 6.0, 9.0])),
                    to=int(rng.integers(0, n_srv)) if rng.random() < 0.7 else 0) for _ in range(n_src)]
    return dict(name=f"graph_{k}", topology="graph", n_sinks=n_snk, servers=servers, links=links, routers=routers, sources=sources,
                end_s=float(np.round(rng.uniform(3.0, 8.0), 3)), seed=int(rng.integers(1, 10_000)))


def lb_graph_spec(k):
    """graph_spec(k) with one to three LoadBalancers wired INTO it (round 6: several LoadBalancers, a LoadBalancer behind Servers and
    routers, `schedule()` on such graphs).  k % 3 != 0: most Sources hand out client ids (ClientKeyEventProvider), the rest are plain
    -- their Requests and the `schedule()`d ones take ConsistentHash's key-less fallback (a RoundRobin of the strategy's own); the
    LoadBalancers are ConsistentHash / RoundRobin anywhere and Random right behind Sources (a Random LoadBalancer chooses by the
    draw of a Source that aims at it).  k % 3 == 0: plain Sources, RoundRobin LoadBalancers only.  Both: Requests `schedule()`d for
    Servers, routers, links and the LoadBalancers themselves (never a Random one).  Backends of a LoadBalancer that a Server or
    router feeds have larger indices than the feeders (no zero-delay cycles, as in graph_spec)."""
    spec = graph_spec(k)
    rng = np.random.default_rng(79_000 + k)
    keyed = k % 3 != 0
    n_srv, n_rtr = len(spec["servers"]), len(spec["routers"])
    lbs = []
    for j in range(int(rng.integers(1, 4))):
        strategy = str(rng.choice(["chash", "chash", "round_robin", "random"] if keyed else ["round_robin"]))
        nb = int(rng.integers(1, min(4, n_srv) + 1))
        lo = int(rng.integers(0, n_srv - nb + 1))
        backends = sorted(int(b) for b in rng.choice(np.arange(lo, n_srv), size=nb, replace=False))
        rng.shuffle(backends)
        lbs.append(dict(strategy=strategy, vnodes=int(rng.choice([3, 17, 100])), backends=[int(b) for b in backends]))
    inner = [j for j, lb in enumerate(lbs) if lb["strategy"] != "random"]        # LoadBalancers that entities may forward to
    for i, sv in enumerate(spec["servers"]):
        ok = [j for j in inner if min(lbs[j]["backends"]) > i]
        if ok and rng.random() < 0.35:
            sv["out"] = ["lb", int(rng.choice(ok))]
    for r, rt in enumerate(spec["routers"]):
        feeders = [i for i, sv in enumerate(spec["servers"]) if sv["out"] == ["router", r]]
        lo = (max(feeders) + 1) if feeders else 0
        ok = [j for j in inner if min(lbs[j]["backends"]) >= lo]
        for t in rt["targets"]:
            if ok and rng.random() < 0.25:
                t[0], t[1] = "lb", int(rng.choice(ok))
    for sc in spec["sources"]:
        if keyed and rng.random() < 0.7:
            sc["n_clients"] = int(rng.choice([5, 50, 1000]))
        if rng.random() < 0.6:
            j = int(rng.integers(0, len(lbs)))
            sc["to"] = ["lb", j]
            if lbs[j]["strategy"] == "random":
                sc["n_clients"] = len(lbs[j]["backends"])
    for j, lb in enumerate(lbs):                                  # every LoadBalancer sees traffic
        if not any(sc["to"] == ["lb", j] for sc in spec["sources"]):
            nc = len(lb["backends"]) if lb["strategy"] == "random" else int(rng.choice([0, 5, 50, 1000])) if keyed else 0
            spec["sources"].append(dict(kind="poisson", rate=float(rng.choice([2.0, 6.0])), to=["lb", j], **({"n_clients": nc} if nc else {})))
    spec["lbs"] = lbs
    end = spec["end_s"]
    pools = [["server", n_srv], ["router", n_rtr], ["link", len(spec["links"])]] + [["lb", j] for j in inner] * 2
    sched = []
    for _ in range(int(rng.integers(2, 9)) if (not keyed or k % 2) else 0):
        kind, cnt = pools[int(rng.integers(0, len(pools)))]
        t = float(rng.choice([0.0, 0.5, 0.5, 1.0, float(np.round(rng.uniform(0.0, end), 3)), end, end + 0.5]))
        sched.append([[kind, cnt if kind == "lb" else int(rng.integers(0, cnt))], t])
    if sched:
        spec["schedule"] = sched
    spec["name"] = f"lb_graph_{k}"
    return spec


def union_spec(specs, name="union"):
    """Several graph specs (graph_spec / lb_graph_spec) side by side in ONE Simulation: disjoint entity sets, indices shifted, the
    Sources of all of them in one list (spec by spec), one end and one seed (the first spec's)."""
    import copy

    out = dict(name=name, topology="graph", n_sinks=0, servers=[], links=[], routers=[], lbs=[], sources=[], schedule=[],
               end_s=specs[0]["end_s"], seed=specs[0]["seed"])
    for sp in specs:
        sp = copy.deepcopy(sp)
        off = {"sink": out["n_sinks"], "server": len(out["servers"]), "link": len(out["links"]), "router": len(out["routers"]),
               "lb": len(out["lbs"])}

        def sh(ref):
            return None if ref is None else [ref[0], ref[1] + off[ref[0]]]

        for sv in sp["servers"]:
            sv["out"] = sh(sv.get("out"))
        for lk in sp["links"]:
            lk["to"] += off["server"]
        for rt in sp["routers"]:
            rt["targets"] = [sh(t) for t in rt["targets"]]
        for lb in sp.get("lbs") or []:
            lb["backends"] = [b + off["server"] for b in lb["backends"]]
        for sc in sp["sources"]:
            sc["to"] = sc["to"] + off["server"] if isinstance(sc["to"], int) else sh(sc["to"])
        for ref, t in sp.get("schedule") or []:
            out["schedule"].append([sh(ref), min(t, out["end_s"] + 0.5)])
        out["n_sinks"] += sp["n_sinks"]
        for k in ("servers", "links", "routers", "sources"):
            out[k].extend(sp[k])
        out["lbs"].extend(sp.get("lbs") or [])
    return out


def tie_spec(k):
    """Tie storms: lock-step constant-rate sources, constant service times that are multiples of one another, Requests
    scheduled at the start instant and at the sources' own tick times, c up to 16, zero-capacity queues -- every same-nanosecond
    order the reference's sort-index ledger decides.  (Reference vs oracle only: across LPs and at the start instant the engines
    document tie-break deviations)"""
    rng = np.random.default_rng(50_000 + k)
    n = int(rng.integers(2, 7))
    base_rate = float(rng.choice([2.0, 4.0, 5.0, 10.0]))
    spec = dict(name=f"tie_{k}", n_chains=n, arr=[str(rng.choice(["constant", "constant", "poisson"])) for _ in range(n)],
                rate=[base_rate * float(rng.choice([1.0, 1.0, 2.0, 0.5])) for _ in range(n)],
                svc=[str(rng.choice(["const", "const", "exp"])) for _ in range(n)],
                mean=[float(rng.choice([0.05, 0.1, 0.2, 0.25, 0.5])) for _ in range(n)],
                concurrency=[int(rng.choice([1, 2, 4, 8, 16])) for _ in range(n)],
                queue_cap=[None if rng.random() < 0.4 else int(rng.integers(0, 3)) for _ in range(n)],
                stop_after_s=None if rng.random() < 0.6 else float(rng.choice([1.0, 2.0, 2.5])),
                downstream=bool(rng.random() < 0.8), end_s=float(rng.choice([2.0, 3.0, 4.5])), rng="philox",
                seed=int(rng.integers(1, 10_000)), mode="single", trace=True)
    if rng.random() < 0.5:
        spec["schedule"] = [[int(rng.integers(0, n)), float(rng.choice([0.0, 0.0, 0.1, 0.2, 0.25, 0.5, 1.0, 2.0]))]
                            for _ in range(int(rng.integers(1, 8)))]
    if rng.random() < 0.3:
        spec["probes"] = [None if rng.random() < 0.4 else
                          [str(rng.choice(["depth", "active_requests", "stats_accepted", "generated_count"])),
                           float(rng.choice([0.1, 0.25, 0.5]))] for _ in range(n)]
    if rng.random() < 0.15 and spec["downstream"] and "probes" not in spec:
        spec["shared_sink"] = True
    return spec


def multi_source_spec(k):
    """Several Sources feeding one Server (entities of their own, lowered onto the station's slots 1..3): a tie storm (even k)
    or a random station configuration (odd k) with up to three further Sources on some chains, listed after their chain's
    first Source or all in front (`sources_order`) -- lock-step constant Sources on ONE Server are the tie storm the prologue
    has to order exactly."""
    rng = np.random.default_rng(70_000 + k)
    spec = tie_spec(1000 + k) if k % 2 == 0 else station_spec(1000 + k)
    spec["name"] = f"multi_source_{k}"
    spec.pop("profile", None)                                       # (not lowered next to further Sources)
    spec.pop("shared_sink", None)
    n = spec["n_chains"]
    rate = spec["rate"] if isinstance(spec["rate"], list) else [spec["rate"]] * n
    more = []
    for i in range(n):
        cnt = int(rng.choice([0, 1, 1, 2, 3]))
        more.append([[str(rng.choice(["constant", "poisson"])), float(rate[i]) * float(rng.choice([1.0, 1.0, 0.5, 2.0]))]
                     for _ in range(cnt)] or None)
    if not any(more):
        more[0] = [["constant", float(rate[0])]]
    spec["more_sources"] = more
    if rng.random() < 0.5:
        spec["sources_order"] = "extras_first"
    if spec.get("probes"):                                          # generated_count is sampled on a chain's first Source only
        spec["probes"] = [None if pr is None or pr[0] == "generated_count" else pr for pr in spec["probes"]]
    return spec


def multi_source_ring_spec(k):
    """A random ring (ring_spec) whose stations carry up to three further Sources on their Servers, in two list orders."""
    rng = np.random.default_rng(80_000 + k)
    spec = ring_spec(500 + k)
    spec["name"] = f"multi_source_ring_{k}"
    n = spec["n"]
    prof = spec.get("profile") or [None] * n
    more = []
    for i in range(n):
        cnt = 0 if (spec["ext_rate"][i] == 0.0 or prof[i] is not None) else int(rng.choice([0, 1, 1, 2, 3]))
        more.append([[str(rng.choice(["constant", "poisson"])), float(rng.choice([2.0, 4.0, 5.0]))] for _ in range(cnt)] or None)
    if not any(more):
        i = next(j for j in range(n) if spec["ext_rate"][j] > 0.0)
        if prof[i] is not None:
            spec["profile"][i] = None
        more[i] = [["constant", 4.0]]
    spec["more_sources"] = more
    if rng.random() < 0.5:
        spec["sources_order"] = "extras_first"
    return spec


def lb_probe_spec(k):
    """A random load-balancer configuration (lb_spec) with probes on some backend Servers and on its Sink(s)."""
    rng = np.random.default_rng(90_000 + k)
    spec = lb_spec(700 + k)
    spec["name"] = f"lb_probes_{k}"
    B = spec["n_backends"]
    metrics = ["depth", "active_requests", "stats_accepted", "stats_dropped", "requests_completed"]
    pr = [["server", int(rng.integers(0, B)), str(rng.choice(metrics)), float(rng.choice([0.1, 0.25, 0.3, 0.5]))]
          for _ in range(int(rng.integers(1, 6)))]
    if rng.random() < 0.7:
        pr.insert(int(rng.integers(0, len(pr) + 1)),
                  ["sink", 0 if spec["shared_sink"] else int(rng.integers(0, B)), "events_received", float(rng.choice([0.2, 0.35]))])
    if spec.get("stop_after_s") is None and rng.random() < 0.5:
        pr.append(["source", int(rng.integers(0, spec["n_sources"])), "generated_count", float(rng.choice([0.15, 0.4]))])
    spec["probes"] = pr
    return spec


def lb_profile_spec(k):
    """A random load-balancer configuration whose Sources follow ramps / spikes (Source.with_profile), some with probes."""
    rng = np.random.default_rng(95_000 + k)
    spec = lb_probe_spec(300 + k) if k % 2 else lb_spec(900 + k)
    spec["name"] = f"lb_profiles_{k}"
    spec["stop_after_s"] = None
    if spec.get("probes"):
        spec["probes"] = [pr for pr in spec["probes"]]
    S = spec["n_sources"]
    prof = [None if rng.random() < 0.35 else profile(rng) for _ in range(S)]
    if not any(prof):
        prof[0] = profile(rng)
    spec["profile"] = prof
    return spec


def lb_strategy_spec(k):
    """A random load-balancer configuration (lb_spec) under the LoadBalancer's default RoundRobin strategy (even k) or Random (odd k);
    every fourth with probes on backend Servers / Sinks."""
    spec = lb_probe_spec(500 + k) if k % 4 == 3 else lb_spec(1500 + k)
    spec["name"] = f"lb_strategy_{k}"
    spec["strategy"] = "round_robin" if k % 2 == 0 else "random"
    spec["vnodes"], spec["n_clients"] = 1, (1 if k % 2 == 0 else spec["n_backends"])
    if spec.get("probes"):                        # (a Source with stop_after is not probed: its ticks without Requests are not logged)
        spec["probes"] = [pr for pr in spec["probes"] if not (pr[0] == "source" and spec.get("stop_after_s") is not None)]
    return spec
