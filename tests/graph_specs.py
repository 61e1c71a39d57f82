"""Graph specs (tests/golden/make_golden.py run_graph_case, tests/random_specs.py graph_spec) -> the PRODUCT's entity objects, wired
the way run_graph_case wires the reference's: `entities=servers + routers + links + sinks`, Sources in list order.  With that order
graph_engine.lower_general numbers the entity streams like the fixtures (Source k / Server s / link l / router r -> base k / s / l /
r), so a run through hs.Simulation is comparable with the live-reference goldens and with the oracle on helpers.oracle_graph."""
import numpy as np

import happy_simulator_amd as hs


def build(spec, extra_schedule=None):
    """Returns (Simulation, dict(sources, servers, links, routers, sinks))."""
    sinks = [hs.Sink(f"sink{j}") for j in range(spec["n_sinks"])]
    servers = [hs.Server(f"srv{i}", concurrency=sv.get("c", 1), service_time=hs.ExponentialLatency(sv["mean"]),
                         queue_capacity=sv.get("cap")) for i, sv in enumerate(spec["servers"])]
    links = []
    for l, lk in enumerate(spec["links"]):
        jit = None
        if lk.get("jm") is not None and lk.get("jk") == "exp":
            jit = hs.ExponentialLatency(lk["jm"])
        elif lk.get("jm") is not None and lk.get("jk") == "const":
            jit = hs.ConstantLatency(lk["jm"])
        links.append(hs.NetworkLink(f"link{l}", latency=hs.ConstantLatency(lk["lat"]), jitter=jit,
                                    packet_loss_rate=lk.get("loss", 0.0), egress=servers[lk["to"]]))
    routers = [None] * len(spec["routers"])
    lbs = [hs.LoadBalancer(f"lb{j}", backends=[servers[b] for b in lb["backends"]],
                           strategy=(hs.ConsistentHash(virtual_nodes=lb["vnodes"]) if lb["strategy"] == "chash" else
                                     hs.RoundRobin() if lb["strategy"] == "round_robin" else hs.Random()))
           for j, lb in enumerate(spec.get("lbs") or [])]
    pools = {"sink": sinks, "link": links, "router": routers, "server": servers, "lb": lbs}
    pending = list(range(len(routers)))
    while pending:
        for r in list(pending):
            tg = spec["routers"][r]["targets"]
            if all(k != "router" or routers[i] is not None for k, i in tg):
                routers[r] = hs.RandomRouter(f"router{r}", targets=[pools[k][i] for k, i in tg])
                pending.remove(r)
    for i, sv in enumerate(spec["servers"]):
        if sv.get("out") is not None:
            servers[i].downstream = pools[sv["out"][0]][sv["out"][1]]
    sources = []
    for k, sc in enumerate(spec["sources"]):
        make = hs.Source.poisson if sc["kind"] == "poisson" else hs.Source.constant
        to = servers[sc["to"]] if isinstance(sc["to"], int) else pools[sc["to"][0]][sc["to"][1]]
        if sc.get("n_clients"):
            sources.append(make(rate=sc["rate"], event_provider=hs.ClientKeyEventProvider(to, n_clients=sc["n_clients"]), name=f"src{k}"))
        else:
            sources.append(make(rate=sc["rate"], target=to, name=f"src{k}"))
    end = None if spec.get("end_s") is None else hs.Instant.from_seconds(spec["end_s"])      # None: auto-termination
    sim = hs.Simulation(end_time=end, sources=sources, entities=servers + lbs + routers + links + sinks, seed=spec["seed"])
    for (kind, idx), t_s in spec.get("schedule") or []:
        sim.schedule(hs.Event(time=hs.Instant.from_seconds(t_s), event_type="Request", target=pools[kind][idx]))
    for kind, idx, t_s in (extra_schedule or []):
        sim.schedule(hs.Event(time=hs.Instant.from_seconds(t_s), event_type="Request", target=pools[kind][idx]))
    return sim, dict(sources=sources, servers=servers, links=links, routers=routers, sinks=sinks, lbs=lbs)


def results(ents):
    """The arrays make_golden.run_graph_case stores, read off the product's objects."""
    out = {}
    s, v, l, r, k = ents["sources"], ents["servers"], ents["links"], ents["routers"], ents["sinks"]
    out["generated"] = np.array([x.generated_count for x in s], np.int64)
    out["accepted"] = np.array([x.stats_accepted for x in v], np.int64)
    out["dropped"] = np.array([x.stats_dropped for x in v], np.int64)
    out["completed"] = np.array([x._requests_completed for x in v], np.int64)
    out["rejected"] = np.array([x._requests_rejected for x in v], np.int64)
    out["depth"] = np.array([x.depth for x in v], np.int64)
    out["active"] = np.array([x.active_requests for x in v], np.int64)
    out["total_service_s"] = np.array([x._total_service_time for x in v], np.float64)
    out["received"] = np.array([x.events_received for x in k], np.int64)
    out["routed"] = np.array([x.stats_routed for x in r], np.int64)
    out["packets_sent"] = np.array([x.packets_sent for x in l], np.int64)
    out["packets_dropped"] = np.array([x.packets_dropped for x in l], np.int64)
    return out
