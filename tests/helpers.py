"""Shared test helpers: golden loading and spec -> oracle graph lowering."""
from __future__ import annotations

import glob
import json
import os

import math

import numpy as np

from oracle import hs_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(topology="chains"):
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))

    def topo(n):
        return ("ring" if n.startswith("ring_") else "lb" if n.startswith("lb_") else "tandem" if n.startswith("tandem_")
                else "parallel" if n.startswith("parallel_") else "graph" if n.startswith("graph_") else "chains")

    return [n for n in names if topo(n) == topology]


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.spec = self.meta["spec"]
        self.arrays = {k: z[k] for k in z.files if k != "meta"}

    @classmethod
    def from_results(cls, out, meta):
        """Wrap what make_golden.run_case / run_ring_case returned (a run of the live reference in this process)."""
        g = cls.__new__(cls)
        g.meta = json.loads(json.dumps(meta))          # the same round trip a fixture goes through
        g.spec = g.meta["spec"]
        g.arrays = {k: np.asarray(v) for k, v in out.items() if k != "meta"}
        return g

    def __getattr__(self, k):
        try:
            return self.arrays[k]
        except KeyError as e:
            raise AttributeError(k) from e

    @property
    def n_probe_slots(self):
        return int(self.arrays["probe_slots"][0]) if "probe_slots" in self.arrays else 1

    def probe_samples(self, chain, slot=0):
        """(times ns, values) the reference's probe number `slot` of the chain appended to its Data (empty: no such probe)."""
        k = chain * self.n_probe_slots + slot
        a, b = self.probe_off[k], self.probe_off[k + 1]
        return self.probe_t_ns[a:b], self.probe_v[a:b]

    def sink_records(self, chain):
        a, b = self.sink_off[chain], self.sink_off[chain + 1]
        return self.sink_t_ns[a:b], self.sink_latency_s[a:b]


def per_chain(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def ns_from_seconds(x: float) -> int:
    """Instant.from_seconds (core/temporal.py:188-207)."""
    if isinstance(x, int):
        return x * 1_000_000_000
    return int(x * 1_000_000_000)


def _probe_lists(spec, n):
    """spec["probes"][c] is None, one [metric, interval] pair, or a list of pairs (several probes on one chain)."""
    out = []
    for pr in (spec.get("probes") or [None] * n):
        out.append([] if pr is None else [list(q) for q in pr] if isinstance(pr[0], (list, tuple)) else [list(pr)])
    return out


def _first_probes(spec, n):
    return [prs[0] if prs else None for prs in _probe_lists(spec, n)]


def spec_chain_params(spec):
    n = spec["n_chains"]
    return dict(
        n=n,
        arr=[O.ARR_POISSON if a == "poisson" else O.ARR_CONSTANT for a in per_chain(spec["arr"], n)],
        rate=[float(r) for r in per_chain(spec["rate"], n)],
        svc=[O.LAT_EXP if s == "exp" else O.LAT_CONST for s in per_chain(spec["svc"], n)],
        mean=[float(m) for m in per_chain(spec["mean"], n)],
        conc=per_chain(spec.get("concurrency", 1), n),
        qcap=[-1 if q is None else int(q) for q in per_chain(spec.get("queue_cap"), n)],
        stop_ns=-1 if spec.get("stop_after_s") is None else ns_from_seconds(spec["stop_after_s"]),
        downstream=spec.get("downstream", True),
        end_ns=ns_from_seconds(spec["end_s"]),
        profile=[None if pr is None else tuple(pr) for pr in (spec.get("profile") or [None] * n)],
        probes=[None if pr is None else (pr[0], float(pr[1])) for pr in _first_probes(spec, n)],
        probe_list=[[(m, float(iv)) for m, iv in prs] for prs in _probe_lists(spec, n)],       # every probe of a chain, in order
        # Simulation.schedule(): (chain, time ns) in the caller's construction order; a chain with rate 0 has no Source
        schedule=[(int(c), ns_from_seconds(float(t))) for c, t in (spec.get("schedule") or [])],
        no_source=[float(r) == 0.0 and pr is None
                   for r, pr in zip(per_chain(spec["rate"], n), spec.get("profile") or [None] * n)],
        # several Sources feeding one Server: per chain [(arrival kind, rate)] of the further ones
        more_sources=[[(O.ARR_POISSON if a == "poisson" else O.ARR_CONSTANT, float(r)) for a, r in (xs or [])]
                      for xs in (spec.get("more_sources") or [None] * n)],
    )


def source_plan(spec, chain_ids):
    """make_golden.source_plan: `sources=[...]` of a case as (chain, slot) pairs in list order and, per chain, its Sources in
    slot order [(oracle arrival kind, rate, is the chain's `rate` / `arr` / `profile` Source)]; a chain's slot = the position
    among ITS Sources in the list."""
    n = spec["n_chains"]
    rate, arr = per_chain(spec["rate"], n), per_chain(spec["arr"], n)
    profiles = spec.get("profile") or [None] * n
    more = spec.get("more_sources") or [None] * n
    kind = lambda a: O.ARR_POISSON if a == "poisson" else O.ARR_CONSTANT
    has_first = {c: not (float(rate[c]) == 0.0 and profiles[c] is None) for c in chain_ids}
    firsts = [(c, (kind(arr[c]), float(rate[c]), True)) for c in chain_ids if has_first[c]]
    if spec.get("sources_order") == "extras_first":
        listed = [(c, (kind(xa), float(xr), False)) for c in reversed(chain_ids) for xa, xr in (more[c] or [])] + firsts
    else:
        listed = []
        for c in chain_ids:
            listed += ([(c, (kind(arr[c]), float(rate[c]), True))] * has_first[c] +
                       [(c, (kind(xa), float(xr), False)) for xa, xr in (more[c] or [])])
    slot_plan = {c: [] for c in chain_ids}
    order = []
    for c, what in listed:
        order.append((c, len(slot_plan[c])))
        slot_plan[c].append(what)
    return order, slot_plan


def source_order_for(spec, chain_ids):
    return source_plan(spec, list(chain_ids))[0]


def xsrc_stream_base(base, j):
    return (1 << 40) | (base << 2) | j


# Probe metric name -> (entity of the chain that carries it, oracle metric id, engine metric id); make_golden.PROBE_METRICS
PROBE_METRICS = {"depth": ("server", 0), "active_requests": ("server", 1), "stats_accepted": ("server", 2),
                 "stats_dropped": ("server", 3), "requests_completed": ("server", 4), "events_received": ("sink", 5),
                 "generated_count": ("source", 6)}


def _probe_arrays(st, p, n):
    """Probes of every chain into StationArrays: slot 0 in probe_metric, slots 1..3 in probe_metric_more, and the construction
    order (chain-major, slot-minor -- the order make_golden lists them in `probes=[...]`)."""
    from happy_simulator_amd import _native as N
    if not any(p["probe_list"]):
        return
    st.probe_metric = np.full(n, N.PROBE_NONE, np.uint8)
    st.probe_interval_s = np.ones(n, np.float64)
    more = max(len(prs) for prs in p["probe_list"]) > 1
    if more:
        st.probe_metric_more = np.full((3, n), N.PROBE_NONE, np.uint8)
        st.probe_interval_more = np.ones((3, n), np.float64)
    order, slots = [], []
    for i, prs in enumerate(p["probe_list"]):
        for j, pr in enumerate(prs):
            if j == 0:
                st.probe_metric[i], st.probe_interval_s[i] = PROBE_METRICS[pr[0]][1], pr[1]
            else:
                st.probe_metric_more[j - 1, i], st.probe_interval_more[j - 1, i] = PROBE_METRICS[pr[0]][1], pr[1]
            order.append(i)
            slots.append(j)
    if more:
        st.probe_order = np.asarray(order, np.int32)
        st.probe_slot_order = np.asarray(slots, np.uint8)


def oracle_graph_for(spec, chain_ids, stream_bases):
    """Oracle node graph for the given chains: sources first (list order), then server[, sink] per chain.
    Returns (graph, chain -> (src, srv, snk) node ids)."""
    p = spec_chain_params(spec)
    g = O.Graph()
    nodes = {}
    base_of = dict(zip(chain_ids, stream_bases))
    made = {}
    order, slot_plan = source_plan(spec, list(chain_ids))
    for c, slot in order:                                          # Source nodes in `sources=[...]` order
        xa, xr, is_first = slot_plan[c][slot]
        made[(c, slot)] = g.source(xa, xr, stop_after_ns=p["stop_ns"],
                                   stream_base=base_of[c] if slot == 0 else xsrc_stream_base(base_of[c], slot - 1),
                                   profile=p["profile"][c] if is_first else None)
    srcs = [made.get((c, 0), -1) for c in chain_ids]
    g.xsrc_nodes = {k: v for k, v in made.items() if k[1] > 0}
    for k, (c, base) in enumerate(zip(chain_ids, stream_bases)):
        sv = g.server(p["svc"][c], p["mean"][c], concurrency=p["conc"][c], queue_cap=p["qcap"][c], stream_base=base)
        if spec.get("shared_sink"):        # one Sink node behind every server; its records are reported under the first chain
            if k == 0:
                shared_sk = g.sink()
            sk = shared_sk
        else:
            sk = g.sink() if p["downstream"] else -1
        if srcs[k] >= 0:
            g.target[srcs[k]] = sv
        for (cc, slot), nd in g.xsrc_nodes.items():
            if cc == c:
                g.target[nd] = sv
        g.target[sv] = sk
        nodes[c] = (srcs[k], sv, sk if not (spec.get("shared_sink") and k > 0) else -1)
    g.probe_nodes, g.probe_nodes_all = {}, {}
    for c in chain_ids:                                   # probes start after every source, in list order
        for j, pr in enumerate(p["probe_list"][c]):
            who, mid = PROBE_METRICS[pr[0]]
            tgt = nodes[c][{"source": 0, "server": 1, "sink": 2}[who]]
            g.probe_nodes_all[(c, j)] = g.probe(tgt, mid, pr[1])
            if j == 0:
                g.probe_nodes[c] = g.probe_nodes_all[(c, j)]
    return g, nodes


def run_oracle_for_spec(spec, trace_cap=0):
    """Run the C oracle the way the golden was produced (single heap, or one heap per replica).
    Returns a list of (chain_ids, nodes, result)."""
    n = spec["n_chains"]
    p = spec_chain_params(spec)
    rng = O.RNG_MT19937 if spec["rng"] == "mt" else O.RNG_PHILOX
    runs = []
    if spec["mode"] == "single":
        groups = [(list(range(n)), spec["seed"], list(range(n)))]
    elif spec["mode"] == "partitions":   # ParallelSimulation without links: a Simulation per chain, ONE seed, stream base = partition
        groups = [([i], spec["seed"], [i]) for i in range(n)]
    else:
        groups = [([i], spec["seed"] + i, [0]) for i in range(n)]
    for chain_ids, seed, bases in groups:
        g, nodes = oracle_graph_for(spec, chain_ids, bases)
        r = O.run(g, p["end_ns"], seed=seed, rng_mode=rng, mt_seed_py=seed & 0xFFFFFFFF,
                  mt_seed_np=seed & 0xFFFFFFFF, trace_cap=trace_cap,
                  schedule=[(nodes[c][1], t) for c, t in p["schedule"] if c in nodes])
        r.probe_nodes, r.probe_nodes_all, r.xsrc_nodes = g.probe_nodes, g.probe_nodes_all, g.xsrc_nodes
        runs.append((chain_ids, nodes, r))
    return runs


def ring_params(spec):
    n = spec["n"]
    return dict(
        n=n, ext_rate=[float(r) for r in per_chain(spec["ext_rate"], n)], mean=float(spec["mean"]),
        conc=int(spec.get("concurrency", 1)), qcap=-1 if spec.get("queue_cap") is None else int(spec["queue_cap"]),
        lat_min=float(spec["lat_min"]), jitter_mean=spec.get("jitter_mean"), end_ns=ns_from_seconds(spec["end_s"]),
        # per link: (kind, mean) with kind "exp" | "const" | None -- spec["jitter_kind"] (default "exp") and spec["jitter_mean"] may be lists
        jitter=[(None, 0.0) if (m is None or k is None) else (k, float(m))
                for k, m in zip(per_chain(spec.get("jitter_kind", "exp"), n), per_chain(spec.get("jitter_mean"), n))],
        loss=[float(x) for x in per_chain(spec.get("loss", 0.0), n)], p_targets=2,
        probes=[None if pr is None else (pr[0], float(pr[1])) for pr in _first_probes(spec, n)],
        probe_list=[[(m, float(iv)) for m, iv in prs] for prs in _probe_lists(spec, n)],
        profile=[None if pr is None else tuple(pr) for pr in (spec.get("profile") or [None] * n)],
        schedule=[(int(c), ns_from_seconds(float(t))) for c, t in (spec.get("schedule") or [])])


def ring_source_plan(spec):
    """make_golden.ring_source_plan with oracle arrival kinds: (station, slot) in `sources=[...]` order and per station its
    Sources in slot order [(kind, rate, is the station's ext_rate / profile Source)]."""
    n = spec["n"]
    rate = [float(r) for r in per_chain(spec["ext_rate"], n)]
    more = spec.get("more_sources") or [None] * n
    kind = lambda a: O.ARR_POISSON if a == "poisson" else O.ARR_CONSTANT
    firsts = [(i, (O.ARR_POISSON, rate[i], True)) for i in range(n) if rate[i] > 0]
    if spec.get("sources_order") == "extras_first":
        listed = [(i, (kind(xa), float(xr), False)) for i in reversed(range(n)) for xa, xr in (more[i] or [])] + firsts
    else:
        listed = []
        for i in range(n):
            listed += ([(i, (O.ARR_POISSON, rate[i], True))] * (rate[i] > 0) +
                       [(i, (kind(xa), float(xr), False)) for xa, xr in (more[i] or [])])
    slot_plan = {i: [] for i in range(n)}
    order = []
    for i, what in listed:
        order.append((i, len(slot_plan[i])))
        slot_plan[i].append(what)
    return order, slot_plan


def oracle_ring_graph(spec):
    """Oracle nodes for a ring golden: sources (list order) first, then per station server, router, sink, link.
    Station i's entities all use stream base i.  Returns (graph, {i: dict(src, srv, rtr, snk, lnk)})."""
    p = ring_params(spec)
    n = p["n"]
    g = O.Graph()
    nodes = {i: {} for i in range(n)}
    order, slot_plan = ring_source_plan(spec)
    for i in range(n):
        nodes[i]["src"] = -1
    for i, slot in order:                                 # Source nodes in `sources=[...]` order
        xa, xr, is_first = slot_plan[i][slot]
        nodes[i]["src" if slot == 0 else f"src{slot}"] = g.source(
            xa, xr, stream_base=i if slot == 0 else xsrc_stream_base(i, slot - 1), profile=p["profile"][i] if is_first else None)
    for i in range(n):
        nodes[i]["srv"] = g.server(O.LAT_EXP, p["mean"], concurrency=p["conc"], queue_cap=p["qcap"], stream_base=i)
        nodes[i]["snk"] = g.sink()
        jk, jm = p["jitter"][i]
        nodes[i]["lnk"] = g.link(p["lat_min"], None if jk is None else jm, stream_base=i, loss=p["loss"][i], jitter_kind=jk or "exp")
        pat = (spec.get("rt_pattern") or ["sl"] * n)[i]
        nodes[i]["rtr"] = g.router([nodes[i]["snk"] if ch == "s" else nodes[i]["lnk"] for ch in pat], stream_base=i)
    for i in range(n):
        for key, nd in nodes[i].items():
            if key.startswith("src") and nd >= 0:
                g.target[nd] = nodes[i]["srv"]
        g.target[nodes[i]["srv"]] = nodes[i]["rtr"]
        g.target[nodes[i]["lnk"]] = nodes[(i + 1) % n]["srv"]
    for i in range(n):                                    # probes start after every source, in list order
        for j, pr in enumerate(p["probe_list"][i]):
            who, mid = PROBE_METRICS[pr[0]]
            nodes[i]["prb" if j == 0 else f"prb{j}"] = g.probe(nodes[i][{"source": "src", "server": "srv", "sink": "snk"}[who]],
                                                              mid, pr[1])
    return g, nodes


def oracle_graph(spec):
    """Oracle nodes of a graph golden (tests/golden/make_golden.py run_graph_case): sources in list order first, then sinks, servers,
    links, routers.  Returns (graph, {"source": [...], "sink": [...], "server": [...], "link": [...], "router": [...]})."""
    g = O.Graph()
    nodes = {"source": [], "sink": [], "server": [], "link": [], "router": [None] * len(spec["routers"]), "lb": []}
    for k, sc in enumerate(spec["sources"]):
        nodes["source"].append(g.source(O.ARR_POISSON if sc["kind"] == "poisson" else O.ARR_CONSTANT, sc["rate"], stream_base=k,
                                        n_clients=sc.get("n_clients", 0)))
    for _ in range(spec["n_sinks"]):
        nodes["sink"].append(g.sink())
    for i, sv in enumerate(spec["servers"]):
        nodes["server"].append(g.server(O.LAT_EXP, sv["mean"], concurrency=sv.get("c", 1),
                                        queue_cap=-1 if sv.get("cap") is None else sv["cap"], stream_base=i, name=f"srv{i}"))
    for lb in spec.get("lbs") or []:                # strategy by the vnodes field (hs_oracle.c on_lb): > 0 ConsistentHash, 0 RoundRobin, -1 Random
        vn = {"chash": lb.get("vnodes", 100), "round_robin": 0, "random": -1}[lb["strategy"]]
        nodes["lb"].append(g.load_balancer([nodes["server"][b] for b in lb["backends"]], vn))
    for l, lk in enumerate(spec["links"]):
        jk = lk.get("jk") if lk.get("jm") is not None else None
        nodes["link"].append(g.link(lk["lat"], None if jk is None else lk["jm"], stream_base=l, loss=lk.get("loss", 0.0),
                                    jitter_kind=jk or "exp"))
    pending = list(range(len(spec["routers"])))
    while pending:
        for r in list(pending):
            tg = spec["routers"][r]["targets"]
            if all(k != "router" or nodes["router"][i] is not None for k, i in tg):
                nodes["router"][r] = g.router([nodes[k][i] for k, i in tg], stream_base=r)
                pending.remove(r)
    for k, sc in enumerate(spec["sources"]):
        g.target[nodes["source"][k]] = nodes["server"][sc["to"]] if isinstance(sc["to"], int) else nodes[sc["to"][0]][sc["to"][1]]
    for i, sv in enumerate(spec["servers"]):
        if sv.get("out") is not None:
            g.target[nodes["server"][i]] = nodes[sv["out"][0]][sv["out"][1]]
    for l, lk in enumerate(spec["links"]):
        g.target[nodes["link"][l]] = nodes["server"][lk["to"]]
    return g, nodes


def oracle_graph_schedule(spec, nodes):
    """spec["schedule"] = [[[kind, index], seconds], ...] -> hso_schedule's (node, ns) list, in call order."""
    return [(nodes[ref[0]][ref[1]], ns_from_seconds(t)) for ref, t in spec.get("schedule") or []]


def lb_params(spec):
    S, B = spec["n_sources"], spec["n_backends"]
    return dict(
        S=S, B=B, rate=[float(r) for r in per_chain(spec["rate"], S)], mean=[float(m) for m in per_chain(spec["mean"], B)],
        conc=[int(c) for c in per_chain(spec.get("concurrency", 1), B)],
        qcap=[-1 if q is None else int(q) for q in per_chain(spec.get("queue_cap"), B)],
        vnodes=int(spec["vnodes"]), n_clients=int(spec["n_clients"]),
        stop_ns=-1 if spec.get("stop_after_s") is None else ns_from_seconds(spec["stop_after_s"]),
        shared_sink=bool(spec.get("shared_sink", True)), end_ns=ns_from_seconds(spec["end_s"]),
        strategy=spec.get("strategy", "chash"))


def oracle_lb_graph(spec):
    """Oracle nodes of a load-balancer golden (make_golden.py run_lb_case): sources 0..S-1, LB = S, backends S+1..S+B,
    then the Sink(s).  Returns (graph, params)."""
    p = lb_params(spec)
    g = O.lb_topology(p["S"], p["B"], p["rate"], p["mean"], p["vnodes"], p["n_clients"], p["conc"], p["qcap"],
                      p["stop_ns"], p["shared_sink"], strategy=p["strategy"])
    for i, pr in enumerate(spec.get("profile") or []):    # Source.with_profile in front of the LoadBalancer
        if pr is not None:
            g.prof_kind[i] = O.PROF_LINEAR_RAMP if pr[0] == "ramp" else O.PROF_SPIKE
            g.prof_p[i] = tuple(float(x) for x in pr[1:]) + (0.0,) * (5 - len(pr))
    # probes on backend Servers / Sinks: [["server" | "sink", index, metric, interval], ...]; nodes after the Sinks
    g.lb_probe_nodes = []
    S, B = p["S"], p["B"]
    for who, idx, metric, interval in spec.get("probes") or []:
        target = {"server": S + 1 + idx, "sink": S + 1 + B + idx, "source": idx}[who]
        g.lb_probe_nodes.append(g.probe(target, PROBE_METRICS[metric][1], float(interval)))
    return g, p


def sched_arrays(n, schedule, per_station=False):
    """Simulation.schedule() calls [(station, t_ns), ...] in construction order -> (sched_off, sched_time_ns per station
    ascending with ties in call order, sched_rank: for every entry of sched_time_ns its position among the calls)."""
    per = [[] for _ in range(n)]
    for j, (c, t) in enumerate(schedule):
        per[c].append((t, j))
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in per])
    flat = [tj for x in per for tj in sorted(x)]          # (t, j): ascending time, ties in call order
    times = np.array([t for t, _ in flat], np.int64)
    rank = np.array([j for _, j in flat], np.int64)
    if per_station:                 # every station is a Simulation of its own: positions among ITS OWN schedule() calls
        pos = {}
        for c, x in enumerate(per):
            for r, (_, j) in enumerate(sorted(x, key=lambda tj: tj[1])):
                pos[j] = r
        rank = np.array([pos[j] for _, j in flat], np.int64)
    return off, times, rank


# ----------------------------------------------------------------------------------------------
# HIP engine side (GPU tests only)
# ----------------------------------------------------------------------------------------------
def engine_for_spec(spec, log_capacity=0, horizon_ns=None, flags=0):
    """Build a StationEngine for a golden spec (all chains resident, one LP per chain)."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    p = spec_chain_params(spec)
    n = p["n"]
    st = StationArrays(
        n=n,
        src_kind=np.array([N.SRC_NONE if ns else N.SRC_POISSON if a == O.ARR_POISSON else N.SRC_CONSTANT
                           for a, ns in zip(p["arr"], p["no_source"])], np.uint8),
        src_rate=np.array([1.0 if ns else r for r, ns in zip(p["rate"], p["no_source"])], np.float64),
        src_stop_after_ns=np.full(n, p["stop_ns"], np.int64),
        concurrency=np.array(p["conc"], np.int32),
        svc_kind=np.array([N.LAT_EXPONENTIAL if s == O.LAT_EXP else N.LAT_CONSTANT for s in p["svc"]], np.uint8),
        svc_mean_s=np.array(p["mean"], np.float64),
        queue_cap=np.array(p["qcap"], np.int64),
        egress=np.full(n, N.EGRESS_SINK if p["downstream"] else N.EGRESS_NONE, np.uint8),
    )
    if any(pr is not None for pr in p["profile"]):
        st.src_profile_kind = np.zeros(n, np.uint8)
        st.src_profile_params = np.zeros((n, 4), np.float64)
        for i, pr in enumerate(p["profile"]):
            if pr is None:
                continue
            st.src_profile_kind[i] = N.PROF_LINEAR_RAMP if pr[0] == "ramp" else N.PROF_SPIKE
            st.src_profile_params[i, :len(pr) - 1] = pr[1:]
            st.src_rate[i] = max(pr[2], pr[3]) if pr[0] == "ramp" else max(pr[1], pr[2])     # peak: sizes the logs
    _probe_arrays(st, p, n)
    if any(p["more_sources"]):
        order, slot_plan = source_plan(spec, list(range(n)))
        st.src_more_kind = np.full((3, n), N.SRC_NONE, np.uint8)
        st.src_more_rate = np.ones((3, n), np.float64)
        st.src_more_stop_after_ns = np.full((3, n), p["stop_ns"], np.int64)
        for i in range(n):
            for slot, (xa, xr, _) in enumerate(slot_plan[i]):
                k = N.SRC_POISSON if xa == O.ARR_POISSON else N.SRC_CONSTANT
                if slot == 0:
                    st.src_kind[i], st.src_rate[i] = k, xr
                else:
                    st.src_more_kind[slot - 1, i], st.src_more_rate[slot - 1, i] = k, xr
        st.source_order = np.array([c for c, _ in order], np.int32)
        st.source_slot_order = np.array([sl for _, sl in order], np.uint8)
    if p["schedule"]:              # Simulation.schedule(): per station ascending, ties in call order (stable sort)
        st.sched_off, st.sched_time_ns, st.sched_rank = sched_arrays(n, p["schedule"], per_station=spec["mode"] != "single")
    if spec["mode"] == "single":
        mode = N.MODE_SINGLE
        seed = spec["seed"]
    else:
        mode = N.MODE_REPLICAS
        seed = 0
        st.seed = np.array([spec["seed"] + i for i in range(n)], np.uint64)
        st.stream_base = np.zeros(n, np.uint64)
    eng = StationEngine(st, mode=mode, horizon_ns=horizon_ns or p["end_ns"], seed=seed, log_capacity=log_capacity)
    if flags:
        eng.set_debug_flags(flags)
    return eng, p


def oracle_per_chain(spec, runs):
    """Flatten oracle runs into per-chain arrays comparable with StationEngine.lp_stats()."""
    n = spec["n_chains"]
    out = {k: np.zeros(n, np.int64) for k in
           ("generated", "accepted", "dropped", "completed", "rejected", "sink_received", "queue_depth", "active")}
    out["total_service_s"] = np.zeros(n, np.float64)
    sinks = {}
    for chain_ids, nodes, r in runs:
        for c in chain_ids:
            src, srv, snk = nodes[c]
            out["generated"][c] = r.generated[src] if src >= 0 else 0
            out["accepted"][c] = r.accepted[srv]
            out["dropped"][c] = r.dropped[srv]
            out["completed"][c] = r.completed[srv]
            out["rejected"][c] = r.rejected[srv]
            out["queue_depth"][c] = r.depth[srv]
            out["active"][c] = r.active[srv]
            out["total_service_s"][c] = r.total_service_s[srv]
            if snk >= 0:
                out["sink_received"][c] = r.received[snk]
                sinks[c] = r.sinks[snk]
    return out, sinks


def _router_pattern(spec, n):
    """spec["rt_pattern"][i] = the router's target list ('s' = the station's Sink, 'l' = its link) -> the hs_network fields."""
    pats = spec.get("rt_pattern")
    if not pats:
        return {}
    rt = np.full((4, n), -1, np.int32)
    for i, pat in enumerate(pats):
        rt[:len(pat), i] = [(-1 if ch == "s" else i) for ch in pat]
    return dict(router_target0=rt[0], router_target1=rt[1], router_target2=rt[2], router_target3=rt[3],
                router_n_targets=np.array([len(p) for p in pats], np.uint8))


def ring_arrays(spec, bag_capacity=0, log_capacity=0):
    """(StationArrays, NetworkArrays, log capacity, params) of a ring spec -- network-wide description."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import NetworkArrays, StationArrays

    p = ring_params(spec)
    n = p["n"]
    rates = np.array(p["ext_rate"], np.float64)
    st = StationArrays(
        n=n,
        src_kind=np.where(rates > 0, N.SRC_POISSON, N.SRC_NONE).astype(np.uint8),
        src_rate=np.where(rates > 0, rates, 1.0),
        src_stop_after_ns=np.full(n, -1, np.int64),
        concurrency=np.full(n, p["conc"], np.int32),
        svc_kind=np.full(n, N.LAT_EXPONENTIAL, np.uint8),
        svc_mean_s=np.full(n, p["mean"], np.float64),
        queue_cap=np.full(n, p["qcap"], np.int64),
        egress=np.full(n, N.EGRESS_SINK, np.uint8),
    )
    _probe_arrays(st, p, n)
    if spec.get("more_sources"):                         # several Sources per Server: slots in `sources=[...]` order
        order, slot_plan = ring_source_plan(spec)
        st.src_more_kind = np.full((3, n), N.SRC_NONE, np.uint8)
        st.src_more_rate = np.ones((3, n), np.float64)
        for i in range(n):
            for slot, (xa, xr, _) in enumerate(slot_plan[i]):
                k = N.SRC_POISSON if xa == O.ARR_POISSON else N.SRC_CONSTANT
                if slot == 0:
                    st.src_kind[i], st.src_rate[i] = k, xr
                else:
                    st.src_more_kind[slot - 1, i], st.src_more_rate[slot - 1, i] = k, xr
        st.source_order = np.array([i for i, _ in order], np.int32)
        st.source_slot_order = np.array([sl for _, sl in order], np.uint8)
        rates = rates + np.array([sum(x[1] for x in slot_plan[i][1:]) if len(slot_plan[i]) > 1 else 0.0 for i in range(n)]) \
            + st.src_rate * (rates <= 0) * (st.src_kind != N.SRC_NONE)
    if any(pr is not None for pr in p["profile"]):
        st.src_profile_kind = np.zeros(n, np.uint8)
        st.src_profile_params = np.zeros((n, 4), np.float64)
        for i, pr in enumerate(p["profile"]):
            if pr is not None:
                st.src_profile_kind[i] = N.PROF_LINEAR_RAMP if pr[0] == "ramp" else N.PROF_SPIKE
                st.src_profile_params[i, :len(pr) - 1] = pr[1:]
                st.src_rate[i] = max(pr[2], pr[3]) if pr[0] == "ramp" else max(pr[1], pr[2])   # peak: sizes the logs
    if p["schedule"]:
        st.sched_off, st.sched_time_ns, st.sched_rank = sched_arrays(n, p["schedule"])
    net = NetworkArrays(
        egress_kind=np.full(n, N.EGRESS_ROUTER, np.uint8),
        **{**dict(router_target0=np.full(n, -1, np.int32),   # targets=[sink_i, link_i] unless spec["rt_pattern"] says otherwise
                  router_target1=np.arange(n, dtype=np.int32)), **_router_pattern(spec, n)},
        link_of=np.full(n, -1, np.int32),
        link_src=np.arange(n, dtype=np.int32),
        link_dst=((np.arange(n) + 1) % n).astype(np.int32),
        link_lat_min_s=np.full(n, p["lat_min"], np.float64),
        link_jitter_kind=np.array([N.LAT_EXPONENTIAL if k == "exp" else N.LAT_CONSTANT for k, _ in p["jitter"]], np.uint8),
        link_jitter_mean_s=np.array([m for _, m in p["jitter"]], np.float64),
        link_loss_rate=np.array(p["loss"], np.float64) if any(p["loss"]) else None,
        bag_capacity=bag_capacity,
    )
    # external rate 4/s + forwarded 4/s per station: size the logs for the total admission rate
    horizon_s = p["end_ns"] / 1e9
    lam = 2.0 * float(max(rates.max(), (st.src_rate * (st.src_kind != N.SRC_NONE)).max())) + 1.0   # (src_rate: a profile's peak)
    cap = log_capacity or int(lam * horizon_s + 10 * (lam * horizon_s) ** 0.5 + 64)
    return st, net, cap, p


def ring_engine_for_spec(spec, flags=0, bag_capacity=0, log_capacity=0):
    """StationEngine for a ring spec: station i = Source_i -> Server_i -> RandomRouter_i([Sink_i, Link_i -> Server_{i+1}])."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    st, net, cap, p = ring_arrays(spec, bag_capacity, log_capacity)
    eng = StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=p["end_ns"], seed=spec["seed"], log_capacity=cap, network=net)
    if flags:
        eng.set_debug_flags(flags)
    return eng, p


def lb_engine_for_spec(spec, flags=0, tick_capacity=0):
    """LoadBalancerEngine for a load-balancer spec (goldens: make_golden.py run_lb_case)."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.lb_engine import LbBackendArrays, LbSourceArrays, LoadBalancerEngine

    p = lb_params(spec)
    S, B = p["S"], p["B"]
    kinds = spec.get("arr", "poisson")
    src = LbSourceArrays(
        n=S, src_rate=np.array(p["rate"], np.float64), n_clients=np.full(S, p["n_clients"], np.int64),
        src_kind=np.array([N.SRC_POISSON if k == "poisson" else N.SRC_CONSTANT for k in per_chain(kinds, S)], np.uint8),
        src_stop_after_ns=np.full(S, p["stop_ns"], np.int64))
    if spec.get("profile"):
        src.src_profile_kind = np.zeros(S, np.uint8)
        src.src_profile_params = np.zeros((S, 4), np.float64)
        for i, pr in enumerate(spec["profile"]):
            if pr is not None:
                src.src_profile_kind[i] = N.PROF_LINEAR_RAMP if pr[0] == "ramp" else N.PROF_SPIKE
                src.src_profile_params[i, :len(pr) - 1] = pr[1:]
                src.src_rate[i] = max(pr[2], pr[3]) if pr[0] == "ramp" else max(pr[1], pr[2])     # peak: sizes the tick log
    svc = spec.get("svc", "exp")
    be = LbBackendArrays(
        n=B, names=[f"srv{j}" for j in range(B)], concurrency=np.array(p["conc"], np.int32),
        svc_kind=np.array([N.LAT_EXPONENTIAL if k == "exp" else N.LAT_CONSTANT for k in per_chain(svc, B)], np.uint8),
        svc_mean_s=np.array(p["mean"], np.float64), queue_cap=np.array(p["qcap"], np.int64),
        egress=np.full(B, N.EGRESS_SINK, np.uint8))
    eng = LoadBalancerEngine(src, be, virtual_nodes=p["vnodes"], horizon_ns=p["end_ns"], shared_sink=p["shared_sink"],
                             seed=spec["seed"], tick_capacity=tick_capacity,
                             strategy={"chash": N.LB_CONSISTENT_HASH, "round_robin": N.LB_ROUND_ROBIN, "random": N.LB_RANDOM}[p["strategy"]])
    if flags:
        eng.set_debug_flags(flags)
    if spec.get("probes"):
        eng.set_probes([{"server": 0, "sink": 1, "source": 2}[who] for who, *_ in spec["probes"]], [i for _, i, *_ in spec["probes"]],
                       [PROBE_METRICS[m][1] for _, _, m, _ in spec["probes"]], [float(iv) for *_, iv in spec["probes"]])
    return eng, p


def oracle_lb_graph_ext(spec):
    """oracle_lb_graph + optional constant-rate sources / constant service (tie storms; oracle-only cases)."""
    g, p = oracle_lb_graph(spec)
    S, B = p["S"], p["B"]
    for i, k in enumerate(per_chain(spec.get("arr", "poisson"), S)):
        g.arr_kind[i] = O.ARR_POISSON if k == "poisson" else O.ARR_CONSTANT
    for j, k in enumerate(per_chain(spec.get("svc", "exp"), B)):
        g.lat_kind[S + 1 + j] = O.LAT_EXP if k == "exp" else O.LAT_CONST
    return g, p


def compare_lb_engine_with_oracle(eng, p, r, check_sink_order=True):
    """Engine (after run) vs an oracle Result of the same load-balancer topology: everything, bit-exact."""
    S, B = p["S"], p["B"]
    s = eng.summary()
    st = eng.stats()
    np.testing.assert_array_equal(s.events_by_kind, r.events_by_kind)
    assert s.events_processed == r.events_processed
    assert s.final_time_ns == r.final_time_ns
    lb = r.lbs[S]
    np.testing.assert_array_equal(st["lb"], lb["stats"])
    np.testing.assert_array_equal(st["total_requests"], lb["total_requests"])
    np.testing.assert_array_equal(st["generated"], r.generated[:S])
    be = slice(S + 1, S + 1 + B)
    for k, ok in (("accepted", r.accepted), ("dropped", r.dropped), ("completed", r.completed),
                  ("rejected", r.rejected), ("queue_depth", r.depth), ("active", r.active),
                  ("total_service_s", r.total_service_s)):
        np.testing.assert_array_equal(st[k], ok[be], err_msg=k)
    sinks = sorted(r.sinks)
    if p["shared_sink"]:
        t, cr = eng.read_sink(0)
        ot, ocr = r.sinks[sinks[0]]
        assert len(t) == len(ot) == s.sink_records
        np.testing.assert_array_equal(t, ot)
        if check_sink_order:
            np.testing.assert_array_equal(cr, ocr)
        else:
            # same records; completions of DIFFERENT backends that share both their completion ns and their service-start
            # ns may swap (the one cross-LP tie the engine does not reconstruct, DESIGN.md "Known deviation")
            np.testing.assert_array_equal(cr[np.lexsort((cr, t))], ocr[np.lexsort((ocr, ot))])
    else:
        np.testing.assert_array_equal(st["sink_received"], [r.received[i] for i in sinks])
        for j, i in enumerate(sinks):
            t, cr = eng.read_sink(j)
            np.testing.assert_array_equal(t, r.sinks[i][0])
            np.testing.assert_array_equal(cr, r.sinks[i][1])


# ---- pipelines of partitions (tests/golden/make_golden.py run_parallel_linked_case) -----------------------------------------------
def pipeline_oracle_graph(spec):
    """Oracle nodes of the `seq_network` form of a parallel_linked_* golden: lane j flows Server(stage 0) -> NetworkLink -> Server
    (stage 1) -> ... -> Sink_j.  Stream base = station index (stage-major, lane-minor); a link draws from its sender's LINK stream.
    Returns (graph, srv[stage][lane], lnk[stage][lane], snk[lane], src[lane])."""
    lanes, stages = spec["lanes"], spec["stages"]
    g = O.Graph()
    rate = per_chain(spec["rate"], lanes)
    src = [g.source(O.ARR_POISSON, float(rate[j]), stream_base=j) for j in range(lanes)]
    srv, lnk = [], []
    for k, stg in enumerate(stages):
        mean = per_chain(stg["mean"], lanes)
        srv.append([g.server(O.LAT_EXP if stg["svc"] == "exp" else O.LAT_CONST, float(mean[j]), concurrency=stg.get("concurrency", 1),
                             queue_cap=-1 if stg.get("queue_cap") is None else int(stg["queue_cap"]), stream_base=k * lanes + j)
                    for j in range(lanes)])
    snk = [g.sink() for _ in range(lanes)]
    ploss = spec.get("packet_loss") or [0.0] * (len(stages) - 1)       # PartitionLink.packet_loss of the pair the hop crosses
    for k in range(len(stages) - 1):
        lnk.append([g.link(spec["hop_latency"], spec.get("hop_jitter"), stream_base=k * lanes + j, ploss=ploss[k]) for j in range(lanes)])
    for j in range(lanes):
        g.target[src[j]] = srv[0][j]
        for k in range(len(stages)):
            g.target[srv[k][j]] = lnk[k][j] if k < len(stages) - 1 else snk[j]
            if k < len(stages) - 1:
                g.target[lnk[k][j]] = srv[k + 1][j]
    return g, srv, lnk, snk, src


def float_sum(vals, compensated):
    """`sum(list_of_floats)` of CPython (Python/bltinmodule.c builtin_sum): left-to-right additions before 3.12, Neumaier's
    compensated sum from 3.12 on (the reference requires >= 3.13).  None = what THIS interpreter's own `sum` does -- the mode
    happy_simulator_amd/_native.py sets on the library (hs_set_float_sum_mode)."""
    if compensated is None:
        return sum(vals)
    total, c = 0.0, 0.0
    for x in vals:
        if not compensated:
            total += x
            continue
        t = total + x
        c += (total - t) + x if abs(total) >= abs(x) else (x - t) + total
        total = t
    if compensated and c and math.isfinite(c):
        total += c
    return total
