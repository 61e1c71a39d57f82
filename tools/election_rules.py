#!/usr/bin/env python3
"""CPU analysis of the one known deviation: which LP processes the ONE event beyond end_time.

The reference pops the pending event with the smallest (time, _sort_index).  The station engine elects among the LPs' first
pending events by (time, creation time, construction rank) -- `cand_less` / `cand_rank` in csrc.  This tool evaluates such keys
WITHOUT a GPU: an analysis build of the oracle (gcc -DHSO_LINEAGE oracle/hs_oracle.c: every event also carries when it was
created, when its creator was, and that one's creator) dumps everything that is pending when the reference pops its event
beyond end_time; each LP's candidate is its pending event with the smallest sort index at that time (the order inside an LP is
exact on the engine), and a key is right when it elects the LP of the reference's event.

    python tools/election_rules.py --first 0 --count 1000          # tests/random_specs.py tie_spec(k)

The key the engine uses must fail on exactly the cases the GPU fails on (k = 85, 134, 279, 978 below 1 000; 2079, 2150 in
2000..2352: profiles/r02_gpu_random_sweep_after.log) -- that is the check of this emulation.  The other keys are candidates for
closing the deviation: the same key extended by the creator's creation time, and by the creator's creator's.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
import random_specs as RS  # noqa: E402
from oracle import hs_oracle as O  # noqa: E402

EV_SOURCE, EV_ENQUEUE, EV_CONTINUATION, EV_PROBE_TICK = 0, 1, 6, 13

ENGINE_KEY = "(created, depth, root created, rank; Probes by their own list position)"
KEYS = {
    "engine before the GPU sweep: (created, LP's first-listed Source)": lambda c: (c["crt"], c["old_rank"]),
    "engine: (created, rank)": lambda c: (c["crt"], c["rank"]),
    "(created, creator created, rank)": lambda c: (c["crt"], c["crt2"], c["rank"]),
    "(created, creator created, its creator created, rank)": lambda c: (c["crt"], c["crt2"], c["crt3"], c["rank"]),
    "(created, rank) with departures after ticks": lambda c: (c["crt"], c["kind"] != EV_SOURCE, c["rank"]),
    "(created, creator created, ticks first, rank)": lambda c: (c["crt"], c["crt2"], c["kind"] != EV_SOURCE, c["rank"]),
    # the heap is a FIFO among the events of one nanosecond: a group runs breadth-first from its roots (the events that were
    # pending from earlier), so of two events created in one nanosecond the one FEWER steps from its root came first, then the
    # one whose root did -- and the roots compare the same way (when created, how deep in that group, ...)
    "(created, depth, rank)": lambda c: (c["crt"], c["cdepth"], c["rank"]),
    "(created, depth, root created, rank)": lambda c: (c["crt"], c["cdepth"], c["rcrt"], c["rank"]),
    "(created, depth, root created, rank; Probes by their own list position)": lambda c: (c["crt"], c["cdepth"], c["rcrt"], c["rank2"]),
    "(created, depth, root created, rank of the lineage's Source)": lambda c: (c["crt"], c["cdepth"], c["rcrt"], c.get("rank3", c.get("rank2", c["rank"]))),
    "(created, depth, root created, root depth, rank)": lambda c: (c["crt"], c["cdepth"], c["rcrt"], c["rcdepth"], c["rank"]),
    "(created, depth, root created, root depth, its root created, its depth, rank)":
        lambda c: (c["crt"], c["cdepth"], c["rcrt"], c["rcdepth"], c["r2crt"], c["r2cdepth"], c["rank"]),
}


def lineage_lib(tmp):
    path = os.path.join(tmp, "libhs_oracle_lineage.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-DHSO_LINEAGE", "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "oracle", "hs_oracle.c"), "-o", path, "-lm"])
    return path


def pending_at_overshoot(spec, run=None):
    """Run the spec on the lineage oracle; returns (rows [n][12], what `run` returned) with row 0 = the reference's event
    beyond end_time."""
    L = O.lib()
    got = {}
    real_destroy = L.hso_destroy

    def destroy(h):                                 # O.run destroys its handle: read the dump first
        buf = np.zeros((4096, 13), np.int64)
        n = L.hso_read_dump(h, buf.ctypes.data, 4096)
        got["rows"] = buf[:min(n, 4096)].copy()
        real_destroy(h)

    L.hso_destroy = destroy
    try:
        runs = (run or H.run_oracle_for_spec)(spec)
    finally:
        L.hso_destroy = real_destroy
    return got.get("rows"), runs


def candidates(spec, rows, runs):
    (chain_ids, nodes, r), = runs
    n = spec["n_chains"]
    where = {}                                       # node -> (LP, what)
    for c in chain_ids:
        src, srv, snk = nodes[c]
        if src >= 0:
            where[src] = (c, ("src", 0))
        where[srv] = (c, "srv")
    for (c, slot), nd in r.xsrc_nodes.items():       # further Sources of a Server: slot = position among ITS Sources
        where[nd] = (c, ("src", slot))
    probe_pos = {}
    for q, ((c, _slot), nd) in enumerate(sorted(r.probe_nodes_all.items())):   # `probes=[...]` lists them chain by chain
        where[nd] = (c, "probe")
        probe_pos[nd] = q
    # csrc/hs_station.hpp cand_rank: a tick ranks by its own Source's position in `sources=[...]`, anything else by the LP's
    # first-listed Source (sourceless LPs behind them), a Probe's tick behind all of that
    order = H.source_plan(spec, list(chain_ids))[0] if spec.get("more_sources") else [(c, 0) for c in chain_ids if nodes[c][0] >= 0]
    src_rank = {cs: q for q, cs in enumerate(order)}
    lp_rank, lp_first = {}, {}
    for q, (c, _sl) in enumerate(order):
        lp_rank.setdefault(c, q)
        lp_first.setdefault(c, len(lp_first))        # (round 2 before the sweep: dense first-appearance rank, for every kind)
    for c in chain_ids:
        lp_rank.setdefault(c, len(order) + c)
        lp_first.setdefault(c, len(lp_first))
    t_star = rows[0, 0]
    best = {}
    for t, idx, kind, node, crt, crt2, crt3, cdepth, rcrt, rcdepth, r2crt, r2cdepth, rsrc in rows:
        if t != t_star or node not in where:
            continue
        lp, what = where[node]
        if lp not in best or idx < best[lp]["idx"]:
            best[lp] = dict(lp=lp, idx=int(idx), kind=int(kind), crt=int(crt), crt2=int(crt2), crt3=int(crt3),
                            cdepth=int(cdepth), rcrt=int(rcrt), rcdepth=int(rcdepth), r2crt=int(r2crt), r2cdepth=int(r2cdepth),
                            rank=(src_rank[(lp, what[1])] if isinstance(what, tuple) else
                                  lp_rank[lp] + (len(order) + n if what == "probe" else 0)),
                            old_rank=lp_first[lp] + (n if what == "probe" else 0))
            # next: a Probe's tick by the Probe's own position in `probes=[...]` (behind every Source and sourceless LP)
            best[lp]["rank2"] = best[lp]["rank"] if what != "probe" else 2 * len(order) + 2 * n + probe_pos[node]
            # round 4: the rank of anything but a tick stands in for the rank of the Source its lineage goes back to -- exact with
            # one Source per Server, a guess with several (csrc/hs_engine.hip set_stations: such ties go to the single heap)
            best[lp]["amb"] = (not isinstance(what, tuple)) and what != "probe" and sum(1 for (c2, _s) in order if c2 == lp) > 1
            # candidate for round 5: anything but a tick ranks by the Source whose tick is the most recent one in its ancestry
            # (hso_event::rsrc, analysis build) -- what a device-side `root source` carried with every pending departure would give
            rs = where.get(int(rsrc))
            best[lp]["rank3"] = (best[lp]["rank2"] if (isinstance(what, tuple) or what == "probe" or rs is None or not isinstance(rs[1], tuple))
                                 else src_rank[(rs[0], rs[1][1])])
    ref_lp = where[rows[0, 3]][0] if rows[0, 3] in where else None
    return ref_lp, list(best.values())


def run_ring(spec):
    g, nodes = H.oracle_ring_graph(spec)
    p = H.ring_params(spec)
    O.run(g, p["end_ns"], seed=spec["seed"], schedule=[(nodes[c]["srv"], t) for c, t in p["schedule"]])
    return nodes


def ring_candidates(spec, rows, nodes):
    """Stations of a ring (tests/helpers.py oracle_ring_graph): a message in transit belongs to the station it is sent to (the
    network engines keep it in that LP's bag).  The order INSIDE a station -- a message against a local event -- is taken as
    exact here, so this counts the election only."""
    n = spec["n"]
    where, probe_pos = {}, {}
    for i in range(n):
        for key, nd in nodes[i].items():
            if nd < 0:
                continue
            if key.startswith("src"):
                where[nd] = (i, ("src", int(key[3:] or 0)))
            elif key == "srv":
                where[nd] = (i, "srv")
            elif key == "lnk":
                where[nd] = ((i + 1) % n, "msg")
            elif key.startswith("prb"):
                where[nd] = (i, "probe")
                probe_pos[nd] = len(probe_pos)
    order = H.ring_source_plan(spec)[0]
    src_rank = {cs: q for q, cs in enumerate(order)}
    lp_rank = {}
    for q, (i, _sl) in enumerate(order):
        lp_rank.setdefault(i, q)
    for i in range(n):
        lp_rank.setdefault(i, len(order) + i)
    t_star = rows[0, 0]
    best = {}
    for t, idx, kind, node, crt, crt2, crt3, cdepth, rcrt, rcdepth, r2crt, r2cdepth, rsrc in rows:
        if t != t_star or node not in where:
            continue
        lp, what = where[node]
        if lp not in best or idx < best[lp]["idx"]:
            rank = src_rank[(lp, what[1])] if isinstance(what, tuple) else lp_rank[lp] + (len(order) + n if what == "probe" else 0)
            best[lp] = dict(lp=lp, idx=int(idx), kind=int(kind), crt=int(crt), crt2=int(crt2), crt3=int(crt3), cdepth=int(cdepth),
                            rcrt=int(rcrt), rcdepth=int(rcdepth), r2crt=int(r2crt), r2cdepth=int(r2cdepth), rank=rank,
                            old_rank=lp,                                   # (the network engines before the sweep: the LP index)
                            rank2=rank if what != "probe" else 2 * len(order) + 2 * n + probe_pos[node])
    ref_lp = where[rows[0, 3]][0] if rows[0, 3] in where else None
    return ref_lp, list(best.values())


def run_tandem(spec):
    import tandem_specs as TS

    g, srcs, servers, sinks = TS.oracle_graph(spec)
    O.run(g, int(spec["end_s"] * 1e9), seed=spec["seed"])
    return srcs, servers


def tandem_candidates(spec, rows, nodes):
    """Tandem queues (tests/tandem_specs.py): one LP per Server, the chain's Source on its first Server's LP; the rank of an LP
    = its chain's position in `sources=[...]`, upstream Servers first (csrc/hs_engine.hip tie_rank of a tandem engine)."""
    import tandem_specs as TS

    srcs, servers = nodes
    order, first = TS.station_index(spec)
    where, rank = {}, {}
    for c, nd in enumerate(srcs):
        where[nd] = first[c]
        rank[nd] = c * 8
    for (c, st), nd in servers.items():
        where[nd] = first[c] + st
        rank[nd] = c * 8 + st
    t_star = rows[0, 0]
    best = {}
    for t, idx, kind, node, crt, crt2, crt3, cdepth, rcrt, rcdepth, r2crt, r2cdepth, rsrc in rows:
        if t != t_star or node not in where:
            continue
        lp = where[node]
        if lp not in best or idx < best[lp]["idx"]:
            best[lp] = dict(lp=lp, idx=int(idx), kind=int(kind), crt=int(crt), crt2=int(crt2), crt3=int(crt3), cdepth=int(cdepth),
                            rcrt=int(rcrt), rcdepth=int(rcdepth), r2crt=int(r2crt), r2cdepth=int(r2cdepth), rank=rank[node],
                            old_rank=lp, rank2=rank[node])
    ref_lp = where.get(rows[0, 3])
    return ref_lp, list(best.values())


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=1000)
    ap.add_argument("--verbose", action="store_true", help="print the candidates of every case a key gets wrong")
    ap.add_argument("--family", choices=("tie", "multi_source", "multi_source_ring", "tandem"), default="tie",
                    help="tests/random_specs.py tie_spec, multi_source_spec (several Sources per Server, two list orders) or "
                         "multi_source_ring_spec (the same on rings: the network engines' election)")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        O._LIB_PATH = lineage_lib(tmp)
        O.build = lambda force=False: O._LIB_PATH
        L = O.lib()
        L.hso_read_dump.restype = C.c_int64
        L.hso_read_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        wrong = {name: [] for name in KEYS}
        silent = {name: 0 for name in KEYS}     # ... of which between two Probes' ticks: the tick beyond end_time records nothing
        ties = skipped = 0
        fb = {"fallbacks": 0, "missed": [], "saved": []}
        for k in range(a.first, a.first + a.count):
            if a.family == "multi_source_ring":
                spec = RS.multi_source_ring_spec(k)
                rows, nodes = pending_at_overshoot(spec, run_ring)
            elif a.family == "tandem":
                import tandem_specs as TS

                spec = TS.tandem_spec(k)
                rows, nodes = pending_at_overshoot(spec, run_tandem)
            else:
                spec = RS.tie_spec(k) if a.family == "tie" else RS.multi_source_spec(k)
                if spec["mode"] != "single":
                    skipped += 1                     # replicas: every LP is its own Simulation, no election
                    continue
                spec["trace"] = False
                spec.pop("shared_sink", None)        # (as tests/test_gpu_random.py runs them on the station engine)
                rows, runs = pending_at_overshoot(spec)
            if rows is None or len(rows) == 0:
                skipped += 1                         # nothing beyond end_time
                continue
            ref_lp, cands = (ring_candidates(spec, rows, nodes) if a.family == "multi_source_ring" else
                             tandem_candidates(spec, rows, nodes) if a.family == "tandem" else candidates(spec, rows, runs))
            if ref_lp is None:
                skipped += 1
                continue
            if len({(c["crt"]) for c in cands}) < len(cands):
                ties += 1                            # at least two LPs' candidates were created on one nanosecond
            if a.family in ("tie", "multi_source"):      # the engine's key + its fallback (round 4)
                ek = lambda c: (c["crt"], c["cdepth"], c["rcrt"])
                w = min(cands, key=KEYS[ENGINE_KEY])
                fell_back = any(c is not w and ek(c) == ek(w) and (c.get("amb") or w.get("amb")) for c in cands)
                fb["fallbacks"] += fell_back
                if w["lp"] != ref_lp and not fell_back:
                    fb["missed"].append(k)
                if w["lp"] != ref_lp and fell_back:
                    fb["saved"].append(k)
            for name, key in KEYS.items():
                win = min(cands, key=key)["lp"]
                if win != ref_lp:
                    wrong[name].append(k)
                    by_lp = {c["lp"]: c for c in cands}
                    silent[name] += by_lp[win]["kind"] == EV_PROBE_TICK and by_lp[ref_lp]["kind"] == EV_PROBE_TICK
                    if a.verbose:
                        print(f"  case {k}: '{name}' elects LP {win}, the reference LP {ref_lp}: "
                              + "; ".join(f"LP{c['lp']} kind {c['kind']} idx {c['idx']} created {c['crt']} depth {c['cdepth']} root {c['rcrt']}/{c['rcdepth']} <- {c['r2crt']}/{c['r2cdepth']}"
                                          for c in sorted(cands, key=lambda c: c["idx"])))
        print(f"{a.family}_spec({a.first}..{a.first + a.count - 1}): {a.count - skipped} runs with an event beyond end_time, "
              f"{ties} where two LPs' candidates share their creation nanosecond")
        for name, ks in wrong.items():
            print(f"  {name:82s} wrong on {len(ks):3d} ({silent[name]} between two Probes' ticks: nothing recorded differs): {ks[:20]}")
        if a.family in ("tie", "multi_source"):
            print(f"  engine key + the round-4 rule (a tie on every key but the rank, one of the two a non-tick of an LP with several Sources -> "
                  f"single heap): {fb['fallbacks']} runs fall back, {len(fb['saved'])} of them would have been wrong {fb['saved'][:10]}, "
                  f"wrong and NOT caught: {len(fb['missed'])} {fb['missed'][:20]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
