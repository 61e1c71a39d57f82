"""GPU: the HIP engine (through the C ABI) against the live-reference goldens and the C oracle.

Bit-exact everywhere: event totals, per-kind histogram, final time, per-LP statistics (including the
fp64 running sum `total_service_time`), and every Sink record in nanoseconds.
"""
import numpy as np
import pytest

import helpers as H
from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu


def _engine_runnable(spec):
    # the engine draws from Philox streams; stock-MT goldens are only comparable when nothing is drawn
    if spec["rng"] == "philox":
        return True
    n = spec["n_chains"]
    return all(a == "constant" for a in H.per_chain(spec["arr"], n)) and all(
        s == "const" for s in H.per_chain(spec["svc"], n))


def _compare_engine_to_oracle(spec, eng, p, runs, check_kinds=True):
    s = eng.summary()
    stats = eng.lp_stats()
    want, sinks = H.oracle_per_chain(spec, runs)
    assert s.events_processed == sum(r.events_processed for _, _, r in runs)
    if check_kinds:
        kinds = sum(r.events_by_kind for _, _, r in runs)
        np.testing.assert_array_equal(s.events_by_kind, kinds)
        assert not s.events_by_kind[8:13].any()        # no link / router / load-balancer events without those entities
    for k, v in want.items():
        np.testing.assert_array_equal(stats[k], v, err_msg=k)
    if spec["mode"] == "single":
        assert s.final_time_ns == runs[0][2].final_time_ns
    else:
        np.testing.assert_array_equal(stats["final_time_ns"], [r.final_time_ns for _, _, r in runs])
        np.testing.assert_array_equal(stats["events"], [r.events_processed for _, _, r in runs])
    counts, t, cr = eng.read_sinks()
    off = 0
    for c in range(spec["n_chains"]):
        if c in sinks:
            ot, ocr = sinks[c]
            assert counts[c] == len(ot)
            np.testing.assert_array_equal(t[off:off + counts[c]], ot, err_msg=f"sink t chain {c}")
            np.testing.assert_array_equal(cr[off:off + counts[c]], ocr, err_msg=f"sink created chain {c}")
        off += counts[c]


def test_device_streams_match_oracle():
    from happy_simulator_amd.engine import debug_draws

    for seed, sid, k0, rate in [(42, 0, 0, 8.0), (42, (5 << 3) | 1, 0, 10.0), (2**63 + 12345, (2**40 << 3) | 2, 10**12 + 1, 3.3)]:
        n = 20000
        u, e, ns = debug_draws(seed, sid, k0, n, rate)
        for i in range(0, n, 7):
            uo = O.uniform(seed, sid, k0 + i)
            assert u[i] == uo
            eo = -O.log(1.0 - uo)
            assert e[i] == eo
            assert ns[i] == int((eo / rate) * 1e9)


def test_constant_divisor_quotients_are_ieee():
    """The engine divides by per-LP constants with multiply + FMA (hs_device.hpp ConstDiv / seconds_from_ns); every
    quotient must be bit-identical to the IEEE division the reference's arithmetic performs."""
    from happy_simulator_amd.engine import debug_const_div

    rng = np.random.default_rng(7)
    n = 4_000_000
    ns = rng.integers(0, 2**40, n).astype(np.float64)
    qf, qi, qn = debug_const_div(ns, 1e9)
    np.testing.assert_array_equal(qf, ns / 1e9)
    np.testing.assert_array_equal(qi, ns / 1e9)
    np.testing.assert_array_equal(qn, ns / 1e9)
    e = -np.log1p(-rng.random(n))                      # the range of E = -log(1 - u)
    tiny = np.ldexp(rng.random(n) + 0.5, rng.integers(-60, 8, n))
    for b in (8.0, 10.0, 1.0 / 0.1, 3.3, 0.1, 7.0, 1.0 / 0.013, 12345.678, float(np.nextafter(2.0, 0.0))):
        for a in (e, tiny):
            qf, qi, _ = debug_const_div(a, b)
            np.testing.assert_array_equal(qf, a / b, err_msg=f"b={b!r}")
            np.testing.assert_array_equal(qi, a / b)


@pytest.mark.parametrize("name", H.golden_names())
def test_engine_matches_reference_golden(name):
    gold = H.Golden(name)
    spec = gold.spec
    if not _engine_runnable(spec):
        pytest.skip("stock MT19937 streams are sequential by construction (oracle-only golden)")
    eng, p = H.engine_for_spec(spec)
    with eng:
        for w in spec.get("windows", ()):           # a golden of the reference driven window by window: so is the engine
            eng.run_until(H.ns_from_seconds(w))
        eng.run_until(p["end_ns"])
        s = eng.summary()
        stats = eng.lp_stats()
        assert s.events_processed == sum(gold.meta["total_events"])
        if spec["mode"] == "single":
            assert s.final_time_ns == gold.meta["final_ns"][0]
        else:
            np.testing.assert_array_equal(stats["final_time_ns"], gold.meta["final_ns"])
            np.testing.assert_array_equal(stats["events"], gold.meta["total_events"])
        if "trace" in gold.arrays:
            np.testing.assert_array_equal(s.events_by_kind, np.bincount(gold.trace[:, 1], minlength=len(s.events_by_kind)))
        if "probe_t_ns" in gold.arrays:          # Probe samples: what the reference appended to each probe's Data
            for c in range(spec["n_chains"]):
                for j in range(gold.n_probe_slots):
                    gt, gv = gold.probe_samples(c, j)
                    pt, pv = eng.read_probe(c, j)
                    np.testing.assert_array_equal(pt, gt, err_msg=f"probe times chain {c} slot {j}")
                    np.testing.assert_array_equal(pv, gv, err_msg=f"probe values chain {c} slot {j}")
        shared = bool(spec.get("shared_sink"))
        for k, g in (("generated", "generated"), ("accepted", "accepted"), ("dropped", "dropped"),
                     ("completed", "completed"), ("rejected", "rejected"), ("sink_received", "received"),
                     ("queue_depth", "depth"), ("active", "active"), ("total_service_s", "total_service_s")):
            if shared and k == "sink_received":      # one Sink behind every server: the golden reports it under chain 0
                assert stats[k].sum() == gold.arrays[g][0]
                continue
            np.testing.assert_array_equal(stats[k], gold.arrays[g], err_msg=k)
        if "generated_more" in gold.arrays:          # several Sources per Server: every Source's own generated_count
            for j in range(3):
                np.testing.assert_array_equal(eng.source_generated(1 + j), gold.generated_more[j], err_msg=f"source slot {1 + j}")
        counts, t, cr = eng.read_sinks()
        if shared:                                   # per-station logs -> the shared Sink's lists (device merge by time)
            from happy_simulator_amd import _native as N
            t, cr = np.ascontiguousarray(t), np.ascontiguousarray(cr)
            assert N.lib().hs_merge_sink_records(0, len(t), t.ctypes.data, cr.ctypes.data) == 0
        np.testing.assert_array_equal(t, gold.sink_t_ns)
        # Sink latency rule (components/common.py:39-40): (t - created_at).to_seconds()
        np.testing.assert_array_equal((t - cr).astype(np.float64) / 1e9, gold.sink_latency_s)


@pytest.mark.parametrize("name", [n for n in H.golden_names()])
def test_general_path_equals_fast_path(name):
    spec = H.Golden(name).spec
    if not _engine_runnable(spec):
        pytest.skip("oracle-only golden")
    res = []
    for flags in (0, 1):
        eng, p = H.engine_for_spec(spec, flags=flags)
        with eng:
            eng.run_until(p["end_ns"])
            s = eng.summary()
            res.append((s.events_processed, tuple(s.events_by_kind), s.final_time_ns,
                        {k: v.tobytes() for k, v in eng.lp_stats().items()}, [a.tobytes() for a in eng.read_sinks()]))
    assert res[0] == res[1]


SWEEP = [
    dict(name="sweep_replicas", n_chains=1024, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=60.0,
         rng="philox", seed=4242, mode="replicas"),
    dict(name="sweep_single", n_chains=1024, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=30.0,
         rng="philox", seed=77, mode="single"),
    dict(name="sweep_c4", n_chains=300, arr="poisson", rate=35.0, svc="exp", mean=0.1, concurrency=4, end_s=20.0,
         rng="philox", seed=5, mode="single"),
    dict(name="sweep_c3_cap2", n_chains=300, arr="poisson", rate=40.0, svc="exp", mean=0.1, concurrency=3, queue_cap=2,
         end_s=20.0, rng="philox", seed=6, mode="replicas"),
    dict(name="sweep_overload_cap", n_chains=257, arr="poisson", rate=14.0, svc="exp", mean=0.1, queue_cap=5, end_s=30.0,
         rng="philox", seed=9, mode="single"),
    dict(name="sweep_const_ties", n_chains=64, arr="constant", rate=[10.0, 20.0, 5.0, 40.0] * 16, svc="const",
         mean=[0.1, 0.05, 0.2, 0.025] * 16, concurrency=[1, 1, 2, 1] * 16, queue_cap=[None, 3, None, 1] * 16,
         end_s=12.0, rng="philox", seed=1, mode="single"),
    dict(name="sweep_const_exp_mix", n_chains=128, arr=["constant", "poisson"] * 64, rate=10.0, svc=["exp", "const"] * 64,
         mean=0.09, end_s=25.0, rng="philox", seed=3, mode="replicas"),
    dict(name="sweep_zero_service", n_chains=16, arr="constant", rate=100.0, svc="const", mean=0.0, end_s=3.0,
         rng="philox", seed=3, mode="single"),
]


@pytest.mark.parametrize("spec", SWEEP, ids=[s["name"] for s in SWEEP])
def test_engine_matches_oracle(spec):
    runs = H.run_oracle_for_spec(spec)
    eng, p = H.engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        _compare_engine_to_oracle(spec, eng, p, runs)


@pytest.mark.parametrize("mode", ["single", "replicas"])
def test_reentrant_windows_match_oracle(mode):
    spec = dict(name="win", n_chains=96, arr="poisson", rate=9.0, svc="exp", mean=0.1, concurrency=1, end_s=12.0,
                rng="philox", seed=31, mode=mode)
    windows = [H.ns_from_seconds(x) for x in (0.5, 1.0, 1.0, 3.25, 7.0)]
    # oracle: _execute_until per window on the same heap(s) (parallel/coordinator.py uses _run_window this way)
    n = spec["n_chains"]
    p = H.spec_chain_params(spec)
    runs = []
    groups = [(list(range(n)), spec["seed"], list(range(n)))] if mode == "single" else [
        ([i], spec["seed"] + i, [0]) for i in range(n)]
    for chain_ids, seed, bases in groups:
        g, nodes = H.oracle_graph_for(spec, chain_ids, bases)
        runs.append((chain_ids, nodes, O.run(g, p["end_ns"], seed=seed, windows=windows)))
    eng, p = H.engine_for_spec(spec)
    with eng:
        for w in windows:
            eng.run_until(w)
        eng.run_until(p["end_ns"])
        _compare_engine_to_oracle(spec, eng, p, runs)


@pytest.mark.parametrize("mode", ["single", "replicas"])
def test_scheduled_requests_match_oracle(mode):
    """Simulation.schedule(): 160 stations (Poisson / constant / no source, c = 1..3, bounded queues), 3 000 Requests
    injected before run() at random nanoseconds plus bursts at one timestamp and some beyond the horizon; whole run and
    re-entrant windows against the oracle: every count, statistic and Sink record."""
    rng = np.random.default_rng(17)
    n = 160
    rate = [0.0 if i % 4 == 0 else float(rng.uniform(3, 12)) for i in range(n)]
    sched = [[int(rng.integers(0, n)), float(rng.integers(1, 14_000_000_000)) / 1e9 + 1e-9] for _ in range(3000)]
    sched += [[5, 3.000000007]] * 6 + [[8, 2.500000001]] * 9 + [[12, 15.5], [12, 14.25], [16, 15.0]]
    spec = dict(name="sched", n_chains=n, arr=["poisson", "constant"] * (n // 2), rate=rate, svc="exp", mean=0.07,
                concurrency=[1, 2, 1, 3] * (n // 4), queue_cap=[None, None, 4, None, 2] * (n // 5), schedule=sched,
                end_s=13.0, rng="philox", seed=19, mode=mode)
    runs = H.run_oracle_for_spec(spec)
    eng, p = H.engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        _compare_engine_to_oracle(spec, eng, p, runs)
    if mode == "single":
        windows = [H.ns_from_seconds(x) for x in (0.5, 2.5, 3.0, 3.000000007, 9.0)]
        g, nodes = H.oracle_graph_for(spec, list(range(n)), list(range(n)))
        r = O.run(g, p["end_ns"], seed=spec["seed"], windows=windows,
                  schedule=[(nodes[c][1], t) for c, t in p["schedule"]])
        eng, p = H.engine_for_spec(spec)
        with eng:
            for w in windows:
                eng.run_until(w)
            eng.run_until(p["end_ns"])
            _compare_engine_to_oracle(spec, eng, p, [(list(range(n)), nodes, r)])


def test_profile_sweep_matches_oracle():
    """192 chains with random LinearRamp / Spike profiles (Poisson and deterministic arrivals, zero start rates, kinks and
    discontinuities inside the horizon) against the oracle: every count, statistic and Sink record."""
    rng = np.random.default_rng(5)
    n = 192
    prof, arr = [], []
    for i in range(n):
        if i % 3 == 0:
            prof.append(["ramp", float(rng.uniform(2, 12)), float(rng.choice([0.0, 3.0, 25.0])), float(rng.uniform(4, 40))])
        elif i % 3 == 1:
            prof.append(["spike", float(rng.uniform(2, 12)), float(rng.uniform(40, 120)), float(rng.uniform(1, 6)),
                         float(rng.uniform(0.5, 4))])
        else:
            prof.append(None)
        arr.append("poisson" if i % 2 else "constant")
    spec = dict(name="profile_sweep", n_chains=n, arr=arr, rate=8.0, svc="exp", mean=0.04, concurrency=[1, 2] * (n // 2),
                profile=prof, end_s=10.0, rng="philox", seed=77, mode="single")
    runs = H.run_oracle_for_spec(spec)
    want, sinks = H.oracle_per_chain(spec, runs)
    eng, p = H.engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary()
        st = eng.lp_stats()
        counts, t, cr = eng.read_sinks()
    assert s.events_processed == sum(r.events_processed for _, _, r in runs)
    assert s.final_time_ns == runs[0][2].final_time_ns
    for k in want:
        np.testing.assert_array_equal(st[k], want[k], err_msg=k)
    off = 0
    for c in range(n):
        ot, ocr = sinks[c]
        np.testing.assert_array_equal(t[off:off + counts[c]], ot)
        np.testing.assert_array_equal(cr[off:off + counts[c]], ocr)
        off += counts[c]


def test_nonzero_start_time_matches_oracle():
    """`Simulation(start_time=Instant.from_seconds(5), ...)`: sources draw their first arrival from start_time
    (load/source.py:120-140); chains, a ring (both network engines) and a load-balancer topology against the oracle."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine
    from happy_simulator_amd.lb_engine import LbBackendArrays, LbSourceArrays, LoadBalancerEngine

    start, end = 5_000_000_000, 13_000_000_000
    # chains
    n = 96
    g = O.mm1_chains(n, rate=8.0, mean=0.1)
    r = O.run(g, end, start_ns=start, seed=9)
    with StationEngine(StationArrays.uniform(n, rate=8.0, mean=0.1), mode=N.MODE_SINGLE, horizon_ns=end, start_ns=start,
                       seed=9) as eng:
        eng.run_until(end)
        s = eng.summary()
        assert (s.events_processed, s.final_time_ns) == (r.events_processed, r.final_time_ns)
        np.testing.assert_array_equal(s.events_by_kind, r.events_by_kind)
        counts, t, cr = eng.read_sinks()
        np.testing.assert_array_equal(t, np.concatenate([r.sinks[i][0] for i in sorted(r.sinks)]))
        assert t.min() > start
    # ring, both engines
    spec = dict(name="ring_start", topology="ring", n=48, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=13.0,
                seed=10)
    gr, nodes = H.oracle_ring_graph(spec)
    rr = O.run(gr, end, start_ns=start, seed=10)
    st, net, cap, _ = H.ring_arrays(spec)
    for flags in (0, 16):
        with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end, start_ns=start, seed=10, log_capacity=cap, network=net) as eng:
            if flags:
                eng.set_debug_flags(flags)
            eng.run_until(end)
            s = eng.summary()
            assert (s.events_processed, s.final_time_ns) == (rr.events_processed, rr.final_time_ns), flags
            np.testing.assert_array_equal(s.events_by_kind, rr.events_by_kind)
    # load balancer
    lspec = dict(n_sources=24, n_backends=40, rate=20.0, mean=0.1, vnodes=100, n_clients=5000, end_s=13.0, seed=11)
    gl, p = H.oracle_lb_graph(lspec)
    rl = O.run(gl, end, start_ns=start, seed=11)
    src = LbSourceArrays(n=24, src_rate=np.full(24, 20.0), n_clients=np.full(24, 5000, np.int64))
    be = LbBackendArrays(n=40, names=[f"srv{j}" for j in range(40)], svc_kind=np.full(40, N.LAT_EXPONENTIAL, np.uint8),
                         svc_mean_s=np.full(40, 0.1))
    with LoadBalancerEngine(src, be, virtual_nodes=100, horizon_ns=end, start_ns=start, seed=11) as eng:
        eng.run(end)
        H.compare_lb_engine_with_oracle(eng, p, rl)


def test_profile_inversion_budget_reports_the_lp_instead_of_stalling():
    """N3: one arrival of LinearRampProfile(3 s, 1 -> 9) on stream base 97, seed 77 needs 8e7 rate evaluations in the
    reference's own adaptive-Simpson inversion (70 s of the reference's Python).  Round 2 refused it
    (a lone lane, 2^20 intervals); the tick-table kernel (csrc/hs_tables.hpp) evaluates the integral with 64 lanes and the run
    is EXACT (tests/test_gpu_tables.py compares the whole network with the oracle).  The evaluation budget is a run-time
    argument now: with a small one the run is refused with the LP's index instead of guessed."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    n = 130
    st = StationArrays.uniform(n, rate=8.0, mean=0.1)
    st.src_profile_kind = np.zeros(n, np.uint8)
    st.src_profile_params = np.zeros((n, 4), np.float64)
    st.src_profile_kind[97] = N.PROF_LINEAR_RAMP
    st.src_profile_params[97, :3] = (3.0, 1.0, 9.0)
    st.src_rate[97] = 9.0
    import time
    t0 = time.perf_counter()
    with pytest.raises(N.EngineError, match="LP 97.*adaptive-Simpson"):
        with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=5_000_000_000, seed=77) as eng:
            eng.set_profile_budget(1 << 10)
            eng.run_until(5_000_000_000)
    assert time.perf_counter() - t0 < 30.0
    t0 = time.perf_counter()
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=5_000_000_000, seed=77) as eng:   # the default budget: it runs
        eng.run_until(5_000_000_000)
        assert eng.lp_stats()["generated"][97] > 5
    assert time.perf_counter() - t0 < 30.0
    st.src_profile_params[97, :3] = (5.0, 3.0, 20.0)             # an ordinary ramp on the same stream
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=5_000_000_000, seed=77) as eng:
        eng.run_until(5_000_000_000)
        assert eng.lp_stats()["generated"][97] > 20
