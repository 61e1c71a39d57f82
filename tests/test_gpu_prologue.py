"""The prologue is skipped where it cannot matter (include/hs_engine.h hs_engine_prologue_path, csrc/hs_engine.hip `lazy_prologue`).

The reference numbers the events constructed before run() first and restarts the count for run-time events
(core/simulation.py:77,145-160; core/event.py:62-77): only a pre-run event that shares its nanosecond with another event of its
LP can be overtaken by a run-time event.  The single-heap prologue (csrc/hs_exact.hpp) replays that -- on ONE lane for the whole
engine.  A station engine runs without it, lets the kernels report such coincidences, and repeats the run behind the prologue
only then.  These tests pin: the skipped path gives what the prologue path gives, coincidences and too-short runs are detected,
and the speed difference is what the skipping is for."""
import time

import numpy as np
import pytest

from happy_simulator_amd import _native as N

pytestmark = pytest.mark.gpu

FORCE_PROLOGUE = 1 << 16


def _grid(n, *, constant=False, probe_interval=1.0, rate=8.0):
    from happy_simulator_amd.engine import StationArrays

    st = StationArrays.uniform(n, src_kind=N.SRC_CONSTANT if constant else N.SRC_POISSON, rate=rate, mean=0.1)
    st.probe_metric = np.zeros(n, np.uint8)                       # depth
    st.probe_interval_s = np.full(n, probe_interval)
    return st


def _everything(eng, n_probe_lps):
    s = eng.summary()
    stats = eng.lp_stats()
    counts, t, cr = eng.read_sinks()
    probes = [eng.read_probe(lp) for lp in range(n_probe_lps)]
    return s, stats, counts, t, cr, probes


def _assert_same(a, b):
    sa, sta, ca, ta, cra, pa = a
    sb, stb, cb, tb, crb, pb = b
    assert sa.events_processed == sb.events_processed
    np.testing.assert_array_equal(sa.events_by_kind, sb.events_by_kind)
    assert sa.final_time_ns == sb.final_time_ns
    for k in sta:
        np.testing.assert_array_equal(sta[k], stb[k], err_msg=k)
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(ta, tb)
    np.testing.assert_array_equal(cra, crb)
    for (t1, v1), (t2, v2) in zip(pa, pb):
        np.testing.assert_array_equal(t1, t2)
        np.testing.assert_array_equal(v1, v2)


def test_poisson_chains_with_probes_skip_the_prologue_and_match_it():
    from happy_simulator_amd.engine import StationEngine

    n, end = 2048, 20_000_000_000
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=end, seed=7) as eng:
        eng.run_until(end)
        assert eng.prologue_path() == 1
        lazy = _everything(eng, 64)
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=end, seed=7) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        assert eng.prologue_path() == 2
        eager = _everything(eng, 64)
    _assert_same(lazy, eager)


def test_windows_skip_the_prologue_too():
    from happy_simulator_amd.engine import StationEngine

    n, end = 512, 6_000_000_000
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=end, seed=11) as eng:
        for e in (2_000_000_000, 2_000_000_001, 4_500_000_000, end):
            eng.run_until(e)
        assert eng.prologue_path() == 1
        lazy = _everything(eng, 32)
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=end, seed=11) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        eager = _everything(eng, 32)
    _assert_same(lazy, eager)


def test_constant_arrivals_with_aligned_probes_match_the_prologue_whichever_path_they_take():
    from happy_simulator_amd.engine import StationEngine

    n, end = 256, 5_000_000_000
    with StationEngine(_grid(n, constant=True, rate=10.0), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.run_until(end)
        print("constant 10/s + 1 s Probe: prologue path", eng.prologue_path())
        lazy = _everything(eng, 16)
    with StationEngine(_grid(n, constant=True, rate=10.0), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        eager = _everything(eng, 16)
    _assert_same(lazy, eager)


def _sched_grid(n, times_ns):
    from happy_simulator_amd.engine import StationArrays

    st = StationArrays.uniform(n, src_kind=N.SRC_CONSTANT, rate=10.0, mean=0.03)
    k = len(times_ns)
    st.sched_off = np.arange(n + 1, dtype=np.int64) * k
    st.sched_time_ns = np.tile(np.asarray(times_ns, np.int64), n)
    return st


@pytest.mark.parametrize("times", [(300_000_000,), (450_000_000, 450_000_000)])
def test_an_injected_request_on_the_nanosecond_of_another_event_goes_through_the_prologue(times):
    """Constant 10/s arrivals and a Request injected at 0.3 s (the third arrival's nanosecond) / two Requests injected at one instant."""
    from happy_simulator_amd.engine import StationEngine

    n, end = 64, 3_000_000_000
    with StationEngine(_sched_grid(n, times), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.run_until(end)
        assert eng.prologue_path() == 2
        s, stats, counts, t, cr, _ = _everything(eng, 0)
    with StationEngine(_sched_grid(n, times), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        s2, stats2, counts2, t2, cr2, _ = _everything(eng, 0)
    _assert_same((s, stats, counts, t, cr, []), (s2, stats2, counts2, t2, cr2, []))


def test_injected_requests_away_from_other_events_skip_the_prologue():
    from happy_simulator_amd.engine import StationEngine

    n, end = 64, 3_000_000_000
    with StationEngine(_sched_grid(n, (1_234_567_891, 2_000_000_007)), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.run_until(end)
        assert eng.prologue_path() == 1
        a = _everything(eng, 0)
    with StationEngine(_sched_grid(n, (1_234_567_891, 2_000_000_007)), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        b = _everything(eng, 0)
    _assert_same(a, b)


def test_a_run_shorter_than_the_pre_run_events_are_many_goes_through_the_prologue():
    from happy_simulator_amd.engine import StationEngine

    n, end = 4096, 50_000_000           # 50 ms: ~0.4 arrivals per chain, 8 192 pre-run events
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=10_000_000_000, seed=5) as eng:
        eng.run_until(end)
        assert eng.prologue_path() == 2
        lazy = _everything(eng, 16)
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=10_000_000_000, seed=5) as eng:
        eng.set_debug_flags(FORCE_PROLOGUE)
        eng.run_until(end)
        eager = _everything(eng, 16)
    _assert_same(lazy, eager)


def test_65536_chains_with_a_probe_each_run_in_milliseconds():
    from happy_simulator_amd.engine import StationEngine

    n, end = 65536, 60_000_000_000
    with StationEngine(_grid(n), mode=N.MODE_SINGLE, horizon_ns=end, seed=42) as eng:
        eng.run_until(end)
        eng.reset()
        t0 = time.perf_counter()
        eng.run_until(end)
        wall = time.perf_counter() - t0
        s = eng.summary()
        assert eng.prologue_path() == 1
        print(f"65 536 chains x 60 s with a Probe each: {wall * 1e3:.2f} ms wall, kernel {s.kernel_ms:.3f} ms, {s.events_processed} events")
        assert wall < 0.25                      # (behind the prologue: 2.2 s)
        stats = eng.lp_stats()
        assert s.events_by_kind[13] >= n * 59 and s.events_by_kind[14] >= n * 59   # every Probe ticked and sampled
        assert int(stats["generated"].sum()) == s.events_by_kind[0]


def test_probes_sampled_in_request_order_equal_the_oracle():
    """Chains that stay on the request-order loop with a Probe (one worker, unbounded FIFO, Poisson arrivals, one Probe per Server):
    every metric, intervals from 20 ms to 1 s, with and without a Sink -- everything the ABI reports and every sample against
    the oracle's single heap (csrc/hs_station.hpp req_probe_*)."""
    import tandem_specs as TS

    metrics = ["depth", "active_requests", "stats_accepted", "stats_dropped", "requests_completed"]
    for seed in range(6):
        rng = np.random.default_rng(31_000 + seed)
        chains, probes = [], []
        for c in range(16):
            chains.append(dict(arr="poisson", rate=float(rng.choice([4.0, 8.0, 12.0, 30.0])), stop_after_s=None, sink=bool(c % 4 != 3),
                               stages=[dict(svc="exp" if c % 5 else "const", mean=float(rng.choice([0.02, 0.05, 0.1])), conc=1, qcap=None)]))
            probes.append(((c, 0), metrics[(c + seed) % len(metrics)], float(rng.choice([0.02, 0.05, 0.1, 0.25, 0.3, 1.0]))))
        spec = dict(chains=chains, end_s=float(rng.choice([2.0, 5.0, 8.0])), seed=int(rng.integers(1, 1 << 30)), probes=probes, sched=[])
        assert TS.run_tandem_probe_case(spec) == (0, 1)


@pytest.mark.parametrize("k", [7932, 8264])
def test_two_requests_in_one_nanosecond_at_an_idle_single_worker_server(k):
    """tandem_probe_case(7932 / 8264), found by the sweep on chains WITHOUT tandem queues: an injected Request on the nanosecond of
    an arrival / two Requests injected at one instant reach an idle one-worker Server together; the reference delivers both and
    rejects the second at the worker (server.py:223-234).  The engine's single-slot shortcut "the m-th completion is the m-th
    admission" then named the wrong created_at for every later Sink record; such engines now carry the explicit column."""
    import tandem_specs as TS

    TS.run_tandem_probe_case(TS.tandem_probe_case(k))


@pytest.mark.parametrize("first", ["summary", "lp_stats", "read_sinks"])
def test_getters_after_an_asynchronous_run_return_the_final_results(first):
    """hs_engine_run_until_async + a getter, with no hs_engine_synchronize in between: the getter itself finalises the run,
    i.e. repeats it behind the prologue when a pre-run event met another event of its LP -- the same results as the blocking call,
    whichever getter comes first."""
    from happy_simulator_amd.engine import StationEngine

    n, end, times = 64, 3_000_000_000, (300_000_000,)
    with StationEngine(_sched_grid(n, times), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.run_until(end)
        want = _everything(eng, 0)
    with StationEngine(_sched_grid(n, times), mode=N.MODE_SINGLE, horizon_ns=end, seed=3) as eng:
        eng.run_until_async(end)
        assert eng.prologue_path() == 1                       # nothing has looked at the results yet
        getattr(eng, first)()
        assert eng.prologue_path() == 2                       # ... the first look repeated the run behind the prologue
        got = _everything(eng, 0)
    _assert_same(want, got)


# ---- networks (round 4): the asynchronous / windowed network engines skip the prologue too ---------------------------------------
def _probed_ring(n, end_s, *, lockstep=False, schedule=True):
    """A ring with a Probe on most stations (some with three), Requests injected with schedule() and a few extra constant Sources;
    `lockstep`: station 1 gets a constant 4 /s Source whose FIRST tick (0.25 s) falls on its Probe's first tick -- two pre-run
    events on one nanosecond of one LP."""
    probes = [None if i % 7 == 6 else [["depth", 0.25], ["active_requests", 0.1], ["stats_accepted", 0.35]] if i % 5 == 0
              else ["depth", 0.25] for i in range(n)]
    more = [[["constant", 4.0]] if (lockstep and i == 1) else [["constant", 3.0]] if i % 11 == 4 else None for i in range(n)]
    if lockstep:
        probes[1] = ["depth", 0.25]
    spec = dict(name=f"probed_ring_{n}", topology="ring", n=n, ext_rate=[0.0 if i % 13 == 12 else 5.0 for i in range(n)], mean=0.1,
                lat_min=0.001, jitter_mean=0.006, end_s=end_s, seed=29, probes=probes, more_sources=more)
    if schedule:
        spec["schedule"] = [[i, end_s * (0.11 + 0.2 * k) + 1e-9 * (i % 5)] for i in range(2, n, 17) for k in range(3)]
    return spec


def _ring_everything(eng, spec):
    s, st, ns = eng.summary(), eng.lp_stats(), eng.net_stats()
    counts, t, cr = eng.read_sinks()
    probes = []
    for i, pr in enumerate(spec["probes"]):
        for j in range(0 if pr is None else len(pr) if isinstance(pr[0], list) else 1):
            probes.append(eng.read_probe(i, j))
    return s, st, ns, counts, t, cr, probes


def _assert_same_ring(a, b):
    sa, sta, nsa, ca, ta, cra, pa = a
    sb, stb, nsb, cb, tb, crb, pb = b
    assert sa.events_processed == sb.events_processed and sa.final_time_ns == sb.final_time_ns
    np.testing.assert_array_equal(sa.events_by_kind, sb.events_by_kind)
    for k in sta:
        np.testing.assert_array_equal(sta[k], stb[k], err_msg=k)
    for k in nsa:
        np.testing.assert_array_equal(nsa[k], nsb[k], err_msg=k)
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(ta, tb)
    np.testing.assert_array_equal(cra, crb)
    assert len(pa) == len(pb) and len(pa) > 0
    for (t1, v1), (t2, v2) in zip(pa, pb):
        np.testing.assert_array_equal(t1, t2)
        np.testing.assert_array_equal(v1, v2)


@pytest.mark.parametrize("engine_flags", [0, 16], ids=["async", "windowed"])
def test_probed_rings_skip_the_prologue_and_match_it(engine_flags):
    """Probes, scheduled Requests and further Sources on a 300-station ring: no pre-run event shares its nanosecond with another
    event of its station, so the network engine runs without the single-lane prologue (path 1) -- and gives exactly what the run
    behind the prologue gives (debug flag 1 << 16), sample for sample."""
    import helpers as H

    spec = _probed_ring(300, 4.0)
    eng, p = H.ring_engine_for_spec(spec, flags=engine_flags)
    with eng:
        eng.run_until(p["end_ns"])
        assert eng.prologue_path() == 1
        lazy = _ring_everything(eng, spec)
    eng, p = H.ring_engine_for_spec(spec, flags=engine_flags | FORCE_PROLOGUE)
    with eng:
        eng.run_until(p["end_ns"])
        assert eng.prologue_path() == 2
        eager = _ring_everything(eng, spec)
    _assert_same_ring(lazy, eager)
    assert lazy[0].events_processed > 50_000


@pytest.mark.parametrize("engine_flags", [0, 16], ids=["async", "windowed"])
def test_a_ring_station_whose_probe_meets_a_pre_run_tick_repeats_the_run_behind_the_prologue(engine_flags):
    """A constant Source's first tick on the nanosecond of its station's first Probe tick: the run reports the coincidence
    (Totals::undecided bit 2, NetStation::run_group) and is repeated behind the prologue (path 2) -- the results are those of an
    engine that never skipped it, and the oracle's."""
    import helpers as H
    from oracle import hs_oracle as O

    spec = _probed_ring(40, 3.0, lockstep=True, schedule=False)
    eng, p = H.ring_engine_for_spec(spec, flags=engine_flags)
    with eng:
        eng.run_until(p["end_ns"])
        assert eng.prologue_path() == 2
        got = _ring_everything(eng, spec)
    eng, p = H.ring_engine_for_spec(spec, flags=engine_flags | FORCE_PROLOGUE)
    with eng:
        eng.run_until(p["end_ns"])
        eager = _ring_everything(eng, spec)
    _assert_same_ring(got, eager)
    g, nodes = H.oracle_ring_graph(spec)
    r = O.run(g, p["end_ns"], seed=spec["seed"])
    assert got[0].events_processed == r.events_processed and got[0].final_time_ns == r.final_time_ns
    np.testing.assert_array_equal(got[0].events_by_kind, r.events_by_kind)


@pytest.mark.parametrize("engine_flags", [0, 16], ids=["async", "windowed"])
def test_an_earlier_window_end_after_a_repeated_network_run_moves_nothing(engine_flags):
    """a network whose lazy prologue met a hazard is repeated behind the prologue (path 2); that repeat reset the
    engine and used to forget the last window end, so a REPEATED or EARLIER `run_until` ran again to the earlier end and the state
    regressed.  The reference's loop condition is already false there (core/simulation.py:527-541): nothing moves."""
    import helpers as H

    spec = _probed_ring(40, 4.5, lockstep=True, schedule=False)
    one, p = H.ring_engine_for_spec(spec, flags=engine_flags)
    end, mid = p["end_ns"], (2 * p["end_ns"]) // 3
    with one:
        one.run_until(mid)
        want_mid = _ring_everything(one, spec)
    one, _ = H.ring_engine_for_spec(spec, flags=engine_flags)
    with one:
        one.run_until(end)
        want_end = _ring_everything(one, spec)
    eng, _ = H.ring_engine_for_spec(spec, flags=engine_flags)
    with eng:
        eng.run_until(mid)
        assert eng.prologue_path() == 2
        _assert_same_ring(_ring_everything(eng, spec), want_mid)
        for e in (mid, mid // 2, mid - 1):             # repeated, earlier, just before: nothing moves
            eng.run_until(e)
            _assert_same_ring(_ring_everything(eng, spec), want_mid)
        eng.run_until(end)                             # a later end continues to what one run gives
        _assert_same_ring(_ring_everything(eng, spec), want_end)
        eng.run_until(mid)
        _assert_same_ring(_ring_everything(eng, spec), want_end)


def test_a_probe_on_every_station_of_a_large_ring_no_longer_costs_the_single_lane_prologue():
    """VERDICT r3 'one-lane cliffs': 16 384 stations with a Probe each = 32 768 pre-run events, ~17 us each on the prologue's single
    lane (0.5 s) before a run of a few milliseconds.  The skipped path must be at least 20x faster than the forced prologue, and equal."""
    import helpers as H

    n = 16384
    spec = dict(name="probed_ring_16k", topology="ring", n=n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=2.0,
                seed=31, probes=[["depth", 0.5]] * n)
    out = {}
    for name, flags in (("lazy", 0), ("eager", FORCE_PROLOGUE)):
        eng, p = H.ring_engine_for_spec(spec, flags=flags)
        with eng:
            eng.run_until(p["end_ns"])                       # (warm-up: code objects, first touch)
            eng.reset()
            t0 = time.perf_counter()
            eng.run_until(p["end_ns"])
            out[name] = (time.perf_counter() - t0, eng.prologue_path(), _ring_everything(eng, spec))
    assert out["lazy"][1] == 1 and out["eager"][1] == 2
    _assert_same_ring(out["lazy"][2], out["eager"][2])
    print(f"probed ring, {n} stations: {out['lazy'][0] * 1e3:.1f} ms without the prologue, {out['eager'][0] * 1e3:.1f} ms behind it")
    assert out["lazy"][0] * 20 < out["eager"][0]
