"""GPU: BASELINE-size runs (65 536 station LPs, 60 s) checked through size-independent properties and
through exact oracle parity on a sample of LPs (LPs are independent, so an LP's results do not depend on
how many other LPs share the heap -- SURVEY.md A4)."""
import hashlib

import numpy as np
import pytest

from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu

N_LP = 65536
END_NS = 60_000_000_000


def _engine(mode, seed=42, per_lp_seed=False):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    st = StationArrays.uniform(N_LP, rate=8.0, mean=0.1)
    return StationEngine(st, mode=N.MODE_SINGLE if mode == "single" else N.MODE_REPLICAS, horizon_ns=END_NS, seed=seed)


def _digest(eng):
    h = hashlib.sha256()
    s = eng.summary()
    h.update(np.array([s.events_processed, s.final_time_ns, s.requests_completed, s.sink_records]).tobytes())
    h.update(s.events_by_kind.tobytes())
    for k, v in sorted(eng.lp_stats().items()):
        h.update(v.tobytes())
    for a in eng.read_sinks():
        h.update(a.tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def single_run():
    eng = _engine("single")
    eng.run_until(END_NS)
    yield eng
    eng.close()


def test_event_accounting_identities(single_run):
    eng = single_run
    s = eng.summary()
    st = eng.lp_stats()
    k = s.events_by_kind
    assert s.events_processed == k.sum() == st["events"].sum()
    assert k[0] == st["generated"].sum()                       # SourceEvent == ticks
    assert k[1] == st["accepted"].sum() + st["dropped"].sum()  # every payload is an enqueue attempt
    assert k[4] == k[5]                                        # each QUEUE_DELIVER re-emits its payload
    assert k[6] == st["completed"].sum() == s.requests_completed
    assert st["rejected"].sum() == 0
    # conservation: accepted = completed + waiting + in service
    np.testing.assert_array_equal(st["accepted"], st["completed"] + st["queue_depth"] + st["active"])
    # the overshoot event (one for the whole Simulation) may complete a request whose Sink event is never processed
    assert 0 <= k[6] - k[7] <= 1 and k[7] == st["sink_received"].sum() == s.sink_records
    # polls = completions + notifies that found a free server; a delivery needs a poll
    assert k[3] >= k[6] and k[3] <= k[6] + k[2] and k[4] <= k[3]
    # about 7.5 reference events per request at rho = 0.8
    assert 7.3 < s.events_processed / s.requests_completed < 7.7
    assert s.final_time_ns > END_NS and s.final_time_ns - END_NS < 1_000_000   # the overshoot is the globally first event


def test_sink_records_are_ordered_and_causal(single_run):
    counts, t, cr = single_run.read_sinks()
    off = np.concatenate([[0], np.cumsum(counts)])
    assert (t >= cr).all() and (cr > 0).all() and (t <= END_NS).all()
    d = np.diff(t)
    dc = np.diff(cr)
    boundary = np.zeros(len(t) - 1, bool)
    boundary[off[1:-1][off[1:-1] < len(t)] - 1] = True
    assert (d[~boundary] >= 0).all()    # completion order within an LP
    assert (dc[~boundary] >= 0).all()   # FIFO: created_at is non-decreasing too


def test_runs_are_deterministic(single_run):
    a = _digest(single_run)
    single_run.reset()
    single_run.run_until(END_NS)
    assert _digest(single_run) == a
    eng2 = _engine("single")
    with eng2:
        eng2.run_until(END_NS)
        assert _digest(eng2) == a


def test_replicas_vs_single_overshoot_relation_and_sampled_oracle_parity(single_run):
    s1 = single_run.summary()
    rep = _engine("replicas")
    with rep:
        rep.run_until(END_NS)
        s2 = rep.summary()
        st = rep.lp_stats()
        # N Simulations process N overshoot events, one Simulation processes 1
        assert s2.events_processed - s1.events_processed == N_LP - 1
        rng = np.random.default_rng(5)
        sample = sorted(set([0, 1, 255, 256, 4095, N_LP - 1] + list(rng.integers(0, N_LP, 42))))
        for lp in sample:
            g = O.mm1_chains(1, rate=8.0, mean=0.1, stream_base0=int(lp))
            r = O.run(g, END_NS, seed=42)
            assert st["events"][lp] == r.events_processed
            assert st["final_time_ns"][lp] == r.final_time_ns
            assert st["generated"][lp] == r.generated[0]
            assert st["completed"][lp] == r.completed[1]
            assert st["total_service_s"][lp] == r.total_service_s[1]
            t, cr = rep.read_sink(int(lp))
            np.testing.assert_array_equal(t, r.sinks[2][0])
            np.testing.assert_array_equal(cr, r.sinks[2][1])


def test_baseline_config1_4096_replicas_seed_matched():
    """BASELINE configs[1] verbatim: 4 096 independent M/M/1 replicas (Poisson 8/s, Exp mean 0.1 s), 60 s, replica i seeded
    base_seed + i as ParallelRunner.run_replicas does (parallel/runner.py:115-142) -- every replica's event count, final
    time, statistics and Sink records against 4 096 separate runs of the oracle."""
    import helpers as H

    spec = dict(name="config1", n_chains=4096, arr="poisson", rate=8.0, svc="exp", mean=0.1, end_s=60.0, rng="philox",
                seed=20260923, mode="replicas")
    runs = H.run_oracle_for_spec(spec)
    want, sinks = H.oracle_per_chain(spec, runs)
    eng, p = H.engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary()
        st = eng.lp_stats()
        counts, t, cr = eng.read_sinks()
    assert s.events_processed == sum(r.events_processed for _, _, r in runs) > 14_000_000
    np.testing.assert_array_equal(st["events"], [r.events_processed for _, _, r in runs])
    np.testing.assert_array_equal(st["final_time_ns"], [r.final_time_ns for _, _, r in runs])
    for k in want:
        np.testing.assert_array_equal(st[k], want[k], err_msg=k)
    np.testing.assert_array_equal(t, np.concatenate([sinks[c][0] for c in range(4096)]))
    np.testing.assert_array_equal(cr, np.concatenate([sinks[c][1] for c in range(4096)]))


def test_specialised_uniform_kind_grid_kernel_equals_the_generic_one():
    """An engine whose LPs are all Source.poisson -> Server(Exp, c = 1, unbounded) -> Sink runs hs_station_run<1, false, true, true>
    (compile-time entity kinds in the request-order loop, csrc/hs_station.hpp HSG); debug flag 1 << 20 keeps the generic
    instantiation: every statistic and every Sink record identical at the headline size."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    end_ns = 60_000_000_000
    got = []
    for flags in (0, 1 << 20):
        eng = StationEngine(StationArrays.uniform(65536), mode=N.MODE_SINGLE, horizon_ns=end_ns, seed=42)
        with eng:
            if flags:
                eng.set_debug_flags(flags)
            eng.run_until(end_ns)
            s = eng.summary()
            got.append((s.events_processed, s.final_time_ns, s.events_by_kind.tolist(), [a.tobytes() for a in eng.read_sinks()],
                        {k: v.tobytes() for k, v in eng.lp_stats().items()}))
    assert got[0] == got[1] and got[0][0] == 237150263
