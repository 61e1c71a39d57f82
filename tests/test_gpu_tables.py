"""GPU: the tick-table kernel (csrc/hs_tables.hpp) -- the arrival times of time-varying Sources and the tick times of Probes,
produced before the run by 64 lanes per integral -- against (a) the same chain walked by ONE lane with the sequential
integrator of csrc/hs_profile.hpp (the restatement of load/arrival_time_provider.py:84-144 that rounds 1 - 2 pinned with
live-reference goldens), bit for bit, and (b) the oracle, on the two configurations round 2 had to refuse."""
import os

import numpy as np
import pytest

import helpers as H
import random_specs as RS
from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu

RAMP, SPIKE, GENERAL_CONSTANT = 1, 2, 3


def _table(profile, poisson, seed, sid, horizon_s, cap, lone, budget=0, start_ns=0):
    from happy_simulator_amd import _native as N

    L = N.lib()
    prof = np.asarray(profile, np.float64)
    out = np.zeros(cap, np.int64)
    status = np.zeros(2, np.uint64)
    n = L.hs_debug_tick_table(0, prof.ctypes.data, int(poisson), seed, sid, start_ns, int(horizon_s * 1e9), cap, budget,
                              int(lone), out.ctypes.data, status.ctypes.data)
    assert n >= 0, L.hs_last_global_error()
    return out[:min(n, cap)].copy(), status


CASES = [
    ("ramp 5 s 3->20, Poisson", (RAMP, 5.0, 3.0, 20.0, 0.0), 1, 4.0),
    ("ramp 5 s 3->20, deterministic", (RAMP, 5.0, 3.0, 20.0, 0.0), 0, 4.0),
    ("ramp 2 s 2->30", (RAMP, 2.0, 2.0, 30.0, 0.0), 1, 3.0),
    ("ramp down to a low rate", (RAMP, 2.0, 25.0, 2.0, 0.0), 1, 4.0),
    ("spike 3 / 40 at 1 s for 0.5 s", (SPIKE, 3.0, 40.0, 1.0, 0.5), 1, 3.0),
    ("spike, deterministic", (SPIKE, 3.0, 40.0, 1.0, 0.5), 0, 3.0),
    ("probe every 0.1 s", (GENERAL_CONSTANT, 10.0, 0.0, 0.0, 0.0), 0, 6.0),
    ("probe every 0.37 s", (GENERAL_CONSTANT, 1.0 / 0.37, 0.0, 0.0, 0.0), 0, 20.0),
    ("probe every 1 ms", (GENERAL_CONSTANT, 1000.0, 0.0, 0.0, 0.0), 0, 0.4),
]


# Two comparisons let ONE lane finish an integral of 10^7 Simpson intervals (that is what they are for): 223 s and 91 s of a
# 9-minute suite.  They run with HS_SLOW_TESTS=1 (last: profiles/r03_gpu_tests_part2.log, both passed); the other eight profile
# cases and the budget / overflow tests always run.
SLOW = bool(os.environ.get("HS_SLOW_TESTS"))
needs_minutes = pytest.mark.skipif(not SLOW, reason="minutes of one lane by design; HS_SLOW_TESTS=1 runs it (passed in profiles/r03_gpu_tests_part2.log)")


@pytest.mark.parametrize("name,profile,poisson,horizon_s",
                         [pytest.param(*c, marks=needs_minutes) if c[0] == "ramp 2 s 2->30" else c for c in CASES], ids=[c[0] for c in CASES])
def test_cooperative_tables_equal_the_lone_lane_chain(name, profile, poisson, horizon_s):
    for seed, sid in ((42, 8 * 3), (77, 8 * 97), (5, 8 * 1234567)):
        coop, st_c = _table(profile, poisson, seed, sid, horizon_s, 4096, lone=0)
        lone, st_l = _table(profile, poisson, seed, sid, horizon_s, 4096, lone=1, budget=1 << 27)   # (the lone lane's own limit)
        assert st_c[0] == 0 and st_l[0] == 0 and st_c[1] == 0, (name, seed, st_c, st_l)
        np.testing.assert_array_equal(coop, lone, err_msg=f"{name} seed {seed}")
        assert len(coop) >= 3 and (coop[-1] > horizon_s * 1e9 or coop[-1] == np.iinfo(np.int64).max)
        if not poisson:
            break


def _oracle_ticks(profile, poisson, seed, sid, horizon_s):
    """The same stream on the ORACLE's event loop (hs_oracle.c: ArrivalTimeProvider.next_arrival_time's general path, adaptive
    Simpson + bracket search + Brent, load/arrival_time_provider.py:84-144): a Source with the profile in front of a Sink -- the
    Sink's record times are the arrival times; a Probe's sample times are its ticks."""
    g = O.Graph()
    if profile[0] == GENERAL_CONSTANT:
        snk = g.sink()
        node = g.probe(snk, 5, 1.0 / profile[1])
        assert g.prof_p[node][0] == profile[1]
    else:
        prof = ("ramp",) + tuple(profile[1:4]) if profile[0] == RAMP else ("spike",) + tuple(profile[1:5])
        src = g.source(O.ARR_POISSON if poisson else O.ARR_CONSTANT, 1.0, stream_base=sid >> 3, profile=prof)
        node = g.sink()
        g.target[src] = node
    r = O.run(g, int(horizon_s * 1e9), seed=seed)
    return r.sinks[node][0]


@pytest.mark.parametrize("name,profile,poisson,horizon_s", CASES, ids=[c[0] for c in CASES])
def test_cooperative_tables_equal_the_oracle(name, profile, poisson, horizon_s):
    """... and against the ORACLE (the CPU restatement the live reference pins: profile_* / probe goldens, live random cases), not
    only against the device's own lone-lane chain: every tick up to the horizon and the one beyond it, bit for bit -- incl. the
    `ramp 2 s 2->30` case whose lone-lane comparison takes minutes."""
    for seed, sid in ((42, 8 * 3), (77, 8 * 97), (5, 8 * 1234567)):
        coop, st = _table(profile, poisson, seed, sid, horizon_s, 4096, lone=0)
        assert st[0] == 0 and st[1] == 0, (name, seed, st)
        want = _oracle_ticks(profile, poisson, seed, sid, horizon_s)
        # (the oracle's run ends with the first event beyond the end -- the tick's SourceEvent --, so the Sink holds every tick up
        #  to the end; the table goes on: its next entry is that tick beyond the end)
        assert len(want) >= 2 and len(coop) > len(want) and coop[len(want)] > horizon_s * 1e9 >= want[-1]
        np.testing.assert_array_equal(coop[:len(want)], want, err_msg=f"{name} seed {seed}")
        if not poisson:
            break


def test_the_explosive_integral_equals_the_oracle():
    """LinearRampProfile(3 s, 1 -> 9), stream base 97, seed 77 (2.65e7 Simpson intervals for the first arrival): the cooperative
    table == the oracle's chain (the lone-lane comparison below needs minutes of one lane)."""
    prof = (RAMP, 3.0, 1.0, 9.0, 0.0)
    coop, st = _table(prof, 1, 77, 8 * 97, 2.0, 64, lone=0)
    assert st[0] == 0
    want = _oracle_ticks(prof, 1, 77, 8 * 97, 2.0)
    assert len(want) >= 1 and len(coop) >= len(want)
    np.testing.assert_array_equal(coop[:len(want)], want)


@needs_minutes
def test_an_explosive_integral_is_split_over_the_lanes_bit_for_bit():
    """LinearRampProfile(3 s, 1 -> 9), stream base 97, seed 77 -- the first arrival's bracket search
    integrates over [0, 9.12 s] and needs 2.65e7 Simpson intervals (7.95e7 rate evaluations).  The lone lane is allowed to
    finish here (budget 2^26): the cooperative result must be the same 64 bits."""
    prof = (RAMP, 3.0, 1.0, 9.0, 0.0)
    coop, st_c = _table(prof, 1, 77, 8 * 97, 2.0, 64, lone=0)
    assert st_c[0] == 0
    lone, st_l = _table(prof, 1, 77, 8 * 97, 2.0, 64, lone=1, budget=1 << 26)
    assert st_l[0] == 0
    np.testing.assert_array_equal(coop, lone)


def test_the_budget_is_a_runtime_argument_and_names_the_stream():
    prof = (RAMP, 3.0, 1.0, 9.0, 0.0)
    _, st = _table(prof, 1, 77, 8 * 97, 2.0, 64, lone=0, budget=1 << 8)
    assert st[0] == 2          # 2 + owner (0)
    _, st = _table(prof, 1, 77, 8 * 97, 2.0, 4, lone=0)
    assert st[0] == 0


def test_a_table_that_is_too_small_is_reported():
    _, st = _table((GENERAL_CONSTANT, 1000.0, 0.0, 0.0, 0.0), 0, 1, 0, 1.0, 100, lone=0)
    assert st[1] == 2


def test_station_spec_1011_is_exact_under_the_default_build():
    """tools/gpu_random_sweep.py, round 2: random_specs.station_spec(1011) -- an arrival that spans the end of a ramp towards a
    low rate, 2^22 intervals -- was refused (or took 22 s with a library built with HS_PROF_BUDGET_LOG2=24)."""
    from test_gpu_random import check_station_case

    check_station_case(1011)


def test_a_ring_with_the_pathological_ramp_equals_the_oracle():
    """130 stations, LinearRampProfile(3 s, 1 -> 9) on station 97 (the configuration that seemed to
    hang in round 1 and was refused in round 2): both network engines against the oracle."""
    from test_gpu_ring import _check_against_oracle

    n = 130
    prof = [None] * n
    prof[97] = ["ramp", 3.0, 1.0, 9.0]
    spec = dict(name="ring_130_ramp97", topology="ring", n=n, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, profile=prof,
                end_s=2.0, seed=77)
    g, nodes = H.oracle_ring_graph(spec)
    p = H.ring_params(spec)
    r = O.run(g, p["end_ns"], seed=spec["seed"], schedule=[(nodes[c]["srv"], t) for c, t in p["schedule"]])
    for flags in (0, 16):
        eng, p = H.ring_engine_for_spec(spec, flags=flags)
        with eng:
            eng.run_until(p["end_ns"])
            _check_against_oracle(spec, eng, r, nodes)
