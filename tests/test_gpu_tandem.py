"""GPU: tandem queues -- `Server(..., downstream=<another Server>)` (components/server/server.py:64-122,271-272; the forwarded Event
keeps its context, core/entity.py:83-105) -- engine == oracle on everything the ABI reports: totals, the per-kind histogram, the
time of the one event beyond end_time, every Server's statistics (binary64 total_service_time included), every Sink record.
The engine runs a chain in passes, upstream Servers first (csrc/hs_station.hpp "tandem queues"); the oracle is the reference's
single heap (oracle/hs_oracle.c on_continuation: forward(event, downstream)).  The oracle's tandem path itself is pinned by the
live-reference golden `tandem_*` (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import tandem_specs as TS
from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu


def _run_case(spec, windows=(), flags=0):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    g, srcs, servers, sinks = TS.oracle_graph(spec)
    end = int(spec["end_s"] * 1e9)
    r = O.run(g, end, seed=spec["seed"], windows=[int(w * 1e9) for w in windows])
    with StationEngine(TS.engine_arrays(spec), mode=N.MODE_SINGLE, horizon_ns=end, seed=spec["seed"]) as eng:
        if flags:
            eng.set_debug_flags(flags)
        for w in windows:
            eng.run_until(int(w * 1e9))
        eng.run_until(end)
        TS.compare(spec, eng, r, srcs, servers, sinks)
    return r


def test_two_servers_in_a_row():
    spec = dict(chains=[dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="exp", mean=0.1, conc=1, qcap=None), dict(svc="exp", mean=0.08, conc=1, qcap=None)])],
                end_s=20.0, seed=42)
    r = _run_case(spec)
    assert r.events_processed > 1500


def test_lock_step_constants_share_every_nanosecond():
    """Arrivals every 100 ms, both Servers take exactly 100 ms: from the second arrival on, the tick, the first Server's
    completion, its forward and the second Server's completion all happen on the same nanosecond."""
    for means in ((0.1, 0.1), (0.1, 0.1, 0.1, 0.1), (0.05, 0.1), (0.0, 0.1), (0.1, 0.0, 0.05), (0.2, 0.1)):
        spec = dict(chains=[dict(arr="constant", rate=10.0, stop_after_s=None, sink=True,
                                 stages=[dict(svc="const", mean=m, conc=1, qcap=None) for m in means])],
                    end_s=3.0, seed=7)
        _run_case(spec)
        spec["chains"][0]["stages"][-1]["conc"] = 2
        spec["chains"].append(dict(arr="constant", rate=10.0, stop_after_s=1.0, sink=False,
                                   stages=[dict(svc="const", mean=m, conc=2, qcap=1) for m in means]))
        _run_case(spec)


@pytest.mark.parametrize("first", range(0, 200, 50))
def test_random_tandems_match_the_oracle(first):
    bad = []
    for k in range(first, first + 50):
        try:
            _run_case(TS.tandem_spec(k))
        except AssertionError as e:
            bad.append((k, str(e).strip().splitlines()[:3]))
    assert not bad, bad


def test_several_windows_continue_the_chain():
    """run_until twice: the event the first window elected beyond its end (a completion whose forward is still pending, a
    forward's enqueue, ...) is part of the state the second window starts from."""
    for k in (1, 4, 8, 12, 17, 20, 33):
        spec = TS.tandem_spec(k)
        _run_case(spec, windows=(0.37 * spec["end_s"], 0.81 * spec["end_s"]))


def test_what_is_not_lowered_is_refused_by_name():
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationEngine

    spec = dict(chains=[dict(arr="poisson", rate=8.0, stop_after_s=None, sink=True,
                             stages=[dict(svc="exp", mean=0.1, conc=1, qcap=None)] * 2)] * 2, end_s=1.0, seed=1)
    st = TS.engine_arrays(spec)
    st.downstream_lp[2] = 1                      # two Servers forward to station 1
    with pytest.raises(N.EngineError, match="one upstream Server per Server"):
        StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=10**9)
    st = TS.engine_arrays(spec)
    with pytest.raises(N.EngineError, match="HS_MODE_SINGLE"):
        StationEngine(st, mode=N.MODE_REPLICAS, horizon_ns=10**9)
    st = TS.engine_arrays(spec)
    st.probe_metric = np.full(st.n, N.PROBE_NONE, np.uint8)
    st.probe_metric[1] = 0
    st.probe_interval_s = np.full(st.n, 0.1)
    with pytest.raises(N.EngineError, match="not lowered yet"):
        StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=10**9)
