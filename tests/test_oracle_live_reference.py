"""CPU, build container only (needs /root/reference): the oracle against the LIVE reference on randomly drawn
configurations -- beyond the 55 committed fixtures, which were produced by exactly the same code path
(tests/golden/make_golden.py: the reference's own event loop, queue protocol, Server generator and sort-index ledger with
per-entity Philox streams plugged in through its extension points).  Every count, statistic, Sink record and, for the
small cases, the full processed-event trace incl. `_sort_index`.  Skipped where the reference is absent (the GPU box)."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from test_oracle_golden import (check_oracle_against_lb_golden, check_oracle_against_ring_golden,
                                check_oracle_against_station_golden, check_oracle_against_tandem_golden)

pytestmark = pytest.mark.live_reference

if not os.path.isdir("/root/reference/happysimulator"):
    pytest.skip("needs /root/reference (build container only)", allow_module_level=True)

sys.path.insert(0, H.GOLDEN_DIR)
import make_golden as MG  # noqa: E402  (imports the reference through refshim)
from random_specs import (lb_probe_spec, lb_profile_spec, lb_spec as _lb_spec, multi_source_ring_spec, multi_source_spec, ring_spec as _ring_spec, station_spec as _station_spec,  # noqa: E402
                          tie_spec)


@pytest.mark.parametrize("k", range(40))
def test_oracle_equals_live_reference_on_random_station_specs(k):
    spec = _station_spec(k)
    if spec["mode"] == "replicas":
        spec["trace"] = False                      # traces are recorded for one Simulation
    out, meta = MG.run_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert sum(gold.meta["total_events"]) > 0
    check_oracle_against_station_golden(gold)


@pytest.mark.parametrize("k", range(30))
def test_oracle_equals_live_reference_on_random_ring_specs(k):
    spec = _ring_spec(k)
    out, meta = MG.run_ring_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 100
    check_oracle_against_ring_golden(gold)


@pytest.mark.parametrize("k", range(20))
def test_oracle_equals_live_reference_on_random_load_balancer_specs(k):
    spec = _lb_spec(k)
    out, meta = MG.run_lb_case(spec)
    gold = H.Golden.from_results(out, meta)
    assert gold.meta["total_events"][0] > 50
    check_oracle_against_lb_golden(gold)


@pytest.mark.parametrize("k", range(40))
def test_oracle_equals_live_reference_on_tie_storms(k):
    """Same-nanosecond orders (lock-step constant sources, Requests injected at the start instant, c up to 16): the oracle's
    sort-index ledger against the reference's, full traces."""
    out, meta = MG.run_case(tie_spec(k))
    check_oracle_against_station_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(60)) + [1374])          # (1374 ...: the cases tools/gpu_random_sweep.py found, tests/test_gpu_random.py)
def test_oracle_equals_live_reference_with_several_sources_per_server(k):
    """Up to four Sources feeding one Server, in two `sources=[...]` orders, on tie storms and random configurations."""
    spec = multi_source_spec(k)
    if spec["mode"] == "replicas":
        spec["trace"] = False
    out, meta = MG.run_case(spec)
    check_oracle_against_station_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(30)) + [1068, 1084, 1282])
def test_oracle_equals_live_reference_with_several_sources_per_server_on_rings(k):
    out, meta = MG.run_ring_case(multi_source_ring_spec(k))
    check_oracle_against_ring_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", range(20))
def test_oracle_equals_live_reference_with_probes_on_load_balancer_graphs(k):
    out, meta = MG.run_lb_case(lb_probe_spec(k))
    check_oracle_against_lb_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(12)) + [1351])
def test_oracle_equals_live_reference_with_profiles_on_load_balancer_sources(k):
    out, meta = MG.run_lb_case(lb_profile_spec(k))
    check_oracle_against_lb_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", range(60))
def test_oracle_equals_live_reference_on_random_tandem_queues(k):
    """Server(downstream=<Server>), up to four in a row (tests/tandem_specs.py; every fourth case a lock-step tie storm): the
    cases tests/test_gpu_tandem.py runs engine == oracle on MI355X."""
    import tandem_specs as TS

    out, meta = MG.run_tandem_case(TS.tandem_spec(k))
    check_oracle_against_tandem_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(40)) + [7932, 8264])
def test_oracle_equals_live_reference_on_tandem_queues_with_probes_and_injected_requests(k):
    """tests/tandem_specs.py tandem_probe_case: Probes on Servers of the chains and Simulation.schedule() Requests next to tandem
    queues -- statistics, Sink records, every Probe sample and the full trace with sort indices.  7932 / 8264: two Requests reach
    an idle one-worker Server in one nanosecond and the reference rejects the second at the worker (server.py:223-234)."""
    import tandem_specs as TS

    out, meta = MG.run_tandem_case(TS.tandem_probe_case(k))
    check_oracle_against_tandem_golden(H.Golden.from_results(out, meta))


@pytest.mark.parametrize("k", list(range(0, 60)) + [1788, 4095, 4329, 4857, 6237])
def test_oracle_equals_live_reference_on_forests_of_servers(k):
    """tests/tandem_specs.py fan_in_case: several Servers forwarding to one, Sources of their own on downstream Servers; every third
    case lock-step constants.  The five extra cases are the ones the GPU sweep differed on while fan-in was being built (the worker's
    rejection of a second same-nanosecond Request; the Sources' first ticks and the restarted sort counter)."""
    import tandem_specs as TS
    from test_oracle_golden import check_oracle_against_fan_in_reference

    case = TS.fan_in_case(k)
    check_oracle_against_fan_in_reference(case, MG.run_fan_in_case(case))
