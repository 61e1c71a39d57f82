"""GPU: K lanes per LP (csrc/hs_kernels_wide.hpp, the strong-scaling kernel of the headline grid) against one lane per LP
(hs_station_run, which the oracle pins at full size: tests/test_gpu_fullsize_oracle.py) -- everything the ABI reports must be
identical: totals, per-kind histogram, final time (= the elected event beyond end_time), every per-LP statistic incl. the
binary64 `total_service_time`, every Sink record; and the LP state it leaves must continue identically (a second run_until on
the one-lane kernel).  Debug flags (csrc/hs_engine.hip wide_lanes): 1 << 22 keeps the one-lane kernel, bits 24..27 force K,
1 << 21 sends every 97th LP through the bail path (event-order loop in hs_station_wide_finish)."""
import hashlib

import numpy as np
import pytest

from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu

ONE_LANE = 1 << 22


def _force(k):
    """4 / 8 / 16 lanes per LP (hs_station_wide); 64 / 65: a wavefront per LP with 16 / 8 LPs per workgroup (hs_station_wave)"""
    return ({4: 3, 8: 4, 16: 5, 64: 7, 65: 8}[k]) << 24


def _run(n, end_ns, flags, seed=42, rate=8.0, mean=0.1, second_end=None, start_ns=0):
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    st = StationArrays.uniform(n, rate=rate, mean=mean)
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=second_end or end_ns, seed=seed, start_ns=start_ns) as eng:
        eng.set_debug_flags(flags)
        eng.run_until(end_ns)
        if second_end:
            eng.run_until(second_end)        # (not fresh any more: the one-lane kernel continues from the state the first left)
        s = eng.summary()
        out = {"tot_events": s.events_processed, "tot_final": s.final_time_ns, "tot_completed": s.requests_completed,
               "tot_sink_records": s.sink_records, "by_kind": s.events_by_kind.copy()}
        out.update(eng.lp_stats())
        c, t, cr = eng.read_sinks()
        out.update(sink_counts=c, sink_t=t, sink_created=cr)
    return out


def _same(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=f"{what}: {k}")


@pytest.mark.parametrize("n,end_s", [(1, 5.0), (7, 3.0), (64, 10.0), (300, 2.5), (4096, 6.0), (8192, 60.0), (32768, 12.0), (43000, 5.0)])
def test_wide_kernel_equals_the_one_lane_kernel(n, end_s):
    end = int(end_s * 1e9)
    ref = _run(n, end, ONE_LANE)
    assert ref["tot_events"] > 20 * n * end_s
    for k in (4, 8, 16, 64, 65):
        _same(_run(n, end, _force(k)), ref, f"K = {k}")
    _same(_run(n, end, 0), ref, "automatic K")


def test_wide_kernel_other_rates_and_seeds():
    for seed, rate, mean in ((1, 3.0, 0.2), (99, 20.0, 0.04), (7, 9.5, 0.11)):       # under- and overloaded
        end = 8_000_000_000
        ref = _run(500, end, ONE_LANE, seed=seed, rate=rate, mean=mean)
        _same(_run(500, end, _force(8), seed=seed, rate=rate, mean=mean), ref, f"seed {seed}")
        _same(_run(500, end, _force(16), seed=seed, rate=rate, mean=mean), ref, f"seed {seed}")
        _same(_run(500, end, _force(64), seed=seed, rate=rate, mean=mean), ref, f"seed {seed}, a wavefront per LP")
        _same(_run(500, end, _force(65), seed=seed, rate=rate, mean=mean), ref, f"seed {seed}, a wavefront per LP, 8 per workgroup")


def test_bailed_lps_rerun_in_event_order():
    end = 6_000_000_000
    ref = _run(2000, end, ONE_LANE)
    _same(_run(2000, end, _force(8) | (1 << 21)), ref, "every 97th LP bails")
    _same(_run(2000, end, _force(64) | (1 << 21)), ref, "every 97th LP bails, a wavefront per LP")


def test_the_state_the_wide_kernel_leaves_continues_identically():
    """Window one on the wide kernel, window two on the one-lane kernel (pending ticks, the request in service, creation
    stamps, lineage, draw counters: everything a continuation reads) == both windows on the one-lane kernel."""
    ref = _run(1000, 3_000_000_000, ONE_LANE, second_end=7_500_000_000)
    _same(_run(1000, 3_000_000_000, _force(16), second_end=7_500_000_000), ref, "two windows")
    ref = _run(1000, 100_000_000, ONE_LANE, second_end=2_000_000_000)                 # a window in which most LPs see no tick
    _same(_run(1000, 100_000_000, _force(4), second_end=2_000_000_000), ref, "short first window")
    _same(_run(1000, 100_000_000, _force(64), second_end=2_000_000_000), ref, "short first window, a wavefront per LP")
    ref = _run(1000, 3_000_000_000, ONE_LANE, second_end=7_500_000_000)
    _same(_run(1000, 3_000_000_000, _force(64), second_end=7_500_000_000), ref, "two windows, a wavefront per LP")


def test_wide_kernel_against_the_oracle():
    import helpers as H
    from test_gpu_parity import _compare_engine_to_oracle

    n = 200
    spec = dict(name="wide_200", n_chains=n, arr="poisson", rate=8.0, svc="exp", mean=0.1, concurrency=1, queue_cap=None,
                stop_after_s=None, downstream=True, end_s=12.0, rng="philox", seed=4242, mode="single", trace=False)
    runs = H.run_oracle_for_spec(spec)
    for flags in (_force(8), _force(16), _force(64), _force(65)):
        eng, p = H.engine_for_spec(spec)
        with eng:
            eng.set_debug_flags(flags)
            eng.run_until(p["end_ns"])
            _compare_engine_to_oracle(spec, eng, p, runs)


def test_8192_lps_run_at_least_twice_as_fast_as_one_lane_each():
    """bench.py --n-lp 8192 (= the 8-GPU strong shard of the metric's 65 536 servers on one GPU): the wide kernel's step time."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays, StationEngine

    end = 60_000_000_000
    times = {}
    for name, flags in (("one lane per LP", ONE_LANE), ("wide", 0)):
        st = StationArrays.uniform(8192, rate=8.0, mean=0.1)
        with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end, seed=42) as eng:
            eng.set_debug_flags(flags)
            eng.bench_runs(end, 3)
            k, _ = eng.bench_runs(end, 10)
            times[name] = float(np.median(k))
    print(times)
    assert times["wide"] < 0.35 * times["one lane per LP"], times      # (measured r4: 0.168 vs 0.39 ms; r5, a wavefront per LP: see DESIGN section 6)


def test_wave_kernel_on_random_uniform_grids():
    """Round 5: one wavefront per LP (hs_station_wave, both workgroup shapes, the FRESH and the loading instantiation) against one lane
    per LP on 160 random uniform grids -- sizes 1 .. 3 000, rates 0.002 .. 60 per second (the slowest step by more than 2^39 ns: every
    arrival takes the reference's own step, no speculation), services 1 ms .. 0.8 s (over- and underloaded), horizons 0.02 .. 40 s, and
    a second window on top of half of them (the state the kernel leaves must continue identically)."""
    import os
    n_cases = int(os.environ.get("HS_WAVE_SWEEP_CASES", "160"))     # (a one-off sweep: profiles/r05_wave_sweep.log ran 4 000 with seed 7)
    rng = np.random.default_rng(int(os.environ.get("HS_WAVE_SWEEP_SEED", "2026")))
    for case in range(n_cases):
        n = int(rng.choice([1, 2, 5, 17, 63, 64, 65, 200, 513, 1500, 3000]))
        rate = float(rng.choice([0.002, 0.05, 0.7, 3.0, 8.0, 8.0, 20.0, 60.0]))
        mean = float(rng.choice([0.001, 0.02, 0.1, 0.1, 0.3, 0.8]))
        end = int(float(rng.choice([0.02, 0.3, 2.0, 7.0, 40.0])) * 1e9)
        if rate * (end / 1e9) * n > 3.0e6:          # keep a case below a few million requests
            end = int(3.0e6 / (rate * n) * 1e9)
        seed = int(rng.integers(1, 1 << 30))
        start = int(rng.choice([0, 0, 17, 1_500_000_000]))                   # Simulation(start_time=...): the bootstrap draws from there
        end += start
        second = start + int((end - start) * float(rng.choice([1.3, 2.0]))) if case % 2 else None
        kw = dict(seed=seed, rate=rate, mean=mean, second_end=second, start_ns=start)
        ref = _run(n, end, ONE_LANE, **kw)
        for k in (64, 65):
            _same(_run(n, end, _force(k), **kw), ref, f"case {case}: n {n} end {end} K {k} {kw}")
        _same(_run(n, end, _force(64) | (1 << 29), **kw), ref, f"case {case} (reset kernel + the loading instantiation): n {n} end {end} {kw}")
