"""CPU: scalar semantics of the oracle -- Philox known-answer vectors, MT19937 against CPython/numpy,
hs_log against an independent pure-Python statement and libm, truncation rules against the reference's
own regression numbers."""
import math
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import hs_streams_py as PY  # noqa: E402
from oracle import hs_oracle as O  # noqa: E402


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors (philox4x32-10)
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert list(PY.philox4x32_10((0, 0, 0, 0), (0, 0))) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]


def test_uniform_streams_c_equals_python():
    rnd = random.Random(1)
    for _ in range(300):
        seed = rnd.getrandbits(64)
        sid = rnd.getrandbits(50)
        k = rnd.getrandbits(40)
        assert O.uniform(seed, sid, k) == PY.uniform(seed, sid, k)
        assert 0.0 <= O.uniform(seed, sid, k) < 1.0
    # both halves of one Philox block
    assert O.uniform(42, 8, 0) != O.uniform(42, 8, 1)


def test_hs_log_c_equals_python_and_is_accurate():
    rnd = random.Random(2)
    worst = 0.0
    xs = [1.0, 0.5, 2.0 ** -53, 1.0 - 2.0 ** -53, 0.7071067811865476, 0.7071067811865475]
    xs += [rnd.random() * (1 - 2.0 ** -53) + 2.0 ** -53 for _ in range(20000)]
    xs += [2.0 ** -rnd.randint(1, 52) * (1 + rnd.random()) for _ in range(2000)]
    for x in xs:
        a = O.log(x)
        assert a == PY.hs_log(x)
        b = math.log(x)
        if b != 0.0:
            worst = max(worst, abs(a - b) / math.ulp(b))
        else:
            assert a == 0.0
    assert worst <= 1.0   # faithfully rounded on the engine's domain [2^-53, 1]


def test_mt19937_matches_cpython_and_numpy():
    L = O.lib()
    for seed in (0, 1, 42, 2**31 + 5, 2**32 - 1):
        random.seed(seed)
        np.random.seed(seed)
        for i in range(5):
            assert L.hso_mt_py_random(seed, i) == random.random()
            assert L.hso_mt_np_random(seed, i) == np.random.random()


def test_truncation_rules_reproduce_reference_regression_numbers():
    """tests/regression/test_arrival_time_regression.py:20-31 (reference): constant rate 50/s gives
    0.02, 0.04, ... with the ns-truncation drift 0.199999999 at the 10th arrival."""
    g = O.Graph()
    s = g.source(O.ARR_CONSTANT, 50.0)
    k = g.sink()
    g.target[s] = k
    r = O.run(g, 1_000_000_000)
    t, _ = r.sinks[k]
    secs = t[:10].astype(np.float64) / 1e9
    expect = [0.02, 0.04, 0.06, 0.08, 0.1, 0.12, 0.14, 0.16, 0.18, 0.199999999]
    assert all(abs(a - b) <= 1e-8 for a, b in zip(secs, expect))
    assert t[9] == 199_999_999


def test_counter_overshoot_known_answer():
    """tests/integration/core_simulation/test_simulation_basic_counter.py:7-34 (reference): a 1 Hz
    constant source into a counter for 60 s: 61 ticks generated, 60 counted -- the one-event overshoot."""
    g = O.Graph()
    s = g.source(O.ARR_CONSTANT, 1.0)
    k = g.sink()
    g.target[s] = k
    r = O.run(g, 60_000_000_000)
    assert r.generated[s] == 61
    assert r.received[k] == 60


class TestProfileArrivalKnownAnswers:
    """The reference's own golden vectors for time-varying profiles, restated
    (tests/regression/test_arrival_time_regression.py:46-103,126-165: ConstantArrivalTimeProvider over LinearRampProfile /
    SpikeProfile, tolerance 1e-8 upstream).  The oracle's general path (adaptive Simpson + bracket + Brent,
    load/arrival_time_provider.py:84-144) is additionally bit-exact against the LIVE reference through the
    tests/golden/profile_*.npz fixtures."""

    LINEAR_RAMP_10_100_FIRST_10 = [0.095864499, 0.184655976, 0.267741515, 0.346097448, 0.420449859, 0.49135612,
                                   0.55925515, 0.624499924, 0.687379335, 0.748133387]
    LINEAR_RAMP_100_10_FIRST_10 = [0.010004504, 0.020018032, 0.030040609, 0.040072259, 0.050113007, 0.060162878,
                                   0.070221897, 0.080290089, 0.090367479, 0.100454092]
    SPIKE_PROFILE_FIRST_30 = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.799999999, 0.899999998, 0.999999998, 1.099999998,
                              1.199999998, 1.299999998, 1.399999998, 1.499999998, 1.599999998, 1.699999998, 1.799999998,
                              1.899999998, 1.999999997, 2.009999997, 2.019999996, 2.029999996, 2.039999995, 2.049999994,
                              2.059999994, 2.069999993, 2.079999993, 2.089999992, 2.099999991]

    @staticmethod
    def _arrivals(profile, n):
        t, out = 0, []
        for _ in range(n):
            t = O.profile_next_arrival(profile, t, 1.0)      # ConstantArrivalTimeProvider: target area 1.0
            assert t >= 0
            out.append(t / 1e9)
        return out

    def test_linear_ramp_up(self):
        got = self._arrivals(("ramp", 10.0, 10.0, 100.0), 10)
        assert max(abs(a - b) for a, b in zip(got, self.LINEAR_RAMP_10_100_FIRST_10)) < 1e-8

    def test_linear_ramp_down(self):
        got = self._arrivals(("ramp", 10.0, 100.0, 10.0), 10)
        assert max(abs(a - b) for a, b in zip(got, self.LINEAR_RAMP_100_10_FIRST_10)) < 1e-8

    def test_spike(self):
        got = self._arrivals(("spike", 10.0, 100.0, 2.0, 1.0), 30)
        assert max(abs(a - b) for a, b in zip(got, self.SPIKE_PROFILE_FIRST_30)) < 1e-8

    def test_zero_rate_forever_is_an_error(self):
        """`RuntimeError: Could not find event ...` upstream (arrival_time_provider.py:124-132)."""
        assert O.profile_next_arrival(("ramp", 1.0, 0.0, 0.0), 0, 1.0) == -1
