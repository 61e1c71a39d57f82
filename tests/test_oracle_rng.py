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
