"""CPU: the arithmetic fact behind the speculated arrival steps of the load-balancer engine's Sources (csrc/hs_lb.hip
`lb_step_encode`): with binary64 operations as the reference performs them (load/arrival_time_provider.py:72-82 through
core/temporal.py:188-211: ns' = int((ns / 1e9 + inc) * 1e9)), whenever frac(RN(inc * 1e9)) lies in [2^-10, 1 - 2^-10] and the
times stay below 2^40 ns, the next tick is exactly ns + floor(RN(inc * 1e9)).  numpy float64 does the same IEEE-754 operations."""
import numpy as np


def exact_step(ns, inc):
    return np.trunc((ns / 1e9 + inc) * 1e9)


def test_whole_nanosecond_step_is_exact_away_from_the_margin():
    rng = np.random.default_rng(7)
    for scale, n in ((6e10, 4_000_000), (1.0e12, 2_000_000), (1e6, 1_000_000)):
        ns = np.floor(rng.uniform(0, scale, n))
        inc = rng.exponential(1.0 / rng.choice([0.5, 6.0, 8.0, 1000.0, 3e5], n))
        F = inc * 1e9
        fl = np.floor(F)
        frac = F - fl
        safe = (frac >= 2.0 ** -10) & (frac <= 1.0 - 2.0 ** -10) & (ns + F < 2.0 ** 40)
        assert safe.mean() > 0.99
        np.testing.assert_array_equal(exact_step(ns[safe], inc[safe]), ns[safe] + fl[safe])
    # adversarial: increments whose product sits just inside the margin, at the largest times allowed
    ns = np.floor(rng.uniform(2.0 ** 39, 2.0 ** 40 - 2e9, 2_000_000))
    k = np.floor(rng.uniform(1, 1e9, 2_000_000))
    for eps in (2.0 ** -10, 1.0 - 2.0 ** -10, 1.5 * 2.0 ** -10):
        inc = (k + eps) / 1e9
        F = inc * 1e9
        fl = np.floor(F)
        frac = F - fl
        safe = (frac >= 2.0 ** -10) & (frac <= 1.0 - 2.0 ** -10)
        np.testing.assert_array_equal(exact_step(ns[safe], inc[safe]), ns[safe] + fl[safe])
    # ... and the margin is needed: right at a whole number the two differ sometimes (constant rates do this on every tick)
    inc = k / 1e9
    assert (exact_step(ns, inc) != ns + np.floor(inc * 1e9)).any()
