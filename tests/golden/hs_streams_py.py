"""Pure-Python statement of the engine's counter-based random streams.

TEST TOOLING: a third, independent implementation (after oracle/hs_rng_ref.h in
C and happy_simulator_amd/csrc in HIP) of the stream definition in DESIGN.md
"Random streams".  Used by make_golden.py to plug per-entity Philox streams
into the LIVE reference through its own extension points
(`LatencyDistribution.get_latency`, `ArrivalTimeProvider._get_target_integral_value`
-- SURVEY.md 8(b) Seam 3), and by CPU tests to cross-check the C oracle.

Python floats are IEEE binary64 and CPython never contracts a*b+c, so the
arithmetic below is bit-identical to the C / HIP versions.
"""
from __future__ import annotations

import struct

M32 = 0xFFFFFFFF
STREAM_ARRIVAL, STREAM_SERVICE, STREAM_LINK, STREAM_ROUTE, STREAM_KEY, STREAM_LOSS = 0, 1, 2, 3, 4, 5


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def res53(a: int, b: int) -> float:
    return (float(a >> 5) * 67108864.0 + float(b >> 6)) / 9007199254740992.0


def uniform(seed: int, sid: int, k: int) -> float:
    b = k >> 1
    o = philox4x32_10((b & M32, (b >> 32) & M32, sid & M32, (sid >> 32) & M32), (seed & M32, (seed >> 32) & M32))
    return res53(o[2], o[3]) if (k & 1) else res53(o[0], o[1])


_LN2_HI = 6.93147180369123816490e-01
_LN2_LO = 1.90821492927058770002e-10
_LG = (6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
       1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01)


def hs_log(x: float) -> float:
    (ix,) = struct.unpack("<Q", struct.pack("<d", x))
    hx = ix >> 32
    k = (hx >> 20) - 1023
    hx &= 0x000FFFFF
    i = (hx + 0x95F64) & 0x100000
    hx |= i ^ 0x3FF00000
    k += i >> 20
    (m,) = struct.unpack("<d", struct.pack("<Q", (hx << 32) | (ix & M32)))
    f = m - 1.0
    hfsq = (0.5 * f) * f
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    Lg1, Lg2, Lg3, Lg4, Lg5, Lg6, Lg7 = _LG
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6))
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)))
    R = t2 + t1
    dk = float(k)
    return (((s * (hfsq + R) + dk * _LN2_LO) - hfsq) + f) + dk * _LN2_HI


def exp1(u: float) -> float:
    return -hs_log(1.0 - u)


class Stream:
    """Draw counter for one (seed, entity, kind) stream."""

    def __init__(self, seed: int, stream_base: int, kind: int):
        self.seed = seed
        self.sid = (stream_base << 3) | kind
        self.k = 0

    def next_uniform(self) -> float:
        u = uniform(self.seed, self.sid, self.k)
        self.k += 1
        return u
