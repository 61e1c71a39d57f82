"""Host-side mirror of the reference's time types (happysimulator/core/temporal.py).

Time is int64 nanoseconds; floats enter by TRUNCATION: `from_seconds(x) = int(x * 1e9)`
(reference :62, :205), `Instant + float = ns + int(d * 1e9)` (:222), `to_seconds = float(ns) / 1e9` (:66, :211).
The device code (csrc/hs_device.hpp) applies the same rules; these classes only carry values across the API.
"""
from __future__ import annotations

from typing import Union

_NS = 1_000_000_000


class Duration:
    __slots__ = ("nanoseconds",)

    def __init__(self, nanoseconds: int):
        self.nanoseconds = nanoseconds

    @classmethod
    def from_seconds(cls, seconds: Union[int, float]) -> "Duration":
        if isinstance(seconds, int):
            return cls(seconds * _NS)
        if isinstance(seconds, float):
            return cls(int(seconds * _NS))
        raise TypeError("seconds must be int or float")

    def to_seconds(self) -> float:
        return float(self.nanoseconds) / _NS

    def __add__(self, other):
        if isinstance(other, Duration):
            return Duration(self.nanoseconds + other.nanoseconds)
        if isinstance(other, (int, float)):
            return Duration(self.nanoseconds + int(other * _NS))
        return NotImplemented

    def __radd__(self, other):
        if isinstance(other, (int, float)):
            return Duration(int(other * _NS) + self.nanoseconds)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, Duration):
            return Duration(self.nanoseconds - other.nanoseconds)
        if isinstance(other, (int, float)):
            return Duration(self.nanoseconds - int(other * _NS))
        return NotImplemented

    def __mul__(self, other):
        if isinstance(other, (int, float)):
            return Duration(int(self.nanoseconds * other))
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, (int, float)):
            return Duration(int(self.nanoseconds / other))
        return NotImplemented

    def _cmp(self, other):
        if not isinstance(other, Duration):
            return NotImplemented
        return self.nanoseconds - other.nanoseconds

    def __eq__(self, other):
        return isinstance(other, Duration) and self.nanoseconds == other.nanoseconds

    def __lt__(self, other):
        return self._cmp(other) < 0

    def __le__(self, other):
        return self._cmp(other) <= 0

    def __gt__(self, other):
        return self._cmp(other) > 0

    def __ge__(self, other):
        return self._cmp(other) >= 0

    def __hash__(self):
        return hash(self.nanoseconds)

    def __repr__(self):
        return f"Duration({self.to_seconds():.9f}s)"


Duration.ZERO = Duration(0)


class Instant:
    __slots__ = ("nanoseconds",)

    def __init__(self, nanoseconds: int):
        self.nanoseconds = nanoseconds

    @classmethod
    def from_seconds(cls, seconds: Union[int, float]) -> "Instant":
        if isinstance(seconds, int):
            return cls(seconds * _NS)
        if isinstance(seconds, float):
            return cls(int(seconds * _NS))
        raise TypeError("seconds must be int or float")

    def to_seconds(self) -> float:
        return float(self.nanoseconds) / _NS

    def __add__(self, other):
        if isinstance(other, Duration):
            return Instant(self.nanoseconds + other.nanoseconds)
        if isinstance(other, (int, float)):
            return Instant(self.nanoseconds + int(other * _NS))
        return NotImplemented

    def __radd__(self, other):
        if isinstance(other, (int, float)):
            return Instant(int(other * _NS) + self.nanoseconds)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, Instant):
            return Duration(self.nanoseconds - other.nanoseconds)
        if isinstance(other, Duration):
            return Instant(self.nanoseconds - other.nanoseconds)
        if isinstance(other, (int, float)):
            return Instant(self.nanoseconds - int(other * _NS))
        return NotImplemented

    def __eq__(self, other):
        if not isinstance(other, Instant):
            return NotImplemented
        return self.nanoseconds == other.nanoseconds

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __lt__(self, other):
        if not isinstance(other, Instant):
            return NotImplemented
        return self.nanoseconds < other.nanoseconds

    def __le__(self, other):
        if not isinstance(other, Instant):
            return NotImplemented
        return self.nanoseconds <= other.nanoseconds

    def __gt__(self, other):
        if not isinstance(other, Instant):
            return NotImplemented
        return self.nanoseconds > other.nanoseconds

    def __ge__(self, other):
        if not isinstance(other, Instant):
            return NotImplemented
        return self.nanoseconds >= other.nanoseconds

    def __hash__(self):
        return hash(self.nanoseconds)

    def __repr__(self):
        total_us = self.nanoseconds // 1_000
        us = total_us % 1_000_000
        total_seconds = total_us // 1_000_000
        return f"T{total_seconds // 3600:02d}:{(total_seconds // 60) % 60:02d}:{total_seconds % 60:02d}.{us:06d}"


class _InfiniteInstant(Instant):
    """Instant.Infinity: compares after every finite Instant (reference :298-368)."""

    def __init__(self):
        super().__init__(2**63 - 1)

    def to_seconds(self) -> float:
        return float("inf")

    def __add__(self, other):
        return self

    def __sub__(self, other):
        if isinstance(other, Instant):
            raise ValueError("cannot subtract from Instant.Infinity")
        return self

    def __eq__(self, other):
        return isinstance(other, _InfiniteInstant)

    def __lt__(self, other):
        return False

    def __le__(self, other):
        return isinstance(other, _InfiniteInstant)

    def __gt__(self, other):
        return not isinstance(other, _InfiniteInstant)

    def __ge__(self, other):
        return True

    def __hash__(self):
        return hash("Instant.Infinity")

    def __repr__(self):
        return "Instant.Infinity"


Instant.Epoch = Instant(0)
Instant.Infinity = _InfiniteInstant()
