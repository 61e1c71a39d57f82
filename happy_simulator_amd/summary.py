"""What `Simulation.run()` hands back: the output contract of happysimulator/instrumentation/summary.py:14-87, kept field
for field (names, `to_dict()` keys, the text `print(summary)` shows) because callers and notebooks read it.

The engine fills these from device counters (`hs_engine_get_summary`, per-LP statistics); nothing here computes anything
beyond formatting.  The text layout is described once, as rows of (label, how to render), instead of being spelled out in
every `__str__`; tests/test_host_api.py compares the rendered text and the dictionaries with the live reference classes."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Iterable


def render(title: str, rows: Iterable[tuple[str, str]]) -> str:
    """`title` followed by one two-space indented `label: value` line per row."""
    return "\n".join([title] + [f"  {label}: {value}" for label, value in rows])


@dataclass
class QueueStats:
    """Queue counters of a QueuedResource; `peak_depth` is always 0 in the reference too (core/simulation.py:572)."""

    peak_depth: int
    total_accepted: int
    total_dropped: int

    def as_text(self) -> str:
        return f"peak={self.peak_depth}, accepted={self.total_accepted}, dropped={self.total_dropped}"


@dataclass
class EntitySummary:
    name: str
    entity_type: str
    events_handled: int
    queue_stats: QueueStats | None = None

    def to_dict(self) -> dict[str, Any]:
        d: dict[str, Any] = dict(name=self.name, type=self.entity_type, events_handled=self.events_handled)
        q = self.queue_stats
        if q is not None:
            d["queue"] = dict(peak_depth=q.peak_depth, total_accepted=q.total_accepted, total_dropped=q.total_dropped)
        return d

    def as_text(self) -> str:
        text = f"{self.name} ({self.entity_type}): {self.events_handled} events"
        return text if self.queue_stats is None else f"{text} | queue: {self.queue_stats.as_text()}"


# (dictionary key, text label, renderer) of the scalar fields, in the reference's order
_SCALARS: tuple[tuple[str, str, Callable[[Any], str]], ...] = (
    ("total_events_processed", "Events processed", str),
    ("events_cancelled", "Events cancelled", str),
    ("events_per_second", "Events/sec (sim)", lambda v: f"{v:.1f}"),
)


class LazyEntities(dict):
    """`SimulationSummary.entities`, built on first use.  The reference builds the per-entity summaries inside `run()`
    (core/simulation.py:560-591); for 131 072 entities that loop costs more than the whole device run, so the dict is filled
    when somebody looks at it -- same keys, same values, same order."""

    def __init__(self, build):
        super().__init__()
        self._build = build

    def _fill(self):
        b, self._build = self._build, None
        if b is not None:
            super().update(b())

    def __getitem__(self, k): self._fill(); return super().__getitem__(k)
    def __iter__(self): self._fill(); return super().__iter__()
    def __len__(self): self._fill(); return super().__len__()
    def __contains__(self, k): self._fill(); return super().__contains__(k)
    def __eq__(self, o): self._fill(); return super().__eq__(o)
    def __ne__(self, o): self._fill(); return super().__ne__(o)
    def __repr__(self): self._fill(); return super().__repr__()
    def __bool__(self): self._fill(); return super().__len__() > 0
    def keys(self): self._fill(); return super().keys()
    def values(self): self._fill(); return super().values()
    def items(self): self._fill(); return super().items()
    def get(self, k, d=None): self._fill(); return super().get(k, d)
    def copy(self): self._fill(); return dict(self)
    def update(self, *a, **kw): self._fill(); return super().update(*a, **kw)
    def setdefault(self, k, d=None): self._fill(); return super().setdefault(k, d)
    def pop(self, *a): self._fill(); return super().pop(*a)
    def __setitem__(self, k, v): self._fill(); return super().__setitem__(k, v)
    def __delitem__(self, k): self._fill(); return super().__delitem__(k)
    def __reduce__(self): self._fill(); return (dict, (dict(self),))
    __hash__ = None


@dataclass
class SimulationSummary:
    duration_s: float
    total_events_processed: int
    events_cancelled: int = 0
    events_per_second: float = 0.0      # per SIMULATED second (core/simulation.py:547), not wall clock
    wall_clock_seconds: float = 0.0
    entities: dict[str, EntitySummary] = field(default_factory=dict)

    def to_dict(self) -> dict[str, Any]:
        d: dict[str, Any] = {"duration_s": self.duration_s}
        d.update((key, getattr(self, key)) for key, _, _ in _SCALARS)
        d["wall_clock_seconds"] = self.wall_clock_seconds
        d["entities"] = {name: e.to_dict() for name, e in self.entities.items()}
        return d

    def __str__(self) -> str:
        rows = [("Duration", f"{self.duration_s:.2f}s (sim) / {self.wall_clock_seconds:.3f}s (wall)")]
        rows += [(label, fmt(getattr(self, key))) for key, label, fmt in _SCALARS]
        text = render("Simulation Summary", rows)
        if self.entities:
            text += "\n  Entities:" + "".join(f"\n    {e.as_text()}" for e in self.entities.values())
        return text
