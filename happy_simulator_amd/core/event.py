"""`Event` -- the caller-facing record of happysimulator/core/event.py:106-182, for `Simulation.schedule()`.

On the engine an Event is not an object of the run: requests live as timestamps in per-LP logs.  This class only
carries what a caller hands to `Simulation.schedule()` before `run()`: a time, a type string and a target entity.
`context["created_at"]` defaults to the event's own time exactly as in the reference (core/event.py:170-181)."""
from __future__ import annotations

from typing import Any

from .temporal import Instant


class Event:
    __slots__ = ("time", "event_type", "target", "daemon", "on_complete", "context", "_cancelled")

    def __init__(self, time: Instant, event_type: str, target=None, *, daemon: bool = False,
                 on_complete: list | None = None, context: dict[str, Any] | None = None):
        if target is None:
            raise ValueError(f"Event '{event_type}' must have a 'target'.")          # core/event.py:159-160
        self.time = time
        self.event_type = event_type
        self.target = target
        self.daemon = daemon
        self.on_complete = on_complete if on_complete is not None else []
        self._cancelled = False
        if context is not None:
            self.context = context
            context.setdefault("created_at", self.time)
            context.setdefault("metadata", {})
        else:
            self.context = {"created_at": self.time, "metadata": {}}

    @property
    def cancelled(self) -> bool:
        return self._cancelled

    def cancel(self) -> None:
        """core/event.py:189-194: a cancelled event is skipped when popped (counted in events_cancelled)."""
        self._cancelled = True

    def __repr__(self) -> str:
        return f"Event({self.time!r}, {self.event_type!r}, target={getattr(self.target, 'name', self.target)!r})"
