"""happy_simulator_amd -- MI355X-native discrete-event engine behind the happy-simulator API.

Only the hot path lives here: `Simulation.run()`'s event loop and
`happysimulator.parallel`, re-designed as a GPU-resident engine (csrc/, C ABI in include/hs_engine.h)
with a host-side mirror of the reference's `Simulation / Source / Server / Sink / Instant` API.
"""
from ._native import EngineError, EngineUnavailable  # noqa: F401

__version__ = "0.1.0"
