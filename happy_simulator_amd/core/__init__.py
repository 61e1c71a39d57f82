from .temporal import Duration, Instant  # noqa: F401
