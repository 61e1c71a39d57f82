"""Host-side mirror of the reference's entity constructors for the hot path.

These classes carry PARAMETERS into the engine and RESULTS back out; they contain no event logic (that
lives in csrc/).  Names, argument meaning, defaults and error behaviour follow the reference:

  Source.poisson / Source.constant / Source(name, event_provider, arrival_time_provider)
                                                   happysimulator/load/source.py:93-268
  SimpleEventProvider(target, event_type, stop_after)            load/source.py:31-86
  ExponentialLatency / ConstantLatency             distributions/exponential.py:15-45, constant.py:15-35
  Server(name, concurrency, service_time, queue_policy, queue_capacity, downstream)
                                                   components/server/server.py:43-273
  Sink / Counter                                   components/common.py:18-95
  LatencyTracker                                   instrumentation/collectors.py:18-60
  NetworkLink(name, latency, bandwidth_bps, packet_loss_rate, jitter, egress)
                                                   components/network/link.py:36-234
  RandomRouter(name, targets=[...])                components/random_router.py:9-45
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .core.temporal import Duration, Instant


class Entity:
    """Base of every simulation actor (core/entity.py:31-127): a name and topology links."""

    def __init__(self, name: str):
        self.name = name

    def downstream_entities(self) -> list["Entity"]:
        return []

    def has_capacity(self) -> bool:
        return True


# ---- distributions -------------------------------------------------------------------------------
class LatencyDistribution:
    __slots__ = ("_mean_latency",)      # (slotted like the other leaf objects of a chain: lowering 65 536 chains is a walk over
                                        #  ~600 000 Python objects, bound by the cache lines each visit touches)

    def __init__(self, mean_latency):
        if isinstance(mean_latency, Duration):
            self._mean_latency = mean_latency.to_seconds()
        else:
            self._mean_latency = float(mean_latency)

    @property
    def mean(self) -> float:
        return self._mean_latency


class ExponentialLatency(LatencyDistribution):
    """Exponentially distributed latency; on the engine: sample = -hs_log(1-u) / (1/mean)."""
    __slots__ = ("_lambda",)

    def __init__(self, mean_latency):
        super().__init__(mean_latency)
        self._lambda = 1 / self._mean_latency


class ConstantLatency(LatencyDistribution):
    __slots__ = ()


# ---- queue policy --------------------------------------------------------------------------------
class FIFOQueue:
    """components/queue_policy.py:75-114 -- the only policy lowered to the engine."""
    __slots__ = ("_capacity",)

    def __init__(self, capacity: float = float("inf")):
        self._capacity = capacity

    @property
    def capacity(self) -> float:
        return self._capacity


# ---- load ----------------------------------------------------------------------------------------
@dataclass(frozen=True, slots=True)
class ConstantRateProfile:
    rate: float

    def get_rate(self, time: Instant) -> float:
        return self.rate

    @property
    def peak_rate(self) -> float:
        return self.rate


@dataclass(frozen=True)
class LinearRampProfile:
    """Rate ramping linearly from start_rate to end_rate over duration_s, constant afterwards (load/profile.py:52-75)."""

    duration_s: float
    start_rate: float
    end_rate: float

    def get_rate(self, time: Instant) -> float:
        t = time.to_seconds()
        if t <= 0:
            return self.start_rate
        if t >= self.duration_s:
            return self.end_rate
        fraction = t / self.duration_s
        return self.start_rate + fraction * (self.end_rate - self.start_rate)

    @property
    def peak_rate(self) -> float:
        return max(self.start_rate, self.end_rate)


@dataclass(frozen=True)
class SpikeProfile:
    """baseline_rate, then spike_rate during [warmup_s, warmup_s + spike_duration_s), then baseline_rate again
    (load/profile.py:78-113)."""

    baseline_rate: float = 10.0
    spike_rate: float = 150.0
    warmup_s: float = 10.0
    spike_duration_s: float = 15.0

    def get_rate(self, time: Instant) -> float:
        t = time.to_seconds()
        if t < self.warmup_s:
            return self.baseline_rate
        if t < self.warmup_s + self.spike_duration_s:
            return self.spike_rate
        return self.baseline_rate

    @property
    def peak_rate(self) -> float:
        return max(self.baseline_rate, self.spike_rate)


# Results of a run on n plain chains whose objects have not been bound to them yet (lowering.write_back_plain): binding 4 x 65 536
# objects costs more than the run, so it happens when somebody first reads or writes a result attribute of ANY lowered entity
# (or starts another run) -- the same laziness as the Sink records that stay on the device until they are read.
_PENDING: list = []


def _flush_pending() -> None:
    if not _PENDING:
        return
    import gc

    was = gc.isenabled()
    gc.disable()                   # (65 536 new tuples would otherwise trigger full collections over ~600 000 live objects)
    try:
        while _PENDING:
            _PENDING.pop(0)()
    finally:
        if was:
            gc.enable()


_LAZY_LOOKUPS = 32                 # objects bound one by one (an identity search in the run's object lists) before everything is


def _resolve(obj) -> None:
    """A result attribute of `obj` is being touched while a run's results are not bound yet.  Round 5: bind THIS object alone --
    the pending run (lowering.write_back_plain) finds its row by an identity search, ~1 ms at 65 536 chains -- instead of all
    4 x n objects (16-33 ms there: the first counter read after run() cost 70 x the device run); whoever walks many objects gets
    the bulk binding after a few lookups.  Several pending runs, or an object the run does not know: bind everything, in run order."""
    if len(_PENDING) == 1:
        ent = _PENDING[0]
        find = getattr(ent, "find_and_bind", None)
        if find is not None and ent.lookups < _LAZY_LOOKUPS and find(obj):
            ent.lookups += 1
            return
    _flush_pending()


class _Stat:
    """A result counter of an entity: its own value, or -- after a run on n plain chains (lowering.write_back_plain) -- row
    `_bound[1]` of the run's per-LP result arrays `_bound[0]`.  Binding an object is ONE attribute store instead of one per
    counter (196 608 objects at the headline size); assigning a value un-binds the object (the general write-back sets them all)."""

    def __init__(self, key, cast=int, default=0):
        self.key, self.cast, self.default = key, cast, default

    def __set_name__(self, owner, name):
        self.own = "_own" + name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        if _PENDING:
            _resolve(obj)
        b = obj._bound
        if b is None:
            return obj.__dict__.get(self.own, self.default)
        key = self.key
        return self.cast(b[0][key][b[1]]) if isinstance(key, str) else key(b[0], b[1])

    def __set__(self, obj, value):
        if _PENDING:
            _flush_pending()
        obj.__dict__[self.own] = value
        if obj._bound is not None:
            obj._bound = None


class SimpleEventProvider:
    # (the attributes the lowering reads live in slots -- inline in the object, no dictionary to chase; everything else, the
    #  _Stat values included, goes to the instance __dict__ as before)
    __slots__ = ("_bound", "_target", "_event_type", "_stop_after", "__dict__")
    _generated = _Stat(lambda st, i: int(st["accepted"][i] + st["dropped"][i]))     # Requests handed out

    def __init__(self, target: Entity, event_type: str = "Request", stop_after: Instant | None = None,
                 context_fn=None):
        if context_fn is not None:
            raise NotImplementedError("context_fn is arbitrary Python and is not lowered to the engine")
        self._bound = None                  # (set first: _Stat's setter reads it; the same keys in the same order keep the instance dicts shared)
        self._target = target
        self._event_type = event_type
        self._stop_after = stop_after
        self._generated = 0


class _ArrivalProvider:
    kind = "constant"
    __slots__ = ("profile",)

    def __init__(self, profile, start_time: Instant = None):
        if not isinstance(profile, (ConstantRateProfile, LinearRampProfile, SpikeProfile)):
            raise NotImplementedError(
                f"profile {type(profile).__name__} is arbitrary Python; the engine lowers ConstantRateProfile, "
                "LinearRampProfile and SpikeProfile (load/profile.py)")
        self.profile = profile


class ConstantArrivalTimeProvider(_ArrivalProvider):
    kind = "constant"
    __slots__ = ()


class PoissonArrivalTimeProvider(_ArrivalProvider):
    kind = "poisson"
    __slots__ = ()


class Source(Entity):
    __slots__ = ("_bound", "_event_provider", "_time_provider")
    _generated_count = _Stat("generated")

    def __init__(self, name: str, event_provider: SimpleEventProvider, arrival_time_provider: _ArrivalProvider):
        super().__init__(name)
        self._bound = None
        self._event_provider = event_provider
        self._time_provider = arrival_time_provider
        self._generated_count = 0

    @classmethod
    def _make(cls, provider_cls, rate, target, event_type, name, stop_after, event_provider):
        if event_provider is None:
            if target is None:
                raise ValueError("Either 'target' or 'event_provider' must be provided")
            event_provider = SimpleEventProvider(target, event_type, cls._resolve_stop_after(stop_after))
        return cls(name=name, event_provider=event_provider,
                   arrival_time_provider=provider_cls(ConstantRateProfile(rate=rate), start_time=Instant.Epoch))

    @classmethod
    def constant(cls, rate: float, target: Entity | None = None, event_type: str = "Request", *,
                 name: str = "Source", stop_after=None, event_provider=None) -> "Source":
        return cls._make(ConstantArrivalTimeProvider, rate, target, event_type, name, stop_after, event_provider)

    @classmethod
    def poisson(cls, rate: float, target: Entity | None = None, event_type: str = "Request", *,
                name: str = "Source", stop_after=None, event_provider=None) -> "Source":
        return cls._make(PoissonArrivalTimeProvider, rate, target, event_type, name, stop_after, event_provider)

    @classmethod
    def with_profile(cls, profile, target: Entity | None = None, event_type: str = "Request", *, poisson: bool = True,
                     name: str = "Source", stop_after=None, event_provider=None) -> "Source":
        """A Source whose arrival rate follows `profile` (load/source.py:271-320)."""
        if event_provider is None:
            if target is None:
                raise ValueError("Either 'target' or 'event_provider' must be provided")
            event_provider = SimpleEventProvider(target, event_type, cls._resolve_stop_after(stop_after))
        provider_cls = PoissonArrivalTimeProvider if poisson else ConstantArrivalTimeProvider
        return cls(name=name, event_provider=event_provider, arrival_time_provider=provider_cls(profile, start_time=Instant.Epoch))

    @staticmethod
    def _resolve_stop_after(stop_after):
        if stop_after is None:
            return None
        if isinstance(stop_after, Instant):
            return stop_after
        return Instant.from_seconds(stop_after)

    @property
    def generated_count(self) -> int:
        return self._generated_count

    @property
    def rate(self) -> float:
        """The profile's rate; for a time-varying profile its peak (what sizes the engine's record logs)."""
        return self._time_provider.profile.peak_rate

    def downstream_entities(self) -> list[Entity]:
        t = getattr(self._event_provider, "_target", None)
        return [t] if isinstance(t, Entity) else []

    def __repr__(self):
        return f"<Source {self.name}>"


# ---- server --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ServerStats:
    requests_completed: int = 0
    requests_rejected: int = 0
    total_service_time: float = 0.0


class _QueueView:
    """What users read off `server.queue`: acceptance / drop counters and depth."""
    _bound = None
    stats_accepted = _Stat("accepted")
    stats_dropped = _Stat("dropped")
    depth = _Stat("queue_depth")

    def __init__(self):
        self._bound = None
        self.stats_accepted = 0
        self.stats_dropped = 0
        self.depth = 0


class Server(Entity):
    __slots__ = ("_bound", "_policy", "_concurrency", "_service_time", "_downstream", "_queue")
    _requests_completed = _Stat("completed")
    _requests_rejected = _Stat("rejected")
    _total_service_time = _Stat("total_service_s", float, 0.0)
    _active = _Stat("active")

    def __init__(self, name: str, concurrency: int = 1, service_time: LatencyDistribution | None = None,
                 queue_policy: FIFOQueue | None = None, queue_capacity: int | None = None,
                 downstream: Entity | None = None):
        super().__init__(name)
        if not isinstance(concurrency, int):
            raise NotImplementedError("only FixedConcurrency (an int) is lowered to the engine")
        if concurrency < 1:
            raise ValueError(f"max_concurrent must be >= 1, got {concurrency}")   # server/concurrency.py:87-88
        if queue_policy is None:
            queue_policy = FIFOQueue(capacity=queue_capacity if queue_capacity is not None else float("inf"))
        elif not isinstance(queue_policy, FIFOQueue):
            raise NotImplementedError("only FIFOQueue is lowered to the engine")
        self._bound = None
        self._policy = queue_policy
        self._concurrency = concurrency
        self._service_time = service_time or ConstantLatency(0.01)
        self._downstream = downstream
        self._queue = _QueueView()
        self._requests_completed = 0
        self._requests_rejected = 0
        self._total_service_time = 0.0
        self._active = 0

    def downstream_entities(self) -> list[Entity]:
        return [self._downstream] if self._downstream is not None else []

    @property
    def downstream(self):
        return self._downstream

    @downstream.setter
    def downstream(self, target):
        self._downstream = target

    @property
    def concurrency(self) -> int:
        return self._concurrency

    @property
    def service_time(self) -> LatencyDistribution:
        return self._service_time

    @property
    def queue(self) -> _QueueView:
        return self._queue

    @property
    def depth(self) -> int:
        return self._queue.depth

    @property
    def stats_accepted(self) -> int:
        return self._queue.stats_accepted

    @property
    def stats_dropped(self) -> int:
        return self._queue.stats_dropped

    @property
    def active_requests(self) -> int:
        return self._active

    @property
    def utilization(self) -> float:
        return self._active / self._concurrency if self._concurrency else 0.0

    @property
    def average_service_time(self) -> float:
        return self._total_service_time / self._requests_completed if self._requests_completed else 0.0

    @property
    def stats(self) -> ServerStats:
        return ServerStats(self._requests_completed, self._requests_rejected, self._total_service_time)

    def has_capacity(self, weight: int = 1) -> bool:
        return self._active < self._concurrency


# ---- sinks ---------------------------------------------------------------------------------------
def _percentile_sorted(sorted_values, p: float) -> float:
    """instrumentation/data.py:197-210."""
    n = len(sorted_values)
    if n == 0:
        return 0.0
    if p <= 0:
        return float(sorted_values[0])
    if p >= 1:
        return float(sorted_values[-1])
    pos = p * (n - 1)
    lo = int(pos)
    hi = min(lo + 1, n - 1)
    frac = pos - lo
    return float(sorted_values[lo] * (1.0 - frac) + sorted_values[hi] * frac)


_DEVICE_STATS_MIN = 4096      # Sink.latency_stats(): records from which the device routine is used


class _RecordSink(Entity):
    """Shared result holder: the engine hands back (completion ns, created_at ns) arrays; the Python lists the
    reference exposes are materialised lazily (31 M-element lists are the user's choice, not ours)."""

    _EMPTY = np.zeros(0, np.int64)

    def __init__(self, name: str):
        super().__init__(name)
        self._rec_t = self._EMPTY
        self._rec_cr = self._EMPTY
        self._lazy = None                   # (LazyRecords, station): the run's records are still on the device (lowering.py)
        self._device = 0

    def _set_records(self, t_ns: np.ndarray, created_ns: np.ndarray):
        if _PENDING:
            _flush_pending()
        self._rec_t = t_ns
        self._rec_cr = created_ns
        self._lazy = None

    def _bind_lazy(self, records, station: int, device: int = 0):
        self._lazy = (records, station)
        self._device = device

    def _materialise(self):
        if _PENDING:
            _resolve(self)
        if self._lazy is not None:
            records, i = self._lazy
            self._rec_t, self._rec_cr = records.records(i)
            self._lazy = None

    @property
    def _t_ns(self) -> np.ndarray:
        self._materialise()
        return self._rec_t

    @property
    def _created_ns(self) -> np.ndarray:
        self._materialise()
        return self._rec_cr

    def _n_records(self) -> int:
        if _PENDING:
            _resolve(self)
        return self._lazy[0].count(self._lazy[1]) if self._lazy is not None else int(len(self._rec_t))

    @property
    def completion_ns(self) -> np.ndarray:
        return self._t_ns

    @property
    def latencies_array(self) -> np.ndarray:
        # (event.time - created_at).to_seconds() = float(ns) / 1e9   (components/common.py:39-40)
        return (self._t_ns - self._created_ns).astype(np.float64) / 1_000_000_000


class Sink(_RecordSink):
    def __init__(self, name: str = "Sink"):
        super().__init__(name)

    @property
    def events_received(self) -> int:
        return self._n_records()

    @property
    def completion_times(self) -> list[Instant]:
        return [Instant(int(t)) for t in self._t_ns]

    @property
    def latencies_s(self) -> list[float]:
        return self.latencies_array.tolist()

    def average_latency(self) -> float:
        lat = self.latencies_s
        return sum(lat) / len(lat) if lat else 0.0

    def latency_time_series_seconds(self):
        return [t.to_seconds() for t in self.completion_times], list(self.latencies_s)

    def latency_stats(self) -> dict:
        dev = getattr(self, "_device_latency_stats", None)
        if dev is not None:                      # a load balancer's shared Sink: computed by the engine (hs_lb_latency_stats)
            return dict(dev)
        n = len(self._t_ns)
        if n >= _DEVICE_STATS_MIN:               # a large record set stays numeric: sort + sum + percentiles on the device
            from . import _native as N

            if N.lib().hs_device_count() > 0:
                import ctypes as C

                t = np.ascontiguousarray(self._t_ns, np.int64)
                cr = np.ascontiguousarray(self._created_ns, np.int64)
                out = (C.c_double * 6)()
                rc = N.lib().hs_sink_latency_stats(getattr(self, "_device", 0), n, t.ctypes.data, cr.ctypes.data, out)
                if rc != N.HS_OK:
                    raise N.EngineError(rc, (N.lib().hs_lb_last_error(None) or b"").decode())
                return {"count": int(out[0]), "avg": out[1], "min": out[2], "max": out[3], "p50": out[4], "p99": out[5]}
        lat = self.latencies_s
        if n == 0:
            return {"count": 0, "avg": 0.0, "min": 0.0, "max": 0.0, "p50": 0.0, "p99": 0.0}
        s = sorted(lat)
        return {"count": n, "avg": sum(s) / n, "min": s[0], "max": s[-1],
                "p50": _percentile_sorted(s, 0.50), "p99": _percentile_sorted(s, 0.99)}


class Counter(_RecordSink):
    def __init__(self, name: str = "Counter"):
        super().__init__(name)
        self._event_type = "Request"

    @property
    def total(self) -> int:
        return self._n_records()

    @property
    def by_type(self) -> dict:
        return {self._event_type: self.total} if self.total else {}


class LatencyTracker(_RecordSink):
    def __init__(self, name: str = "LatencyTracker"):
        super().__init__(name)

    @property
    def count(self) -> int:
        return self._n_records()

    def mean_latency(self) -> float:
        lat = self.latencies_array
        return float(sum(lat.tolist()) / len(lat)) if len(lat) else 0.0

    def _pct(self, p):
        return _percentile_sorted(sorted(self.latencies_array.tolist()), p)

    def p50(self) -> float:
        return self._pct(0.50)

    def p99(self) -> float:
        return self._pct(0.99)


# ---- networks of stations ------------------------------------------------------------------------
@dataclass(frozen=True)
class NetworkLinkStats:
    bytes_transmitted: int = 0
    packets_sent: int = 0
    packets_dropped: int = 0


class NetworkLink(Entity):
    """Point-to-point link (components/network/link.py:36-234).  Lowered: a constant base latency > 0 (it is the
    lookahead of the conservative windows) plus optional jitter -- ExponentialLatency (one draw per packet) or ConstantLatency (a
    constant on top, as the reference's datacenter_network preset has it) --, towards a Server.  `packet_loss_rate` is
    lowered with the link's own Philox LOSS stream in place of the process-wide `random.random()` (link.py:131).
    `bandwidth_bps` is accepted: requests of the lowered event providers carry no payload_size, so their transmission
    time is 0 s and `bytes_transmitted` stays 0, as in the reference (link.py:209-234)."""

    def __init__(self, name: str, latency: LatencyDistribution, bandwidth_bps: float | None = None,
                 packet_loss_rate: float = 0.0, jitter: LatencyDistribution | None = None, egress: Entity | None = None):
        super().__init__(name)
        if packet_loss_rate < 0.0 or packet_loss_rate > 1.0:
            raise ValueError(f"packet_loss_rate must be in [0, 1], got {packet_loss_rate}")   # link.py:71-72
        self.latency = latency
        self.bandwidth_bps = bandwidth_bps
        self.packet_loss_rate = packet_loss_rate
        self.jitter = jitter
        self.egress = egress
        self.bytes_transmitted = 0
        self.packets_sent = 0
        self.packets_dropped = 0
        self._entered = 0

    def downstream_entities(self) -> list[Entity]:
        return [self.egress] if self.egress is not None else []

    @property
    def link_stats(self) -> NetworkLinkStats:
        return NetworkLinkStats(self.bytes_transmitted, self.packets_sent, self.packets_dropped)

    @property
    def current_utilization(self) -> float:
        return 0.0          # bandwidth is infinite on the lowered path (link.py:93-95)


# ---- network condition presets (components/network/conditions.py:13-258): the same nine links, from one table ----------------
_NETWORK_PRESETS = {
    # name of the function: (default link name, base latency s, bandwidth bit/s, packet loss, jitter kind, jitter mean s)
    "local_network": ("local", 0.0001, 1_000_000_000, 0.0, None, 0.0),
    "datacenter_network": ("datacenter", 0.0005, 10_000_000_000, 0.0, "const", 0.0001),
    "cross_region_network": ("cross_region", 0.050, 1_000_000_000, 0.0001, "exp", 0.005),
    "internet_network": ("internet", 0.100, 100_000_000, 0.001, "exp", 0.020),
    "satellite_network": ("satellite", 0.600, 10_000_000, 0.005, "exp", 0.050),
    "mobile_3g_network": ("mobile_3g", 0.100, 2_000_000, 0.005, "exp", 0.030),
    "mobile_4g_network": ("mobile_4g", 0.050, 20_000_000, 0.001, "exp", 0.015),
}


def _preset_link(key: str, name: str | None = None) -> "NetworkLink":
    dflt, lat, bw, loss, jk, jm = _NETWORK_PRESETS[key]
    jitter = None if jk is None else ConstantLatency(jm) if jk == "const" else ExponentialLatency(jm)
    return NetworkLink(name=dflt if name is None else name, latency=ConstantLatency(lat), bandwidth_bps=bw, packet_loss_rate=loss,
                       jitter=jitter)


def local_network(name: str = "local"): return _preset_link("local_network", name)
def datacenter_network(name: str = "datacenter"): return _preset_link("datacenter_network", name)
def cross_region_network(name: str = "cross_region"): return _preset_link("cross_region_network", name)
def internet_network(name: str = "internet"): return _preset_link("internet_network", name)
def satellite_network(name: str = "satellite"): return _preset_link("satellite_network", name)
def mobile_3g_network(name: str = "mobile_3g"): return _preset_link("mobile_3g_network", name)
def mobile_4g_network(name: str = "mobile_4g"): return _preset_link("mobile_4g_network", name)


def lossy_network(loss_rate: float, name: str = "lossy", base_latency: float = 0.010):
    """conditions.py:148-178: a 100 Mbit/s link that drops `loss_rate` of its packets."""
    if loss_rate < 0.0 or loss_rate > 1.0:
        raise ValueError(f"loss_rate must be in [0, 1], got {loss_rate}")
    return NetworkLink(name=name, latency=ConstantLatency(base_latency), bandwidth_bps=100_000_000, packet_loss_rate=loss_rate, jitter=None)


def slow_network(latency_seconds: float, name: str = "slow", bandwidth_bps: float = 1_000_000):
    """conditions.py:181-204."""
    return NetworkLink(name=name, latency=ConstantLatency(latency_seconds), bandwidth_bps=bandwidth_bps, packet_loss_rate=0.0, jitter=None)


class RandomRouter(Entity):
    """Uniform random fan-out (components/random_router.py:9-45); the engine draws the target from the router's own
    Philox stream: index = int(u * len(targets))."""

    def __init__(self, name: str, *, targets: list[Entity]):
        super().__init__(name)
        self.targets = targets
        self.stats_routed = 0
        self.target_counts: dict[str, int] = {}

    def downstream_entities(self) -> list[Entity]:
        return list(self.targets)


# ---- load balancing ------------------------------------------------------------------------------
class ClientKeyEventProvider:
    """One Request per tick whose metadata carries a random `client_id` for key-based strategies -- the request factory
    of examples/visual/chash_example.py:69-88 (`ClientRequestProvider`: `client_id = rng.randint(0, NUM_CLIENTS - 1)`).
    On the engine the id is `int(u * n_clients)` with u from the source's own Philox KEY stream, and the routing key is
    its decimal string, as `ConsistentHash._default_get_key` would return it (strategies.py:369-375)."""

    def __init__(self, target: Entity, n_clients: int, stop_after: Instant | float | None = None,
                 event_type: str = "Request"):
        if n_clients < 1:
            raise ValueError(f"n_clients must be >= 1, got {n_clients}")
        self._target = target
        self._n_clients = int(n_clients)
        self._event_type = event_type
        self._stop_after = Source._resolve_stop_after(stop_after)
        self._generated = 0


class ConsistentHash:
    """Consistent hashing with virtual nodes (components/load_balancer/strategies.py:336-433): every backend owns
    `virtual_nodes` points md5("<name>:<i>") on a ring; a request goes to the first point whose hash is >= md5(key).
    The ring is built (and searched) inside libhs_hip.so; `ring_backends` / `select_name` expose it for inspection."""

    def __init__(self, virtual_nodes: int = 100, get_key=None):
        if virtual_nodes < 1:
            raise ValueError(f"virtual_nodes must be >= 1, got {virtual_nodes}")        # strategies.py:356-357
        if get_key is not None:
            raise NotImplementedError("a custom get_key is arbitrary Python; the engine hashes metadata['client_id']")
        self._virtual_nodes = int(virtual_nodes)
        self._fallback = RoundRobin()       # what a Request without a key gets (strategies.py:362,420-421): only the single-heap path runs those

    @property
    def virtual_nodes(self) -> int:
        return self._virtual_nodes


class RoundRobin:
    """Cycles through the backends (components/load_balancer/strategies.py:50-73), the LoadBalancer's default strategy: the k-th
    Request the LoadBalancer processes goes to backends[k % len(backends)].  On the engine the Requests of all Sources are ranked
    by arrival on the device (csrc/hs_lb.hip hs_lb_rr_assign); `_index` holds the number of selections after a run."""

    def __init__(self):
        self._index = 0

    def reset(self) -> None:
        self._index = 0


class Random:
    """Random backend selection (strategies.py:137-150: `random.choice(backends)`).  On the engine the choice is
    backends[int(u * len(backends))] with u the Request's draw from its Source's own Philox KEY stream -- the seed-matched form
    of the process-wide `random`, pinned against the reference with that plug (tests/golden/make_golden.py
    `_PerRequestChoice`)."""


@dataclass(frozen=True)
class LoadBalancerStats:
    """components/load_balancer/load_balancer.py:40-58."""

    requests_received: int = 0
    requests_forwarded: int = 0
    requests_failed: int = 0
    no_backend_available: int = 0
    backends_marked_unhealthy: int = 0
    backends_marked_healthy: int = 0


@dataclass
class BackendInfo:
    """load_balancer.py:61-80 (health bookkeeping is host-side state the engine never changes)."""

    backend: Entity
    weight: int = 1
    is_healthy: bool = True
    consecutive_failures: int = 0
    consecutive_successes: int = 0
    total_requests: int = 0
    total_failures: int = 0


class LoadBalancer(Entity):
    """Distributes requests over backends (components/load_balancer/load_balancer.py:83-473).  Lowered: the ConsistentHash,
    RoundRobin (the default) and Random strategies over Server backends that all stay healthy for the whole run."""

    def __init__(self, name: str, backends: list[Entity] | None = None, strategy=None, on_no_backend: str = "reject"):
        super().__init__(name)
        if on_no_backend not in ("reject", "queue"):
            raise ValueError(f"on_no_backend must be 'reject' or 'queue', got {on_no_backend}")   # load_balancer.py:120-121
        if strategy is None:
            strategy = RoundRobin()                                                       # load_balancer.py:112
        if not isinstance(strategy, (ConsistentHash, RoundRobin, Random)):
            raise NotImplementedError(f"strategy {type(strategy).__name__} is not lowered to the engine (ConsistentHash, "
                                      "RoundRobin and Random are)")
        self._strategy = strategy
        self._on_no_backend = on_no_backend
        self._backends: dict[str, BackendInfo] = {}
        for b in backends or []:
            self.add_backend(b)
        self._requests_received = 0
        self._requests_forwarded = 0
        self._requests_failed = 0
        self._no_backend_available = 0
        self._in_flight_count = 0

    def add_backend(self, backend: Entity, weight: int = 1) -> None:
        if weight < 1:
            raise ValueError(f"weight must be >= 1, got {weight}")                        # load_balancer.py:207-208
        if backend.name in self._backends:
            self._backends[backend.name].weight = weight
            return
        self._backends[backend.name] = BackendInfo(backend=backend, weight=weight)

    def remove_backend(self, backend: Entity) -> None:
        self._backends.pop(backend.name, None)

    @property
    def strategy(self):
        return self._strategy

    @property
    def all_backends(self) -> list[Entity]:
        return [i.backend for i in self._backends.values()]

    healthy_backends = all_backends

    @property
    def unhealthy_backends(self) -> list[Entity]:
        return []

    @property
    def backend_count(self) -> int:
        return len(self._backends)

    @property
    def healthy_count(self) -> int:
        return len(self._backends)

    def downstream_entities(self) -> list[Entity]:
        return self.all_backends

    def get_backend_info(self, backend: Entity) -> BackendInfo | None:
        return self._backends.get(backend.name)

    def get_backend_info_by_name(self, name: str) -> BackendInfo | None:
        return self._backends.get(name)

    @property
    def stats(self) -> LoadBalancerStats:
        return LoadBalancerStats(self._requests_received, self._requests_forwarded, self._requests_failed,
                                 self._no_backend_available, 0, 0)


# ---- probes --------------------------------------------------------------------------------------
class Data:
    """Time-series container of a Probe (instrumentation/data.py:20-110): `values` = [(time_s, value), ...]."""

    def __init__(self) -> None:
        self._tt = np.zeros(0, np.int64)
        self._vv = np.zeros(0, np.int64)
        self._scale = None           # utilisation: value / concurrency
        self._lazy = None            # () -> (t_ns, v): the samples are still on the device (lowering.write_back_plain_probes)

    def _set(self, t_ns: np.ndarray, v: np.ndarray, scale=None) -> None:
        self._tt, self._vv, self._scale, self._lazy = t_ns, v, scale, None

    def _set_lazy(self, fetch, scale=None) -> None:
        self._lazy, self._scale = fetch, scale

    def _load(self) -> None:
        if self._lazy is not None:
            fetch, self._lazy = self._lazy, None
            self._tt, self._vv = fetch()

    @property
    def _t_ns(self) -> np.ndarray:
        self._load()
        return self._tt

    @property
    def _v(self) -> np.ndarray:
        self._load()
        return self._vv

    def _val(self, v):
        if self._scale is None:
            return int(v)
        return self._scale(int(v)) if callable(self._scale) else int(v) / self._scale

    @property
    def values(self) -> list:
        return [(float(t) / 1_000_000_000, self._val(v)) for t, v in zip(self._t_ns.tolist(), self._v.tolist())]

    def times(self) -> list:
        return [float(t) / 1_000_000_000 for t in self._t_ns.tolist()]

    def raw_values(self) -> list:
        return [self._val(v) for v in self._v.tolist()]

    def count(self) -> int:
        return int(len(self._v))

    def mean(self) -> float:
        vals = self.raw_values()
        return sum(vals) / len(vals) if vals else 0.0

    def min(self) -> float:
        vals = self.raw_values()
        return min(vals) if vals else 0.0

    def max(self) -> float:
        vals = self.raw_values()
        return max(vals) if vals else 0.0

    def __len__(self) -> int:
        return self.count()


class Probe(Entity):
    """Periodic metric sampler (instrumentation/probe.py:81-164): a daemon Source that reads `getattr(target, metric)`
    every `interval` seconds into a Data container.  Lowered metrics: Server.depth / active_requests / utilization / available_capacity / has_capacity /
    stats_accepted / stats_dropped / requests completed, Sink.events_received, Source.generated_count."""

    _LOWERED = {"depth", "active_requests", "utilization", "available_capacity", "has_capacity", "stats_accepted", "stats_dropped",
                "requests_completed", "_requests_completed", "events_received", "generated_count", "_generated_count"}
    # Server attributes that are functions of `active_requests` and the (fixed) concurrency: the engine samples the integer,
    # the Data container applies the reference's expression (components/server/server.py:153-173,191-200, concurrency.py:117-131)
    _ON_ACTIVE = ("utilization", "available_capacity", "has_capacity")

    @classmethod
    def engine_metric(cls, metric: str) -> str:
        """The counter the engine samples for `metric`."""
        return "active_requests" if metric in cls._ON_ACTIVE else "generated_count" if metric == "_generated_count" else metric

    @classmethod
    def value_map(cls, metric: str, server):
        """What `Data` does with a sampled integer: None = keep, a number = divide by it, a callable = apply it."""
        if metric not in cls._ON_ACTIVE:
            return None
        c = int(server.concurrency)
        if metric == "utilization":
            return c                                        # active / limit
        if metric == "available_capacity":
            return lambda active: c - active                # FixedConcurrency.available
        return lambda active: active < c                    # has_capacity(): a callable attribute, the probe calls it (probe.py:52-55)

    def __init__(self, target: Entity, metric: str, data: Data, interval: float = 1.0, start_time: Instant | None = None):
        if interval <= 0:
            raise ValueError("Probe interval must be positive.")                  # probe.py:29-30
        if metric not in self._LOWERED:
            raise NotImplementedError(f"probe metric '{metric}' is an arbitrary attribute; lowered: {sorted(self._LOWERED)}")
        super().__init__(f"Probe_{target.name}_{metric}")
        self.target = target
        self.metric = metric
        self.data_sink = data
        self.interval = float(interval)
        # `start_time` only seeds the arrival-time provider, and Source.start() overwrites that with the Simulation's start time
        # before the first tick is drawn (load/source.py:120-127): the reference samples at start + k * interval whatever it is
        # (tests/test_oracle_live_reference.py::test_live_reference_ignores_a_probe_start_time)
        self.start_time = start_time if start_time is not None else Instant.Epoch

    @classmethod
    def on(cls, target: Entity, metric: str, interval: float = 1.0):
        data = Data()
        return cls(target=target, metric=metric, data=data, interval=interval), data

    @classmethod
    def on_many(cls, target: Entity, metrics: list, interval: float = 1.0):
        probes, data = [], {}
        for m in metrics:
            p, d = cls.on(target, m, interval=interval)
            probes.append(p)
            data[m] = d
        return probes, data
