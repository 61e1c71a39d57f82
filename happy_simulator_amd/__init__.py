"""happy_simulator_amd -- MI355X-native discrete-event engine behind the happy-simulator API.

Only the hot path lives here: `Simulation.run()`'s event loop and
`happysimulator.parallel`, re-designed as a GPU-resident engine (csrc/, C ABI in include/hs_engine.h)
with a host-side mirror of the reference's `Simulation / Source / Server / Sink / Instant` API:

    from happy_simulator_amd import Simulation, Source, Server, Sink, Instant, ExponentialLatency

    sink = Sink()
    server = Server("srv", service_time=ExponentialLatency(0.1), downstream=sink)
    source = Source.poisson(rate=8, target=server)
    summary = Simulation(end_time=Instant.from_seconds(60), sources=[source], entities=[server, sink]).run()
"""
from ._native import EngineError, EngineUnavailable  # noqa: F401
from .core.event import Event  # noqa: F401
from .core.temporal import Duration, Instant  # noqa: F401
from .entities import (BackendInfo, ClientKeyEventProvider, ConsistentHash, ConstantArrivalTimeProvider, Data, Probe,  # noqa: F401
                       ConstantLatency, ConstantRateProfile, Counter, Entity, ExponentialLatency, FIFOQueue,
                       LatencyTracker, LinearRampProfile, LoadBalancer, LoadBalancerStats, NetworkLink, NetworkLinkStats,
                       PoissonArrivalTimeProvider, Random, RandomRouter, RoundRobin, Server, ServerStats, SimpleEventProvider, Sink, Source,
                       SpikeProfile)
from .entities import (cross_region_network, datacenter_network, internet_network, local_network, lossy_network,  # noqa: F401
                       mobile_3g_network, mobile_4g_network, satellite_network, slow_network)
from .lowering import UnsupportedTopology  # noqa: F401
from .parallel import (ParallelResult, ParallelRunner, ParallelSimulation, ParallelSimulationSummary,  # noqa: F401
                       PartitionLink, RunConfig, SimulationPartition, reduce_summaries, shard_range)
from .simulation import Simulation, seed  # noqa: F401
from .summary import EntitySummary, QueueStats, SimulationSummary  # noqa: F401

__version__ = "0.1.0"
