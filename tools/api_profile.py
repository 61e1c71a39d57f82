#!/usr/bin/env python3
"""Where `hs.Simulation(...).run()` of the 65 536-chain grid spends its host time (cProfile), and the first reads after it."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import happy_simulator_amd as hs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def build():
    sinks = [hs.Sink(f"sink{i}") for i in range(n)]
    servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sinks[i]) for i in range(n)]
    sources = [hs.Source.poisson(rate=8.0, target=servers[i], name=f"src{i}") for i in range(n)]
    return sinks, servers, sources, hs.Simulation(end_time=hs.Instant.from_seconds(60.0), sources=sources,
                                                  entities=[e for pair in zip(servers, sinks) for e in pair], seed=42)


for rep in range(2):           # the second pass is the warm one
    sinks, servers, sources, sim = build()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    sim.run()
    pr.disable()
    t1 = time.perf_counter()
    pr2 = cProfile.Profile()
    pr2.enable()
    done = servers[n // 2].stats.requests_completed
    pr2.disable()
    t2 = time.perf_counter()
    pr3 = cProfile.Profile()
    pr3.enable()
    lat = sinks[n // 2].latencies_s
    pr3.disable()
    t3 = time.perf_counter()
    print(f"pass {rep}: run {1e3 * (t1 - t0):.1f} ms, first counter read {1e3 * (t2 - t1):.1f} ms, first sink read {1e3 * (t3 - t2):.1f} ms ({len(lat)} records)")
    if rep == 1:
        for name, p in (("run", pr), ("first counter read", pr2), ("first sink read", pr3)):
            print(f"==== {name}")
            pstats.Stats(p).sort_stats("cumulative").print_stats(18)
