"""ON THE GPU BOX: ONE Simulation of many independent chains outside the station shape -- five Poisson Sources -> Server(c = 40) ->
Sink, the station engines stop at four Sources and c = 32 -- through hs.Simulation(...).run(): the graph falls into parts, 2 048 heaps
side by side (hs_graph_run_parts).  Wall time by phase and the device time."""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import happy_simulator_amd as hs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    end_s = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    per = 5
    t0 = time.monotonic()
    sinks = [hs.Sink(f"k{i}") for i in range(n)]
    servers = [hs.Server(f"s{i}", concurrency=40, service_time=hs.ExponentialLatency(0.1), downstream=sinks[i]) for i in range(n)]
    sources = [hs.Source.poisson(rate=1.6, target=servers[k // per], name=f"src{k}") for k in range(n * per)]
    sim = hs.Simulation(end_time=hs.Instant.from_seconds(end_s), sources=sources, entities=servers + sinks, seed=42)
    t1 = time.monotonic()
    g = sim.lowered()
    t2 = time.monotonic()
    summary = sim.run()
    t3 = time.monotonic()
    dev_ms = float(sim._engine_summary.last_run_ms)
    print(json.dumps(dict(
        what=f"{n} chains of five Poisson Sources (1.6/s each) -> Server(c=40, Exp 0.1) -> Sink, {end_s:g} s, ONE hs.Simulation on the single-heap "
             "path: the graph's parts side by side (hs_graph_run_parts)",
        chains=n, nodes=int(g.arrays.n), parts=sim._graph_parts, events=summary.total_events_processed,
        construct_s=round(t1 - t0, 3), lower_s=round(t2 - t1, 3), run_s=round(t3 - t2, 3), device_ms=round(dev_ms, 3),
        events_per_s_device=round(summary.total_events_processed / (dev_ms / 1e3), 1),
        events_per_s_run=round(summary.total_events_processed / (t3 - t2), 1))))


if __name__ == "__main__":
    main()
