#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path (SURVEY.md 8(d)(i),(ii)) in the build container and write the result to
profiles/<tag>_reference_python.json, which bench.py quotes beside the GPU number as `cpu_baseline.reference_python`
(labelled with the hardware it was measured on: /root/reference does not exist on the GPU box).

  (i)  stock `Simulation.run()`, one process, one core: N Source.poisson(8) -> Server(Exp 0.1) -> Sink chains in ONE
       Simulation (the headline grid's shape, stock MT19937 streams), a bounded horizon -- events/s is horizon-independent
       in steady state;
  (ii) `ParallelRunner(max_workers=nproc).run_replicas` over single-chain Simulations (BASELINE configs[1]'s shape).

    python tools/measure_reference_python.py r02
"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import refshim  # noqa: E402,F401  (import-time PEP 695 rewriter; nothing is copied)

refshim.install() if hasattr(refshim, "install") else None
import happysimulator as hs  # noqa: E402
from happysimulator import Instant  # noqa: E402


def build_grid(n, end_s):
    sinks = [hs.Sink(f"sink{i}") for i in range(n)]
    servers = [hs.Server(f"srv{i}", service_time=hs.ExponentialLatency(0.1), downstream=sinks[i]) for i in range(n)]
    sources = [hs.Source.poisson(rate=8, target=servers[i], name=f"src{i}") for i in range(n)]
    return hs.Simulation(end_time=Instant.from_seconds(end_s), sources=sources, entities=servers + sinks)


def build_one():
    return build_grid(1, 60.0)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main(tag):
    import random

    import numpy as np
    import subprocess
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = None
    out = {"python": platform.python_version(), "cpu": cpu_model(), "cores_available": os.cpu_count(),
           "where": "build container, NOT the GPU box (the GPU box has no /root/reference and the reference must not be copied "
                    "into the repo, so a same-box figure is not obtainable)",
           "host": platform.node(), "date": time.strftime("%Y-%m-%d"), "repo_head": head,
           "reference": "adamfilli/happy-simulator v0.2.5"}
    # (i) one Simulation, one core
    for n, end_s in ((4096, 2.0), (65536, 0.25)):
        random.seed(42); np.random.seed(42)
        t0 = time.perf_counter()
        sim = build_grid(n, end_s)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        s = sim.run()
        dt = time.perf_counter() - t0
        out[f"single_process_{n}_chains"] = {
            "value": s.total_events_processed / dt, "unit": "events/s", "cores": 1,
            "sample": f"{n} chains in one Simulation, {end_s:g} s simulated, {s.total_events_processed} events in {dt:.2f} s "
                      f"(+ {t_build:.2f} s to construct the entities)"}
        print(n, out[f"single_process_{n}_chains"], flush=True)
    # (ii) ParallelRunner, all cores
    from happysimulator.parallel import ParallelRunner

    nproc = os.cpu_count() or 1
    reps = 4 * nproc
    t0 = time.perf_counter()
    res = ParallelRunner(max_workers=nproc).run_replicas(build_one, reps, base_seed=42)
    dt = time.perf_counter() - t0
    ev = sum(r.summary.total_events_processed for r in res)
    out["parallel_runner"] = {"value": ev / dt, "unit": "events/s", "cores": nproc,
                              "sample": f"ParallelRunner(max_workers={nproc}).run_replicas: {reps} single-chain replicas x 60 s, "
                                        f"{ev} events in {dt:.2f} s (process start-up included)"}
    print(out["parallel_runner"], flush=True)
    path = os.path.join(ROOT, "profiles", f"{tag}_reference_python.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
