"""ON THE GPU BOX: replicas of one general graph (the `graph_three_load_balancers` fixture's topology) through
ParallelRunner.run_replicas -- hs_graph_run_many, one workgroup per replica -- against the same graph run alone: events/s."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))

import graph_specs as GS  # noqa: E402
import happy_simulator_amd as hs  # noqa: E402
import helpers as H  # noqa: E402


SIDE = []


def main():
    spec = dict(H.Golden("graph_three_load_balancers").spec)
    spec["end_s"] = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    out = dict(what="general graph (3 LoadBalancers, 6 Servers, router, link; 32 Requests/s) x end_s on the single-heap path",
               end_s=spec["end_s"], runs=[])
    sim, _ = GS.build(spec)
    sim.run()                                                      # (first touch of the library)
    for n in (1, 64, 256, 1024, 4096):
        sims = []

        def build_fn():
            sims.append(GS.build(spec)[0])
            return sims[-1]

        t0 = time.monotonic()
        res = hs.ParallelRunner().run_replicas(build_fn, n, base_seed=7)
        dev_ms = sims[0]._engine_summary.last_run_ms            # the batch's device time (hs_graph_run_many)
        SIDE.append(dict(device_ms=round(dev_ms, 3), device_events_per_s=round(sum(r.summary.total_events_processed for r in res) / (dev_ms / 1e3), 1)))
        wall = time.monotonic() - t0
        ev = sum(r.summary.total_events_processed for r in res)
        out["runs"].append(dict(replicas=n, events=ev, wall_s=round(wall, 4), events_per_s=round(ev / wall, 1), **SIDE.pop()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
