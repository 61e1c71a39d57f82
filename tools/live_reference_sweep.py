"""Build container only (needs /root/reference): the LIVE reference against the oracle on the case ranges that
tools/gpu_random_sweep.py ran on MI355X, so that "engine == oracle" there also means "engine == reference".  Same generators
(tests/random_specs.py), same checkers as tests/test_oracle_live_reference.py (every count, statistic, Sink record, probe sample
and the full processed-event trace with sort indices).

    python tools/live_reference_sweep.py --first 1000 --count 400 [--jobs 8] > profiles/rNN_live_reference_sweep.log
"""
import argparse
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)

FAMILIES = ("station", "tie", "multi_source", "ring", "multi_source_ring", "jitter_ring", "graph", "lb", "lb_probes", "lb_profiles", "lb_strategies", "lb_workers",
            "tandem", "tandem_probes", "tandem_fan_in")


def one(job):
    fam, k = job
    import helpers as H
    import make_golden as MG
    import random_specs as RS
    from test_oracle_golden import (check_oracle_against_lb_golden, check_oracle_against_ring_golden,
                                    check_oracle_against_station_golden, check_oracle_against_tandem_golden)
    try:
        if fam == "tandem_fan_in":
            import tandem_specs as TS
            from test_oracle_golden import check_oracle_against_fan_in_reference

            case = TS.fan_in_case(k)
            check_oracle_against_fan_in_reference(case, MG.run_fan_in_case(case))
            return fam, k, ""
        if fam in ("tandem", "tandem_probes"):
            import tandem_specs as TS

            out, meta = MG.run_tandem_case(TS.tandem_spec(k) if fam == "tandem" else TS.tandem_probe_case(k))
            check_oracle_against_tandem_golden(H.Golden.from_results(out, meta))
            return fam, k, ""
        if fam in ("station", "tie", "multi_source"):
            spec = {"station": RS.station_spec, "tie": RS.tie_spec, "multi_source": RS.multi_source_spec}[fam](k)
            if spec["mode"] == "replicas":
                spec["trace"] = False
            out, meta = MG.run_case(spec)
            check_oracle_against_station_golden(H.Golden.from_results(out, meta))
        elif fam == "graph":
            from test_oracle_golden import check_oracle_against_graph_golden

            out, meta = MG.run_graph_case(RS.graph_spec(k))
            check_oracle_against_graph_golden(H.Golden.from_results(out, meta))
        elif fam in ("ring", "multi_source_ring", "jitter_ring"):
            spec = {"ring": RS.ring_spec, "multi_source_ring": RS.multi_source_ring_spec, "jitter_ring": RS.jitter_ring_spec}[fam](k)
            out, meta = MG.run_ring_case(spec)
            check_oracle_against_ring_golden(H.Golden.from_results(out, meta))
        else:
            spec = {"lb": RS.lb_spec, "lb_probes": RS.lb_probe_spec, "lb_profiles": RS.lb_profile_spec, "lb_strategies": RS.lb_strategy_spec,
                    "lb_workers": RS.lb_workers_spec}[fam](k)
            out, meta = MG.run_lb_case(spec)
            check_oracle_against_lb_golden(H.Golden.from_results(out, meta))
        return fam, k, ""
    except Exception as e:  # noqa: BLE001 -- a sweep reports and goes on
        return fam, k, f"{type(e).__name__}: " + " | ".join(str(e).strip().splitlines()[:4])[:300]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--count", type=int, default=400)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--families", default="")
    a = ap.parse_args()
    fams = [f for f in FAMILIES if not a.families or f in a.families.split(",")]
    jobs = [(f, a.first + i) for i in range(a.count) for f in fams]
    t0 = time.time()
    tally = {f: [0, 0] for f in fams}
    with ProcessPoolExecutor(a.jobs) as ex:
        for fam, k, err in ex.map(one, jobs, chunksize=8):
            tally[fam][0] += 1
            if err:
                tally[fam][1] += 1
                print(f"DIFF {fam} {k}: {err}", flush=True)
    print(f"{'family':24s} {'run':>6s} {'differ':>7s}")
    for f, (n, d) in tally.items():
        print(f"{f:24s} {n:6d} {d:7d}")
    print(f"cases {a.first}..{a.first + a.count - 1}, live reference vs oracle, {time.time() - t0:.0f} s; differences: {sum(d for _, d in tally.values())}")
    return 1 if any(d for _, d in tally.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
