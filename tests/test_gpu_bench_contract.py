"""GPU: the driver-facing contract of bench.py -- one JSON line on stdout with the metric, the whole-job value, the
roofline object and the CPU baseline -- on small instances of the three workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]              # exactly ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("args", [
    ("--n-lp", "4096", "--end-s", "20", "--cpu-sample-s", "2", "--extras", "0"),
    ("--workload", "ring", "--n-lp", "2048", "--end-s", "5", "--cpu-sample-s", "2"),
    ("--workload", "lb", "--lb-sources", "512", "--lb-backends", "512", "--end-s", "10", "--cpu-sample-s", "2"),
], ids=["grid", "ring", "lb"])
def test_bench_prints_one_contract_line(args):
    d = _run("--steps", "3", "--warmup", "1", *args)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "events/s" and d["value"] > 1e6 and d["ms_per_step"] > 0
    assert d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if r["bound"] == "valu":                 # grid / ring: the HBM figures of the contract + the measured VALU issue fraction
        assert "valu" in r and (r["valu"] is None or 0.0 < r["valu"]["busy_frac"] <= 1.0)
        # the binding resource and SURVEY 8(d)'s own model as FIELDS (valu_frac is None away from the profiled size)
        assert "valu_frac" in r and (r["valu_frac"] is None or 0.0 < r["valu_frac"] <= 1.0) and r["frac_survey_8d"] > r["frac"]
        # the class-weighted issue floor (None away from the profiled size) and SURVEY 8(d)'s second denominator
        assert "valu_floor_frac" in r and (r["valu_floor_frac"] is None or 0.0 < r["valu_floor_frac"] <= 1.0)
        assert abs(r["frac_of_measured_stream"] - r["achieved"] / 6290.0) < 1e-12
    if "--workload" not in args:             # the grid line also carries the API run and the reference's own Python path
        assert "api_run_s" in d["config"] and d["config"]["api_events"] == d["config"]["events_per_step_per_gpu"]
        # the reference's own Python loop was timed on ANOTHER host (the GPU box has no /root/reference): labelled as such
        assert "reference_python" not in d["cpu_baseline"]
        ref = d["cpu_baseline"].get("reference_python_other_host")
        assert ref is None or (ref["single_process_65536_chains"]["value"] > 1e3 and
                               d["cpu_baseline"]["reference_python_host"]["cpu"] == ref["cpu"])
        assert d["cpu_baseline"]["same_box_host"]["cores_available"] >= 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e4 and c["unit"] == "events/s" and c["sample"]
    assert d["value"] > c["value"]


def test_bench_strong_scaling_and_self_launch_flags():
    """`--scaling strong` keeps the metric's total station count (here on one GPU: the same numbers as weak), and a plain
    `python bench.py --gpus N` without a launcher starts its own ranks (checked with N = 1: no torch.distributed.run child)."""
    weak = _run("--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "5", "--cpu-sample-s", "0", "--api-run", "0", "--extras", "0")
    strong = _run("--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "5", "--cpu-sample-s", "0", "--api-run", "0",
                  "--scaling", "strong", "--extras", "0")
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert strong["config"]["events_per_step_per_gpu"] == weak["config"]["events_per_step_per_gpu"]


def test_default_line_carries_ring_lb_and_the_strong_shard():
    """The default invocation (grid, one GPU) times the ring, the load balancer and the 8-GPU strong shard with the same
    --steps / --warmup and prints them inside the ONE line -- each workload with its own roofline and CPU-port baseline."""
    d = _run("--steps", "3", "--warmup", "1", "--n-lp", "4096", "--end-s", "10", "--cpu-sample-s", "2", "--lb-sources", "512",
             "--lb-backends", "512", "--api-run", "0")
    for name, metric in (("ring", "ring network"), ("lb", "load balancer")):
        w = d["workloads"][name]
        assert metric in w["metric"] and w["value"] > 1e6 and w["ms_per_step"] > 0 and w["steps"] == 3 and w["warmup"] == 1
        assert w["roofline"]["achieved"] > 0 and w["roofline"]["peak"] == 8000.0
        assert w["cpu_baseline"]["kind"] == "port" and w["cpu_baseline"]["value"] > 1e4
    sh = d["strong_shard"]
    assert sh["n_lp"] == 512 and 0 < sh["kernel_ms_avg"] <= sh["ms_per_step"]
    brief = d["config"]["other_workloads"]           # ... and again under `config`, which every consumer of the line keeps whole
    assert brief["ring"]["ms_per_step"] == d["workloads"]["ring"]["ms_per_step"] and brief["lb"]["events_per_s"] == d["workloads"]["lb"]["value"]
    assert brief["strong_shard_8192"]["ms_per_step"] == sh["ms_per_step"]
    assert sh["events_per_step"] < d["config"]["events_per_step_per_gpu"]
    gr = brief["graph_replicas"]                     # round 6: replicas of a graph outside the station shape, one workgroup each
    assert gr["replicas"] == 1024 and gr["events"] > 1024 * 1000 and 0 < gr["device_ms"] < 1e3 * gr["wall_s_python_api"]
    assert gr["events_per_s_device"] > gr["events_per_s_python_api"] > 1e6
    gp = brief["graph_parts"]                        # ... and ONE Simulation's disconnected parts on heaps of their own
    assert gp["chains"] == 16384 and gp["heaps"] == 2048 and gp["events"] > 16384 * 100
    assert gp["events_per_s_device"] > gp["events_per_s_python_api"] > 1e6


def test_fake_ranks_run_the_multi_rank_bench_paths_on_one_gpu():
    """`--fake-ranks 2`: two processes on one GPU over gloo run what `--gpus 2` runs on two GPUs -- the self-launch, the strong
    split of the grid, `other_scaling`, the MAX / SUM reductions, and the sharded ring's exchange rounds on real shards."""
    one = _run("--steps", "2", "--warmup", "1", "--n-lp", "4096", "--end-s", "5", "--cpu-sample-s", "0", "--api-run", "0", "--extras", "0")
    two = _run("--steps", "2", "--warmup", "1", "--n-lp", "4096", "--end-s", "5", "--cpu-sample-s", "0", "--fake-ranks", "2")
    assert two["fake_ranks"] == 2 and two["n_gpus"] == 1 and two["scaling"] == "strong"
    assert two["config"]["n_lp_per_gpu"] == 2048 and two["other_scaling"]["scaling"] == "weak"
    # the strong split's ranks hold the same 4 096 chains (disjoint stream ids by lp_base): the same events in total
    # (every rank is a Simulation of its own here, so each processes its own one event beyond end_time)
    assert abs(round(two["value"] * two["ms_per_step"] * 1e-3) - one["config"]["events_per_step_per_gpu"]) <= 2
    assert two["other_scaling"]["n_lp_per_gpu"] == 4096
    ring1 = _run("--workload", "ring", "--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "3", "--cpu-sample-s", "0")
    ring2 = _run("--workload", "ring", "--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "3", "--cpu-sample-s", "0",
                 "--fake-ranks", "2")
    # round 6: the LIVE exchange by default -- one launch per rank and run, the two kernels talk through each other's link queues
    assert ring2["fake_ranks"] == 2 and ring2["config"]["parallelism"].count("LIVE") == 1
    assert ring2["config"]["events_per_step"] == ring1["config"]["events_per_step"]       # two shards == one engine
    assert ring2["config"]["launches_per_step"] == 1
    # ... and the asynchronous rounds with the device-side exchange (round 5) on request
    ring3 = _run("--workload", "ring", "--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "3", "--cpu-sample-s", "0",
                 "--fake-ranks", "2", "--ring-exchange", "device")
    assert "gloo" in ring3["config"]["parallelism"] and "peers' buffers" in ring3["config"]["parallelism"]
    assert ring3["config"]["events_per_step"] == ring1["config"]["events_per_step"] and ring3["config"]["launches_per_step"] >= 1
