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
    ("--n-lp", "4096", "--end-s", "20", "--cpu-sample-s", "2"),
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
    if "--workload" not in args:             # the grid line also carries the API run and the reference's own Python path
        assert "api_run_s" in d["config"] and d["config"]["api_events"] == d["config"]["events_per_step_per_gpu"]
        ref = c_ref = d["cpu_baseline"].get("reference_python")
        assert ref is None or (ref["single_process_65536_chains"]["value"] > 1e3 and "cpu" in c_ref)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e4 and c["unit"] == "events/s" and c["sample"]
    assert d["value"] > c["value"]


def test_bench_strong_scaling_and_self_launch_flags():
    """`--scaling strong` keeps the metric's total station count (here on one GPU: the same numbers as weak), and a plain
    `python bench.py --gpus N` without a launcher starts its own ranks (checked with N = 1: no torch.distributed.run child)."""
    weak = _run("--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "5", "--cpu-sample-s", "0", "--api-run", "0")
    strong = _run("--steps", "2", "--warmup", "1", "--n-lp", "2048", "--end-s", "5", "--cpu-sample-s", "0", "--api-run", "0",
                  "--scaling", "strong")
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert strong["config"]["events_per_step_per_gpu"] == weak["config"]["events_per_step_per_gpu"]
