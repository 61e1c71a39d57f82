"""GPU: the multi-GPU path through torch.distributed with backend "nccl" (= RCCL on ROCm) on REAL shards -- the
all_to_all_single of the outbox rows, the all_reduce (MAX of the cross links' bounds / MIN of the GVT), the all_gather of the
overshoot candidates and the totals' reductions all run on device tensors.  With one GPU the process group has world size 1
(every collective still goes through RCCL); with two or more visible GPUs the same worker runs as two ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_ring_worker.py")


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _check(out, world):
    one = out["single"]
    for proto in ("rounds", "windows"):
        r = out[proto]
        assert r["world"] == world
        assert r["events"] == one["events"] and r["final"] == one["final"], (proto, r, one)
        assert r["exchanges"] >= 1
    assert out["rounds"]["exchanges"] < out["windows"]["exchanges"]       # bounds travel further than the 1 ms link floor
    assert out["rounds"]["completed_local"] == out["windows"]["completed_local"] == one["completed_local"]


def test_distcomm_over_nccl_world_1_on_a_real_shard():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    p = subprocess.run([sys.executable, WORKER, "4096", "3.0"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    _check(out, 1)


def test_two_ranks_over_rccl_equal_the_single_engine():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier); world size 1 is covered above")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), WORKER, "8192", "3.0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    _check(out, 2)
