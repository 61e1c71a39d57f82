"""CPU: the design check behind DESIGN.md section 9 (arrival tables from a cooperative kernel) stays true for the header the
engine compiles: an adaptive-Simpson integral of csrc/hs_profile.hpp that is expanded breadth-first, evaluated sub-tree by
sub-tree and added in the tree's own order has the bits of the sequential walk (tools/simpson_split_check.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_evaluation_of_the_adaptive_simpson_tree_is_bit_identical():
    for lanes, tasks in ((64, 1024), (7, 5)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "simpson_split_check.py"), "--lanes", str(lanes),
                              "--tasks", str(tasks)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [ln for ln in out.stdout.splitlines() if "intervals" in ln]
        assert len(lines) == 6 and all("bit-identical" in ln for ln in lines)
        assert any("12120637" in ln for ln in lines)         # the explosive case really is walked (12 M intervals)
