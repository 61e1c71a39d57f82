"""CPU: the emulation of the election of the one event beyond end_time (tools/election_rules.py, an analysis build of the
oracle).  Round 2's engines elected by (time, creation time, construction rank); the emulation of THAT key fails on exactly the
tie storms MI355X failed on then (profiles/r02_gpu_random_sweep*.log), which validates the emulation -- and the key the kernels
use since round 3 (the heap is a FIFO inside one nanosecond: creation time, steps from the group's root, the root's creation
time, construction rank; Probes by their own list position -- csrc/hs_station.hpp StationState lineage, csrc/hs_kernels.hpp
cand_less) elects the reference's LP on every one of them.  The GPU side: tests/test_gpu_random.py (2 000 tie storms, 1 000
several-Sources rings), profiles/r03_gpu_sweep_*_election.log."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "election_rules.py"), *args], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = {}
    for ln in out.stdout.splitlines():
        if " wrong on " in ln:
            name, rest = ln.strip().split(" wrong on ")
            rows[name.strip()] = eval(rest.split("):", 1)[1])         # the list of case numbers
    return rows


def test_election_emulation_matches_the_gpu_and_the_fifo_key_closes_the_deviation():
    rows = _run("--first", "0", "--count", "1000")
    assert rows["engine: (created, rank)"] == [85, 134, 279, 978]          # = the GPU's four of 1 000
    assert rows["(created, depth, root created, rank)"] == []
    rows = _run("--first", "2000", "--count", "353")
    assert rows["engine: (created, rank)"] == [2079, 2150]                 # = the second GPU sweep's two
    assert rows["(created, depth, root created, rank; Probes by their own list position)"] == []


def test_a_tick_ranks_by_its_own_source_position():
    """The rank fix of round 2 (csrc/hs_station.hpp cand_rank), on the several-Sources generator: the key the engine had before
    the GPU sweep fails on multi_source_spec(1374) -- the case the sweep found -- and the present one does not."""
    rows = _run("--family", "multi_source", "--first", "1300", "--count", "100")
    assert 1374 in rows["engine before the GPU sweep: (created, LP's first-listed Source)"]
    assert rows["engine: (created, rank)"] == []


def test_a_departure_of_a_server_with_several_sources_ranks_with_a_stand_in_and_such_ties_go_to_the_single_heap():
    """Round 4, found by the GPU sweep (multi_source_spec(22522)): two lock-step constant Sources of different Servers, the winner a
    DEPARTURE whose Server's first-listed Source is another (earlier constructed) one -- the construction rank of a non-tick is a
    stand-in once a Server has several Sources.  The engine now repeats such a run on the single heap when the election comes down to
    that key (csrc/hs_engine.hip set_stations, csrc/hs_kernels.hpp tie check): the rule catches the case, and no case slips past it."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "election_rules.py"), "--family", "multi_source", "--first", "22500",
                          "--count", "60"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if "round-4 rule" in ln][0]
    assert "would have been wrong [22522]" in line and "wrong and NOT caught: 0 []" in line, line
