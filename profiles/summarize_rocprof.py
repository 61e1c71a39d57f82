#!/usr/bin/env python3
"""Turn a rocprofv3 `*_results.db` (rocpd sqlite) into the plain-text per-kernel summary we commit.

    python profiles/summarize_rocprof.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# name | calls | total_us | avg_us | min_us | max_us | pct")
    rows = cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    for name, calls, tot, avg, mn, mx in rows:
        print(f"{name} | {calls} | {tot:.3f} | {avg:.3f} | {mn:.3f} | {mx:.3f} | {100 * tot / total:.2f}")
    print("# per-dispatch resources (first dispatch of each kernel): grid, workgroup, vgpr, sgpr, lds, scratch")
    for row in cur.execute(
            "select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
            "from kernels group by name"):
        print(" | ".join(str(x) for x in row))


if __name__ == "__main__":
    main(sys.argv[1])
