#!/usr/bin/env python3
"""Turn a rocprofv3 `*_results.db` (rocpd sqlite) into the plain-text summary we commit under profiles/.

    python profiles/summarize_rocprof.py gpurun_out/prof_r01/trace/r01_results.db > profiles/r01_trace.txt

Prints the per-kernel time statistics (what `--stats` reports) and, when the run collected PMC counters, the
per-kernel counter sums and per-dispatch averages.
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 summary of {path}")
    print("# name | calls | total_us | avg_us | min_us | max_us | pct")
    rows = cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    for name, calls, tot, avg, mn, mx in rows:
        print(f"{name} | {calls} | {tot:.3f} | {avg:.3f} | {mn:.3f} | {mx:.3f} | {100 * tot / total:.2f}")
    print("# per-dispatch resources: name | grid | workgroup | vgpr | agpr | sgpr | lds | scratch")
    for row in cur.execute(
            "select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
            "from kernels group by name"):
        print(" | ".join(str(x) for x in row))
    try:
        cols = [d[0] for d in cur.execute("select * from pmc_events limit 1").description]
    except sqlite3.Error:
        cols = []
    if cols:
        print(f"# pmc_events columns: {cols}")
        namecol = "counter_name" if "counter_name" in cols else ("name" if "name" in cols else None)
        valcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        kcol = "kernel_name" if "kernel_name" in cols else None
        if namecol and valcol:
            if kcol is None and "dispatch_id" in cols:
                q = (f"select k.name, p.{namecol}, count(distinct p.dispatch_id), sum(p.{valcol}) from pmc_events p "
                     f"join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.{namecol}")
            else:
                q = f"select {kcol}, {namecol}, count(*), sum({valcol}) from pmc_events group by {kcol}, {namecol}"
            print("# kernel | counter | dispatches | sum | per_dispatch")
            try:
                for k, c, n, v in cur.execute(q):
                    print(f"{k} | {c} | {n} | {v:.6g} | {v / max(n, 1):.6g}")
            except sqlite3.Error as e:
                print(f"# pmc query failed: {e}")


if __name__ == "__main__":
    main(sys.argv[1])
