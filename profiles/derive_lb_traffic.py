#!/usr/bin/env python3
"""profiles/<tag>_{trace,fetch,write}.txt of a `bench.py --workload lb` collection -> profiles/<tag>_roofline_lb.json: the HBM
bytes of ALL radix-sort kernels (radix_hist / radix_scan_* / radix_scatter, every pass of both sorts) per pipeline run, from the
separate --pmc passes (FETCH_SIZE x 2: gfx950 correction, MI355X_MICROARCH.md HBM section; WRITE_SIZE), their time per run from
the kernel trace, and the commit + hash of csrc/ the profile was taken at (bench.py prints traffic = null once the kernels change).

    python profiles/derive_lb_traffic.py r02lb        # pipeline runs under the profiler = calls of hs_lbk_backends in the trace
"""
import json
import os
import sys

from derive_roofline import HERE, csrc_sha16, head_commit

tag = sys.argv[1]
runs = 0
for line in open(os.path.join(HERE, f"{tag}_trace.txt")):
    if not line.startswith("#") and "hs_lbk_backends" in line:
        parts = [p.strip() for p in line.rsplit("|", 6)]
        if len(parts) == 7 and parts[1].isdigit():
            runs = int(parts[1])
            break
assert runs > 0, "no hs_lbk_backends row in the trace"
kb = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for name, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for line in open(os.path.join(HERE, f"{tag}_{name}.txt")):
        if line.startswith("#") or "radix_" not in line:
            continue
        parts = [p.strip() for p in line.rsplit("|", 4)]
        if len(parts) == 5 and parts[1] == key:
            kb[key] += float(parts[3])
sort_us = 0.0
for line in open(os.path.join(HERE, f"{tag}_trace.txt")):
    if line.startswith("#") or "radix_" not in line:
        continue
    parts = [p.strip() for p in line.rsplit("|", 6)]
    if len(parts) == 7:
        try:
            sort_us += float(parts[2])
        except ValueError:
            pass
out = {
    "kernels": "radix_hist + radix_scan_* + radix_scatter (all passes of both sorts)", "tag": tag, "workload": "lb",
    "commit": head_commit(),
    "csrc_sha16": csrc_sha16(), "pipeline_runs_profiled": runs,
    "FETCH_SIZE_KB_per_step": kb["FETCH_SIZE"] / runs, "WRITE_SIZE_KB_per_step": kb["WRITE_SIZE"] / runs,
    "fetch_correction": "x2 (gfx950: rocprofv3 reports half the bytes of coalesced streaming reads)",
    "hbm_bytes_per_launch": int((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) / runs * 1024),
    "sort_us_per_step_rocprof": sort_us / runs,
}
path = os.path.join(HERE, f"{tag}_roofline_lb.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
print("wrote", path)
