#!/usr/bin/env python3
"""profiles/<tag>_{trace,fetch,write,sq}.txt (profiles/summarize_rocprof.py output of profiles/collect.sh) ->
profiles/<tag>_roofline_<workload>.json, the measured side of bench.py's `roofline` object:

  hbm_bytes_per_launch   FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KB -> bytes, of the
                         dominant kernel, per dispatch (separate --pmc passes)
  valu_busy_frac         SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES summed over the SEs' SIMDs ...) -- see below
  kernel_us_rocprof      average duration of that kernel under `rocprofv3 --kernel-trace --stats`
  commit                 the commit the profiled tree was built from (head_commit(): env / profiles/HEAD_COMMIT / git);
                         bench.py prints traffic = null when the kernel sources (`csrc_sha16`) changed since

VALU issue fraction: SQ_ACTIVE_INST_VALU counts, per SIMD, the (quad-)cycles in which the VALU was executing; SQ_WAVE_CYCLES
counts wave-resident (quad-)cycles summed over the waves.  With ONE wave per SIMD the ratio of the two is the fraction of the
kernel's time the VALU was busy; with W waves per SIMD it is that fraction divided by W -- `waves_per_simd` undoes it.

    python profiles/derive_roofline.py r02 grid "hs_station_run<1, false, true>" 2
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def csrc_sha16():
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(HERE), "happy_simulator_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def head_commit():
    """The commit the profiled tree was built from.  The GPU box has no .git (gpurun snapshots the work tree), so the commit
    travels with the snapshot: HS_PROFILE_COMMIT, else profiles/HEAD_COMMIT (profiles/collect.sh's caller writes it with
    `git rev-parse --short=12 HEAD > profiles/HEAD_COMMIT` before the gpurun call), else git here.  The identity bench.py
    checks is `csrc_sha16`, the hash of the kernel sources themselves."""
    c = os.environ.get("HS_PROFILE_COMMIT", "").strip()
    f = os.path.join(HERE, "HEAD_COMMIT")
    if not c and os.path.exists(f):
        c = open(f).read().strip()
    if not c:
        c = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=HERE).stdout.strip()
    return c or None


def rows(path, kernel):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("#") or kernel not in line:
            continue
        parts = [p.strip() for p in line.rsplit("|", 4)]
        if len(parts) == 5 and parts[1].isupper():
            try:
                out[parts[1]] = (int(parts[2]), float(parts[3]), float(parts[4]))   # dispatches, sum, per dispatch
            except ValueError:
                pass
    return out


def kernel_time(path, kernel):
    for line in open(path):
        if line.startswith("#") or kernel not in line:
            continue
        parts = [p.strip() for p in line.rsplit("|", 6)]
        if len(parts) == 7:
            try:
                return float(parts[3]), int(parts[1])
            except ValueError:
                continue
    return None, 0


def main(tag, workload, kernel, waves_per_simd):
    f = rows(os.path.join(HERE, f"{tag}_fetch.txt"), kernel)
    w = rows(os.path.join(HERE, f"{tag}_write.txt"), kernel)
    sq = rows(os.path.join(HERE, f"{tag}_sq.txt"), kernel)
    avg_us, calls = kernel_time(os.path.join(HERE, f"{tag}_trace.txt"), kernel)
    out = {"kernel": kernel, "tag": tag, "workload": workload,
           "commit": head_commit(),
           "kernel_us_rocprof": avg_us, "dispatches": calls, "csrc_sha16": csrc_sha16()}
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        out["FETCH_SIZE_KB_per_launch"] = f["FETCH_SIZE"][2]
        out["WRITE_SIZE_KB_per_launch"] = w["WRITE_SIZE"][2]
        out["fetch_correction"] = "x2 (gfx950: rocprofv3 reports half the bytes of coalesced streaming reads)"
        out["hbm_bytes_per_launch"] = int(round((2.0 * f["FETCH_SIZE"][2] + w["WRITE_SIZE"][2]) * 1024))
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_WAVE_CYCLES" in sq:
        out["SQ_per_launch"] = {k: v[2] for k, v in sq.items()}
        out["waves_per_simd"] = waves_per_simd
        out["valu_busy_frac"] = sq["SQ_ACTIVE_INST_VALU"][2] / sq["SQ_WAVE_CYCLES"][2] * waves_per_simd
        if "SQ_WAIT_ANY" in sq:
            out["wait_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"][2] / sq["SQ_WAVE_CYCLES"][2]
    path = os.path.join(HERE, f"{tag}_roofline_{workload}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
