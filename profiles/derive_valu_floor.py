#!/usr/bin/env python3
"""Class-weighted VALU issue floor of a kernel.

    python profiles/derive_valu_floor.py <tag> <workload> "<kernel name>" <waves per SIMD> <classes.txt> <valu_rates.json>

Inputs: the DYNAMIC instruction counts of the kernel by class -- a `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 ... _INT32
_INT64 _CVT` pass summarised by profiles/summarize_rocprof.py (wave-instructions per dispatch) -- and the issue cost of each class
MEASURED on the same kind of device at the kernel's occupancy (tools/valu_rates.hip: ns of one SIMD per wave-instruction with W
wavefronts per SIMD).  Output profiles/<tag>_valu_floor_<workload>.json:

    floor_simd_ns_per_launch = sum over classes of count x ns / SIMDs    (what the kernel's vector instructions cost if every issue
                                                                          slot were used: no dependency stall, no LDS / memory wait)

bench.py divides it by the kernel time it measures: `roofline.valu_floor_frac`.  Every class is priced with the CHEAPEST instruction
it contains (INT32 as v_add_u32, "other" -- moves, logic, compares, selects -- as v_mov_b32), so the figure is a floor, not an
estimate.  The flat "4 cycles per instruction on a SIMD16" of rounds 4-5 was wrong twice: the CU has four SIMD-32s, a wave64 32-bit
op issues in 2 cycles (MI355X_MICROARCH.md "Wave scheduling"), binary64 in 4.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from derive_roofline import csrc_sha16, head_commit  # noqa: E402

N_SIMD = 256 * 4

# SQ counter -> the tools/valu_rates.hip class that prices it (the cheapest member of the class)
PRICE = {
    "SQ_INSTS_VALU_ADD_F64": "v_add_f64",
    "SQ_INSTS_VALU_MUL_F64": "v_mul_f64",
    "SQ_INSTS_VALU_FMA_F64": "v_fma_f64",
    "SQ_INSTS_VALU_TRANS_F64": "v_rcp_f64",
    "SQ_INSTS_VALU_CVT": "v_cvt_f64_i32",
    "SQ_INSTS_VALU_INT32": "v_add_u32",
    "SQ_INSTS_VALU_INT64": "v_lshl_add_u64",
    "other (moves, logic, compares, selects, DPP)": "v_mov_b32",
}


def counts(path, kernel):
    out = {}
    for line in open(path):
        if line.startswith("#") or kernel not in line:
            continue
        parts = [p.strip() for p in line.rsplit("|", 4)]
        if len(parts) == 5 and parts[1].startswith("SQ_"):
            try:
                out[parts[1]] = float(parts[4])
            except ValueError:
                pass
    return out


def main():
    tag, workload, kernel, wps, cls_path, rates_path = sys.argv[1:7]
    wps = int(wps)
    c = counts(cls_path, kernel)
    if "SQ_INSTS_VALU" not in c:
        sys.exit(f"no SQ_INSTS_VALU row for {kernel!r} in {cls_path}")
    rates = {}
    for r in json.load(open(rates_path))["rates"]:
        rates.setdefault(r["class"], {})[r["waves_per_simd"]] = r["ns_per_wave_inst_per_simd"]
    w_avail = sorted({w for v in rates.values() for w in v})
    w_use = max([w for w in w_avail if w <= wps] or [min(w_avail)])
    named = sum(v for k, v in c.items() if k in PRICE)
    c["other (moves, logic, compares, selects, DPP)"] = c["SQ_INSTS_VALU"] - named
    table, floor_ns = [], 0.0
    for cls, inst in PRICE.items():
        n = c.get(cls, 0.0)
        ns = rates[inst][w_use]
        table.append({"class": cls, "wave_insts_per_launch": n, "priced_as": inst, "ns_per_wave_inst_per_simd": ns,
                      "cycles_at_2p4GHz": ns * 2.4, "simd_ns_per_launch": n * ns / N_SIMD})
        floor_ns += n * ns / N_SIMD
    out = {"kernel": kernel, "tag": tag, "workload": workload, "commit": head_commit(), "csrc_sha16": csrc_sha16(),
           "waves_per_simd": wps, "rates_measured_at_waves_per_simd": w_use, "n_simd": N_SIMD,
           "SQ_INSTS_VALU_per_launch": c["SQ_INSTS_VALU"], "classes": table, "floor_simd_ns_per_launch": floor_ns,
           "flat_2_cycle_floor_simd_ns_per_launch": c["SQ_INSTS_VALU"] * 2 / 2.4 / N_SIMD,
           "rates_file": os.path.basename(rates_path), "classes_file": os.path.basename(cls_path)}
    path = os.path.join(HERE, f"{tag}_valu_floor_{workload}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "floor_simd_ns_per_launch", "flat_2_cycle_floor_simd_ns_per_launch")}))
    print("wrote", path)


if __name__ == "__main__":
    main()
