#!/usr/bin/env python3
"""ISA-level instruction mix of a kernel of libhs_hip.so ("instruction floor" needs evidence).

Extracts the gfx950 code object from an object file of the build (happy_simulator_amd/lib/obj/hs_inst_<k>.o: the group that
instantiates the kernel, csrc/hs_kernels.hpp HS_INST_GROUP_<k>), disassembles it with llvm-objdump and counts the instructions
of one kernel by class -- whole kernel and, with --loop, the innermost backward-branch region that contains the most
instructions (the request-order loop of hs_station_run<1, false, true, true>: consumer and producer halves are two loops of the
same kernel; both are reported).  Static counts: one execution of every instruction of the region.

    python tools/instruction_mix.py --group 0 --kernel 'hs_station_run<1, false, true, true>' > profiles/r03_isa_mix_grid.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("v_mad_u64_u32 (Philox: quarter rate)", r"^v_mad_u64_u32"),
    ("fp64 fma / mul / add", r"^v_(fma|mul|add|max|min|trunc|floor|ldexp|frexp|rndne|fract)_f64|^v_div_(scale|fmas|fixup)_f64|^v_rcp_f64"),
    ("fp64 <-> int conversions", r"^v_cvt_(f64_[iu]32|[iu]32_f64)"),
    ("fp64 compares", r"^v_cmp\w*_f64"),
    ("64-bit integer add / sub (2 instructions each)", r"^v_(add|sub|subb|addc)_co(_ci)?_u32|^v_lshl_add_u64"),
    ("64-bit integer compares", r"^v_cmp\w*_[iu]64"),
    ("selects (v_cndmask)", r"^v_cndmask"),
    ("32-bit integer / logic VALU", r"^v_(and|or|xor|not|lshl|lshr|ashr|add|sub|mul|mad|bfe|bfi|alignbit|perm|min|max|med3)\w*_[biu](16|32)|^v_(add|sub)_u32|^v_add3|^v_lshl_or|^v_and_or|^v_or3|^v_xad|^v_mov_b32|^v_readlane|^v_writelane|^v_readfirstlane|^v_accvgpr|^v_cmp\w*_[iu]32|^v_cmp\w*_u16|^v_bcnt|^v_mbcnt"),
    ("other VALU", r"^v_"),
    ("LDS (ds_*)", r"^ds_"),
    ("global / flat / scratch memory", r"^(global|flat|scratch|buffer)_"),
    ("s_waitcnt / s_sleep / s_barrier / s_nop", r"^s_(waitcnt|sleep|barrier|nop)"),
    ("branches", r"^s_(cbranch|branch)"),
    ("other SALU", r"^s_"),
]


def classify(mn):
    for name, pat in CLASSES:
        if re.match(pat, mn):
            return name
    return "unclassified"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        # the fat object: its .hip_fatbin section is a clang offload bundle; unbundle the gfx950 code object
        fb, out = os.path.join(d, "fatbin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fb}", obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], stderr=subprocess.DEVNULL)
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--demangle", out], text=True)


def kernel_body(asm, kernel):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            if start is not None:
                return lines[start:i]
            if m.group(1).startswith(kernel) or kernel in m.group(1):
                if ".kd" in m.group(1):
                    continue
                start = i + 1
    return lines[start:] if start is not None else []


def parse(body):
    ins = []          # (address, mnemonic, operands)
    for ln in body:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def loops(ins, asm_body):
    """Backward branches: (target address, branch address)."""
    addr = {a: k for k, (a, _, _) in enumerate(ins)}
    out = []
    for k, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>|(-?\d+)\s*$", ops)
            tgt = None
            if m and m.group(2) is not None:
                off = int(m.group(2))
                off -= 65536 if off > 32767 else 0            # (simm16, printed unsigned)
                tgt = a + 4 + 4 * off
            if tgt is not None and tgt in addr and tgt <= a:
                out.append((addr[tgt], k))
    return out


def table(ins, title):
    c = collections.Counter(classify(mn) for _, mn, _ in ins)
    total = sum(c.values())
    print(f"## {title}: {total} instructions")
    for name, _ in CLASSES + [("unclassified", "")]:
        if c.get(name):
            print(f"  {c[name]:6d}  {100.0 * c[name] / total:5.1f} %  {name}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--group", type=int, default=0)
    ap.add_argument("--kernel", default="void hs_station_run<1, false, true, true>")
    ap.add_argument("--obj", default=None)
    ap.add_argument("--top-loops", type=int, default=3)
    a = ap.parse_args()
    obj = a.obj or os.path.join(ROOT, "happy_simulator_amd", "lib", "obj", f"hs_inst_{a.group}.o")
    asm = disassemble(obj)
    body = kernel_body(asm, a.kernel)
    ins = parse(body)
    if not ins:
        sys.exit(f"kernel {a.kernel!r} not found in {obj}")
    print(f"# {a.kernel} in {os.path.relpath(obj, ROOT)} (llvm-objdump -d of the gfx950 code object; static counts)")
    table(ins, "whole kernel")
    lp = sorted(set(loops(ins, body)), key=lambda ab: (ab[0], -ab[1]))
    print(f"## backward-branch regions ({len(lp)}): first..last instruction, size, regions nested inside")
    outer = []
    for s, e in lp:
        inside = sum(1 for s2, e2 in lp if (s2, e2) != (s, e) and s <= s2 and e2 <= e)
        print(f"  {s:6d}..{e:6d}  {e - s + 1:6d}  {inside:3d}")
        if not any(s2 <= s and e <= e2 and (s2, e2) != (s, e) for s2, e2 in lp):
            outer.append((s, e))
    outer.sort(key=lambda ab: ab[0] - ab[1])
    for n, (s, e) in enumerate(outer[:a.top_loops]):
        table(ins[s:e + 1], f"outermost loop {n + 1}: instructions {s}..{e} (backward branch at +{ins[e][0] - ins[0][0]:#x})")
    inner = [(s, e) for s, e in lp if not any((s2, e2) != (s, e) and s <= s2 and e2 <= e for s2, e2 in lp)]
    inner.sort(key=lambda ab: ab[0] - ab[1])
    for n, (s, e) in enumerate(inner[:a.top_loops]):
        table(ins[s:e + 1], f"innermost loop {n + 1}: instructions {s}..{e}")


if __name__ == "__main__":
    main()
