#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop (the LM iteration) from the gfx950 disassembly of the built library.

    python scripts/isa_mix.py <substring of the demangled kernel name> [lib.so]

Finds the backward branch that spans the most instructions (the K loop), and prints how many VALU / SALU / MFMA / DPP / LDS
instructions its body holds, the FMA / mul-add / transcendental split, and the executed flops per VALU lane-operation
(FMA = 2, mul / add / sub = 1, everything else 0) that bench.py multiplies SQ_INSTS_VALU with."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, extract_code_objects  # noqa: E402


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


FMA = re.compile(r"^v_(fma|fmac|fmaak|fmamk|mad|mac)_f32|^v_pk_fma_f32")
MULADD = re.compile(r"^v_(mul|add|sub|subrev)_f32|^v_pk_(mul|add)_f32")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_")


def analyse(lines):
    """lines: list of (addr, op, text) of one kernel.  Returns (stats of the biggest loop body, stats of the whole kernel)."""
    addr_index = {a: i for i, (a, _, _) in enumerate(lines)}
    best = None
    for i, (a, op, text) in enumerate(lines):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>|(?:^|\s)(?:0x)?([0-9a-f]{4,})\s*$", text)
            tgt = None
            mm = re.search(r"// ([0-9A-Fa-f]+): ", text)
            mt = re.search(r"<[^+>]+\+0x([0-9a-fA-F]+)>", text)
            if mt:
                tgt = int(mt.group(1), 16)
            if tgt is None or tgt not in addr_index:
                continue
            j = addr_index[tgt]
            if j < i and (best is None or i - j > best[1] - best[0]):
                best = (j, i)

    def stats(sub):
        st = {"n": len(sub), "valu": 0, "salu": 0, "mfma": 0, "lds": 0, "vmem": 0, "dpp": 0, "fma": 0, "muladd": 0, "trans": 0}
        for _, op, text in sub:
            c = classify(op)
            if c in st:
                st[c] += 1
            if c == "valu":
                if "dpp" in op or "quad_perm" in text or "row_" in text:
                    st["dpp"] += 1
                if FMA.match(op):
                    st["fma"] += 2 if op.startswith("v_pk") else 1
                elif MULADD.match(op):
                    st["muladd"] += 2 if op.startswith("v_pk") else 1
                elif TRANS.match(op):
                    st["trans"] += 1
        st["flops_per_valu_lane_op"] = (2 * st["fma"] + st["muladd"]) / max(st["valu"], 1)
        return st

    return (stats(lines[best[0] : best[1] + 1]) if best else None), stats(lines)


def disassemble(co, mangled):
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={mangled}", co], capture_output=True, text=True,
                         check=True).stdout  # fmt: skip
    lines = []
    base = None
    for ln in out.splitlines():
        m = re.match(r"^\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", ln)
        if not m:
            continue
        addr = int(m.group(3), 16)
        if base is None:
            base = addr
        lines.append((addr - base, m.group(1), ln))
    return lines


def main():
    flt = sys.argv[1]
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "..", "cppflow_amd", "csrc", "libcppflow_hip.so")
    for co in extract_code_objects(lib, "/tmp/cppflow_gfx950.co"):  # one code object per translation unit
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True, check=True).stdout
        names = sorted({ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln})
        for mangled in names:
            dn = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            if flt not in dn:
                continue
            lines = disassemble(co, mangled)
            loop, whole = analyse(lines)
            print(dn[:160])
            for title, st in (("  hottest loop", loop), ("  whole kernel", whole)):
                if st:
                    print(f"{title}: {st['n']} instructions: VALU {st['valu']} (DPP {st['dpp']}, FMA {st['fma']}, mul/add {st['muladd']}, "
                          f"trans {st['trans']}), MFMA {st['mfma']}, SALU {st['salu']}, LDS {st['lds']}, VMEM {st['vmem']}; "
                          f"flops per VALU lane-op {st['flops_per_valu_lane_op']:.3f}")


if __name__ == "__main__":
    main()
