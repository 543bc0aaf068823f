#!/usr/bin/env python3
"""Opcode census of the headline kernel, region by region, from `hipcc -S` output (no GPU needed).

    python scripts/isa_mix.py [file.s | --compile] [-DMACRO=value ...] [kernel-name filter]

`--compile` (the default when no file is given) compiles scripts/fused_dev.hip -- lm_fused_kernel<StaRobot<Panda>, 1> alone -- with
the flags cppflow_amd/build.py gives csrc/fused_static.hip (5 s) into build_var/fused_dev.s.

The kernel is cut into REGIONS with LLVM's own loop annotations ("=>This Loop Header: Depth=1", "in Loop: Header=...", "Parent
Loop ..."): the code in front of the first depth-1 loop (load + the launch's first, lean iteration), every depth-1 loop, the code
between / behind them (the last, canonical iteration; the finish stage).  Inside a depth-1 loop the HOT BODY is the likely
path of one trip: from the loop header, falling through every conditional branch (the rare paths -- the conditioning gate's re-solve
rounds, the general form of the angle functions -- are the taken ones) and following unconditional ones up to the back-edge; the LM
iteration is the depth-1 loop with the largest hot body.

Per region: VALU instructions by class (fma / mul / add-sub / min-max-med / cmp / cndmask / mov / cvt / bit ops / transcendental /
DPP / fp64 / other), 4-byte vs 8-byte encodings (VOP2 / VOP1 / VOPC e32 against VOP3 / literal forms: an 8-byte VALU instruction
issues ~17 % slower at four wavefronts per SIMD, profiles/r4_valu_issue_rate_calibration.txt), and the executed-flops-per-VALU-lane-op
figure (FMA = 2, mul / add = 1, everything else 0) that bench.py multiplies SQ_INSTS_VALU with."""
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize",
         "-Wno-comment", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-enable-post-misched=0"]

CLASSES = OrderedDict([
    ("fp64", re.compile(r"^v_\w+_f64|^v_cvt_f64|^v_cvt_f32_f64")),
    ("dpp", re.compile(r"_dpp$")),
    ("fma", re.compile(r"^v_(fma|fmac|fmaak|fmamk|mad|mac)_f32|^v_pk_fma_f32")),
    ("mul", re.compile(r"^v_mul_f32|^v_pk_mul_f32")),
    ("addsub", re.compile(r"^v_(add|sub|subrev)_f32|^v_pk_add_f32")),
    ("minmaxmed", re.compile(r"^v_(min|max|med)3?_f32")),
    ("trans", re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_")),
    ("cmp", re.compile(r"^v_cmpx?_")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("mov", re.compile(r"^v_mov|^v_accvgpr|^v_readlane|^v_readfirstlane|^v_writelane")),
    ("cvt", re.compile(r"^v_cvt_")),
    ("bits", re.compile(r"^v_(and|or|xor|not|bfi|bfe|lshl|lshr|ashr|bitop|perm|alignbit|mbcnt|bcnt|ffb)")),
    ("int", re.compile(r"^v_(add|sub|mul|mad|addc|subb)\w*_(u32|i32|u24|i24|co_u32|u64)|^v_add3|^v_lshl_add|^v_add_lshl|^v_lshl_or")),
])
EIGHT = re.compile(r"_e64|^v_fma_f32$|^v_fmaak|^v_fmamk|^v_med3|^v_max3|^v_min3|^v_bfi|^v_bfe|^v_bitop|^v_perm|^v_add3|^v_lshl_add|^v_add_lshl|^v_lshl_or|"
                   r"^v_fma_f64|^v_mul_f64|^v_add_f64|^v_mad|_dpp$|^v_alignbit|^v_div|^v_cndmask_b32_e64|^v_ldexp|^v_pk_")


def compile_dev(defines=()):
    out = os.path.join(ROOT, "build_var", "fused_dev.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["hipcc"] + FLAGS + list(defines) + [f"-I{ROOT}/cppflow_amd/csrc", f"-I{ROOT}/include", "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "scripts", "fused_dev.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def parse(path, flt):
    """-> {kernel: [block]}, block = dict(label, depth (LLVM's loop depth of the block), is_header, ins [(op, args)])"""
    kernels, cur, blocks = {}, None, None
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = name if (flt in name and "lm_fused_kernel" in name or (flt and flt in name)) else None
            if cur is not None:
                blocks = kernels.setdefault(cur, [dict(label="entry", depth=0, is_header=False, ins=[])])
            continue
        if cur is None:
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\w+):\s*(;.*)?$", ln)
        if m:
            c = m.group(2) or ""
            depth = int(re.search(r"Depth=(\d+)", c).group(1)) if "Depth=" in c else 0
            blocks.append(dict(label=m.group(1), depth=depth, is_header="This Loop Header" in c, ins=[]))
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", ln)
        if m and not m.group(1).startswith(".") and blocks is not None:
            blocks[-1]["ins"].append((m.group(1), m.group(2)))
    return kernels


def census(ins):
    c = Counter()
    for op, _ in ins:
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["valu8" if EIGHT.search(op) else "valu4"] += 1
            for name, rx in CLASSES.items():
                if rx.search(op):
                    c[name] += 1
                    break
            else:
                c["other"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    c["flops_per_valu"] = (2 * c["fma"] + c["mul"] + c["addsub"]) / max(c["valu"], 1)
    # issue-cycle estimate at four wavefronts per SIMD from the calibration (profiles/r4_valu_issue_rate_calibration.txt): 2.41 cycles for a
    # 4-byte VALU instruction, 2.89 for an 8-byte one (2.65 for the literal forms), ~12 for a transcendental among multiply-adds
    lit = sum(1 for op, _ in ins if op.startswith(("v_fmaak", "v_fmamk")))
    c["issue_cycles"] = 2.41 * c["valu4"] + 2.89 * (c["valu8"] - lit) + 2.65 * lit + (12 - 2.41) * c["trans"]
    return c


def fmt(c):
    cls = " ".join(f"{k} {c[k]}" for k in list(CLASSES) + ["other"] if c[k])
    return (f"VALU {c['valu']:5d} (4-byte {c['valu4']}, 8-byte {c['valu8']})  [{cls}]  SALU {c['salu']} LDS {c['lds']} VMEM {c['vmem']}  "
            f"flops/VALU lane-op {c['flops_per_valu']:.3f}  est. issue cycles {c['issue_cycles']:.0f}")


def regions(blocks):
    """cut at depth-1 loop headers: [(title, hot instructions, all instructions)]"""
    # a depth-1 loop spans from its header to the last block that names it (directly or through a parent)
    out, i, n = [], 0, len(blocks)
    straight = []
    k = 0
    while i < n:
        b = blocks[i]
        if b.get("is_header") and b["depth"] == 1:
            if straight:
                out.append((f"straight-line region {k}", [x for bb in straight for x in bb["ins"]], None))
                k += 1
                straight = []
            j = i + 1
            while j < n and blocks[j]["depth"] >= 1:
                j += 1
            loop = blocks[i:j]
            # hot body: the LIKELY path of one trip -- from the header, fall through every conditional branch (the rare paths are the
            # taken ones: the kernels mark them with __builtin_expect, and the compiler lays the likely successor out next), follow
            # unconditional branches, stop at the back-edge or when the walk leaves the loop
            index = {bb["label"]: n for n, bb in enumerate(loop)}
            hot, n_, seen = [], 0, set()
            while n_ is not None and n_ < len(loop) and n_ not in seen:
                seen.add(n_)
                bb = loop[n_]
                hot += bb["ins"]
                last = bb["ins"][-1] if bb["ins"] else ("", "")
                if last[0] == "s_branch":
                    tgt = last[1].strip().split()[0]
                    n_ = index.get(tgt) if tgt != loop[0]["label"] else None
                else:
                    n_ = n_ + 1
            tail = [bb for bb in loop[1:] if bb["depth"] == 1]
            out.append((f"depth-1 loop at {b['label']} ({len(loop)} blocks)", hot, [x for bb in loop for x in bb["ins"]], [x for bb in tail for x in bb["ins"]]))
            i = j
            continue
        straight.append(b)
        i += 1
    if straight:
        out.append((f"straight-line region {k}", [x for bb in straight for x in bb["ins"]], None))
    return out


def main():
    args = [a for a in sys.argv[1:]]
    path = None
    if args and args[0].endswith(".s"):
        path = args.pop(0)
    if args and args[0] == "--compile":
        args.pop(0)
    defines = [a for a in args if a.startswith("-D")]
    args = [a for a in args if not a.startswith("-D")]
    flt = args[0] if args else ""
    if path is None:
        path = compile_dev(defines)
    for name, blocks in parse(path, flt).items():
        print(name[:150])
        allins = [x for b in blocks for x in b["ins"]]
        print("  whole kernel (static):", fmt(census(allins)))
        regs = regions(blocks)
        # the LM iteration = the lean loop: the depth-1 loop whose likely path stores nothing (the general iteration's holds the J / e
        # output stores) and is the longest of those
        lean = [r for r in regs if len(r) == 4 and census(r[1])["vmem"] == 0 and census(r[1])["lds"] == 0]
        best = max(lean or [r for r in regs if len(r) == 4], key=lambda r: census(r[1])["valu"], default=None)
        for r in regs:
            if len(r) == 4:
                tag = "  <== the LM iteration (hot loop)" if r is best else ""
                print(f"  {r[0]}{tag}")
                print("      likely path of one trip (rare branches not taken):       ", fmt(census(r[1])))
                print("      other depth-1 blocks (update / clamp, gate bookkeeping):  ", fmt(census(r[3])))
                print("      whole loop incl. nested (gate re-solve rounds):            ", fmt(census(r[2])))
            else:
                print(f"  {r[0]}:", fmt(census(r[1])))
        if best is not None:
            c = census(best[1])
            top = Counter(op for op, _ in best[1] if op.startswith("v_")).most_common(14)
            print("  hot loop opcodes:", ", ".join(f"{op} {n}" for op, n in top))
            non = c["valu"] - c["fma"] - c["mul"] - c["addsub"]
            print(f"  hot loop: {non} of {c['valu']} VALU are not fma / mul / add ({100.0 * non / c['valu']:.1f} %); {c['valu8']} of {c['valu']} are 8-byte encodings")


if __name__ == "__main__":
    main()
