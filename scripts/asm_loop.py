#!/usr/bin/env python3
"""Instruction mix of every loop of the kernels in a hipcc -S assembly file (developer tool; see scripts/quad_dev.hip).

    python scripts/asm_loop.py build_var/quad_dev.s [kernel-name-filter]
"""
import re
import subprocess
import sys
from collections import Counter

FMA = re.compile(r"^v_(fma|fmac|fmaak|fmamk|mad|mac)_f32|^v_pk_fma_f32")
MULADD = re.compile(r"^v_(mul|add|sub|subrev)_f32|^v_pk_(mul|add)_f32")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_")


def parse(path):
    funcs, cur, name = {}, None, None
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None:
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur.append(("label", m.group(1)))
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", ln)
        if m and not m.group(1).startswith("."):
            cur.append(("ins", m.group(1), m.group(2)))
    return funcs


def stats(body):
    c = Counter()
    for it in body:
        if it[0] != "ins":
            continue
        op, args = it[1], it[2]
        c["n"] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "dpp" in op or "quad_perm" in args or "row_" in args:
                c["dpp"] += 1
            if FMA.match(op):
                c["fma"] += 1
            elif MULADD.match(op):
                c["muladd"] += 1
            elif TRANS.match(op):
                c["trans"] += 1
            elif op.startswith("v_mov"):
                c["mov"] += 1
            elif op.startswith("v_cndmask"):
                c["cnd"] += 1
            elif op.startswith("v_cmp"):
                c["cmp"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op == "s_nop":
                c["nop"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["vmem"] += 1
    return c


def main():
    funcs = parse(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, items in funcs.items():
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dn:
            continue
        print(dn[:150])
        pos = {it[1]: i for i, it in enumerate(items) if it[0] == "label"}
        loops = []
        for i, it in enumerate(items):
            if it[0] == "ins" and (it[1].startswith("s_cbranch") or it[1] == "s_branch"):
                tgt = it[2].strip()
                if tgt in pos and pos[tgt] < i:
                    loops.append((pos[tgt], i))
        c = stats(items)
        print(f"  whole kernel : {c['n']} instr, VALU {c['valu']}")
        for a, b in sorted(loops, key=lambda t: t[0] - t[1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 4]:
            c = stats(items[a : b + 1])
            print(f"  loop [{a}..{b}]: {c['n']} instr: VALU {c['valu']} (FMA {c['fma']}, mul/add {c['muladd']}, trans {c['trans']}, DPP {c['dpp']}, "
                  f"mov {c['mov']}, cndmask {c['cnd']}, cmp {c['cmp']}), MFMA {c['mfma']}, SALU {c['salu']} (s_nop {c['nop']}), LDS {c['lds']}, VMEM {c['vmem']}; "
                  f"flops/VALU lane-op {(2 * c['fma'] + c['muladd']) / max(c['valu'], 1):.3f}")


if __name__ == "__main__":
    main()
