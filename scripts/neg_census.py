#!/usr/bin/env python3
"""Where do the lean loop's 8-byte VALU encodings come from?  Compiles scripts/fused_dev.hip with -gline-tables-only and counts, per source
line, the VOP3 instructions of the hot loop that are VOP3 only because of a source modifier (neg / abs): a 4-byte VOP2 issues ~17 %
faster at four wavefronts per SIMD (profiles/r4_valu_issue_rate_calibration.txt).   python scripts/neg_census.py [file_g.s]"""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_mix as M

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build_var", "fused_dev_g.s")
if len(sys.argv) <= 1:
    subprocess.run(["hipcc"] + M.FLAGS + ["-gline-tables-only", f"-I{ROOT}/cppflow_amd/csrc", f"-I{ROOT}/include", "-S", "--cuda-device-only", "-o", path,
                    os.path.join(ROOT, "scripts", "fused_dev.hip")], check=True, stderr=subprocess.DEVNULL)
files, cur, in_loop = {}, None, False
per_line, per_line_all = Counter(), Counter()
# find the lean loop by label: the same selection as isa_mix (no VMEM / LDS on the likely path, most VALU)
kern = M.parse(path, "")
name, blocks = next(iter(kern.items()))
regs = M.regions(blocks)
lean = [r for r in regs if len(r) == 4 and M.census(r[1])["vmem"] == 0 and M.census(r[1])["lds"] == 0]
best = max(lean, key=lambda r: M.census(r[1])["valu"])
header = re.search(r"at (\.LBB\w+)", best[0]).group(1)
src = {}
for ln in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    m = re.match(r"^(\.LBB\w+):\s*(;.*)?$", ln)
    if m:
        c = m.group(2) or ""
        if m.group(1) == header:
            in_loop = True
        elif in_loop and "Depth=" not in c:
            in_loop = False
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    m = re.match(r"^\t(v_[a-z_0-9]+)\s*(.*)", ln)
    if m and in_loop and cur:
        op, args = m.group(1), m.group(2).split(";")[0]
        per_line_all[cur] += 1
        if op in ("v_fma_f32", "v_mul_f32_e64", "v_add_f32_e64", "v_sub_f32_e64") and ("-" in args or "|" in args):
            per_line[cur] += 1
print("VOP3-by-modifier instructions of the lean loop (all blocks of the loop), by source line:", sum(per_line.values()))
for (f, l), n in sorted(per_line.items(), key=lambda kv: -kv[1])[:60]:
    text = ""
    p = os.path.join(ROOT, "cppflow_amd", "csrc", f)
    if os.path.exists(p):
        text = open(p).read().splitlines()[l - 1].strip()[:110]
    print(f"  {n:4d} (of {per_line_all[(f, l)]:4d} VALU)  {f}:{l}  {text}")
