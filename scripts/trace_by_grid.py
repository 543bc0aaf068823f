#!/usr/bin/env python3
"""Mean kernel duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV (developer tool):
python scripts/trace_by_grid.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        acc[(short, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items()):
    v.sort()
    print(f"{name:64s} grid {grid:9d}  n={len(v):4d}  median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}")
