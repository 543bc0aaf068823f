#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc CSV (developer tool): python scripts/pmc_kernels.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        key = (name[:60], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key[0], "grid", key[1], " ".join(f"{c}={sum(v) / len(v):.0f}" for c, v in sorted(cs.items())), f"(n={len(next(iter(cs.values())))})")
