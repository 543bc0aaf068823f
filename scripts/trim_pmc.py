#!/usr/bin/env python3
"""Keep only the rows of a rocprofv3 pmc_counter_collection.csv that scripts/summarize_profiles.py reads: the LAST `n` dispatches of
the probe's kernels (the input generators launch thousands of others; untrimmed the file is > 100 MB).

    python scripts/trim_pmc.py <csv> <n_dispatches>
"""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2])
KEEP = ("lm_fused_kernel", "lm_quad_kernel", "collision_kernel")
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    fields = rd.fieldnames
    rows = [r for r in rd if any(k in r["Kernel_Name"] for k in KEEP)]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-n:]
keep = set(ids)
with open(path, "w", newline="") as f:
    wr = csv.DictWriter(f, fieldnames=fields)
    wr.writeheader()
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            wr.writerow(r)
print(f"{path}: kept {len(keep)} dispatches")
