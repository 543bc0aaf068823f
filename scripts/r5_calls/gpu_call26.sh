#!/usr/bin/env bash
# round 5, GPU call 26: kernel trace of the dp_search forms
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
D="$OUT/dp_trace"; rm -rf "$D"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$D" -o kt -- python3 "$ROOT/scripts/dp_trace_probe.py" > /dev/null 2> "$D.stderr")
python3 - "$D" <<'PY' | tee "$OUT/dp_trace.txt"
import csv, glob, sys, collections
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
by = collections.OrderedDict()
for s, e, n, g in rows:
    if "dp_" in n or "fill" in n.lower():
        by.setdefault((n, g), []).append(e - s)
for (n, g), du in by.items():
    print(f"{len(du):5d} x  dur median {np.median(du) / 1e3:8.2f} us  grid {g:>8s}  {n}")
PY
rm -rf "$D"
