#!/usr/bin/env bash
# round 5, GPU call 21: kernel trace of the one-trajectory coupled step (which kernels, how long, what gaps)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
D="$OUT/coupled_trace"; rm -rf "$D"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$D" -o kt -- python3 "$ROOT/scripts/coupled_trace_probe.py" fetch > /dev/null 2> "$D.stderr")
python3 - "$D" <<'PY' | tee "$OUT/coupled_trace.txt"
import csv, glob, sys, collections
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 100 steps of each robot: find the repeating kernel sequence
names = [r[2] for r in rows]
for robot_key in ("Fetch", "Panda", "<8", "<7"):
    pass
seq = rows[-2000:]
by = collections.OrderedDict()
for i, (s, e, n) in enumerate(seq):
    gap = s - seq[i - 1][1] if i else 0
    d = by.setdefault(n, [[], []]); d[0].append(e - s); d[1].append(gap)
for n, (du, ga) in by.items():
    print(f"{len(du):5d} x  dur {np.median(du) / 1e3:7.2f} us  gap before {np.median(ga) / 1e3:6.2f} us   {n}")
PY
rm -rf "$D"
