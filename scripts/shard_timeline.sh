# developer measurement: per-queue timeline of the shard regime (32 768-row launches, four in flight) from a rocprofv3 kernel trace:
# kernel durations, the gap between consecutive kernels of one queue, and how many kernels run at any time.
set -e
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/shard_tl"; mkdir -p "$OUT"; export TMPDIR=/tmp
SEEDS="${1:-128}"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --seeds $SEEDS --steps 1024 --warmup 128 --streams 4 --no-cpu-baseline --no-siblings --repeats 1 > "$OUT/bench.json" 2> "$OUT/stderr.txt")
python3 - "$OUT/kt" <<'PY'
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lm_fused_kernel" in r["Kernel_Name"]]
print(len(rows), "fused launches; columns:", list(rows[0].keys()))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
alls = []
for q, v in byq.items():
    v.sort()
    v = v[len(v) // 4:]  # the timed region, roughly (the pre-warm and warm-up come first)
    d = np.array([e - s for s, e in v]) / 1e3
    g = np.array([v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]) / 1e3
    g = g[g < 100]
    print(f"queue {q}: {len(v)} kernels  duration median {np.median(d):.2f} us (p10 {np.percentile(d,10):.2f}, p90 {np.percentile(d,90):.2f})   gap to the next on the same queue median {np.median(g):.2f} us (p10 {np.percentile(g,10):.2f}, p90 {np.percentile(g,90):.2f})")
    alls += v
alls.sort()
t0, t1 = alls[len(alls) // 4][0], alls[-len(alls) // 8][1]
ev = sorted([(s, 1) for s, e in alls] + [(e, -1) for s, e in alls])
cur, last, hist = 0, None, collections.Counter()
for t, dlt in ev:
    if last is not None and t0 <= last and t <= t1:
        hist[cur] += t - last
    cur += dlt
    last = t
tot = sum(hist.values())
print("kernels in flight (fraction of time):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
PY
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('bench under the profiler: us/step %.2f' % (d['ms_per_step']*1e3))"
