#!/usr/bin/env python3
"""Dispatch timeline of the driver's own bench command under `rocprofv3 --kernel-trace` -> profiles/<round>_overlap.json.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap -o kt -- python3 $ROOT/bench.py --gpus 1 --steps 20 \
        --warmup 5 --no-cpu-baseline --no-siblings > $OUT/overlap_stdout.txt
    python3 scripts/overlap_summary.py $OUT/overlap $OUT/overlap_stdout.txt profiles/r5_overlap.json [steps]

bench.py's headline `ms_per_step` (two launches in flight on two streams) is BELOW the fused kernel's own duration; this shows why
with the profiler's begin / end stamps of every dispatch: the timed regions are the runs of exactly `steps` fused dispatches with no
idle gap between them; per region: the union-busy time (any fused dispatch resident), per step = union / steps, the mean duration of
a dispatch under overlap, and the share of the busy time with two dispatches resident."""
import csv
import glob
import json
import sys

import numpy as np

KERNEL = "lm_fused_kernel"
GAP_NS = 1_500  # a region ends where NO fused dispatch is resident for this long: inside a timed region two launches are in flight
# all the time (the host enqueues a step in ~5 us, the GPU takes ~37), between regions the host synchronises and reads its clock


def load(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    return rows


def clusters(rows):
    out, cur, busy_until = [], [], None
    for r in rows:
        if cur and r[0] > busy_until + GAP_NS:
            out.append(cur)
            cur = []
        busy_until = r[1] if not cur else max(busy_until, r[1])
        cur.append(r)
    if cur:
        out.append(cur)
    return out


def region_stats(c):
    ev = sorted([(s, +1) for s, _, _, _ in c] + [(e, -1) for _, e, _, _ in c])
    t_prev, depth, at = ev[0][0], 0, {}
    for t, d in ev:
        at[depth] = at.get(depth, 0) + (t - t_prev)
        depth += d
        t_prev = t
    union = sum(v for k, v in at.items() if k >= 1)
    return {"dispatches": len(c), "begin_ns": c[0][0], "span_us": (max(e for _, e, _, _ in c) - c[0][0]) / 1e3, "union_busy_us": union / 1e3,
            "idle_inside_us": at.get(0, 0) / 1e3, "two_or_more_resident_frac": sum(v for k, v in at.items() if k >= 2) / max(union, 1),
            "max_resident": max(at), "mean_dispatch_us": float(np.mean([e - s for s, e, _, _ in c])) / 1e3,
            "queues": sorted({q for _, _, q, _ in c}), "dispatch_begin_end_us": [[(s - c[0][0]) / 1e3, (e - c[0][0]) / 1e3] for s, e, _, _ in c]}


def main():
    d, stdout_path, out_path = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    rows = load(d)
    with open(out_path.replace(".json", "_dispatches.csv"), "w") as f:  # (kept beside the summary: begin, end, queue of every fused dispatch)
        f.write("start_ns,end_ns,queue\n" + "".join(f"{s - rows[0][0]},{e - rows[0][0]},{q}\n" for s, e, q, _ in rows))
    cl = clusters(rows)
    line = None
    for ln in open(stdout_path):
        if ln.startswith('{"metric"'):
            line = json.loads(ln)
    # the timed regions: exactly `steps` dispatches, more than one queue (the isolated-launch loop behind roofline.kernel_ms is one queue)
    regions = [region_stats(c) for c in cl if len(c) == steps]
    timed = [r for r in regions if r["max_resident"] >= 2]
    # the isolated launches behind roofline.kernel_ms: dispatches that never share the chip with another fused dispatch (the last 400)
    alone = [c for c in cl if region_stats(c)["max_resident"] == 1]
    iso = [e - s for c in alone for s, e, _, _ in c][-400:]
    rec = {
        "what": "rocprofv3 --kernel-trace of `python3 bench.py --gpus 1 --steps %d --warmup 5 --no-cpu-baseline --no-siblings`: every timed region "
                "(a run of exactly %d overlapped lm_fused_kernel dispatches on two queues)" % (steps, steps),
        "library_build_id": line["roofline"]["library_build_id"] if line else None,
        "bench_ms_per_step_under_profiler": line["ms_per_step"] if line else None,
        "bench_kernel_ms_isolated_under_profiler": line["roofline"]["kernel_ms"] if line else None,
        "fused_dispatches_total": len(rows), "clusters": len(cl), "cluster_sizes": sorted({len(c) for c in cl}), "timed_regions": len(timed),
        "isolated_dispatch_us_mean": float(np.mean(iso)) / 1e3 if iso else None,
    }
    if timed:
        rec.update({
            "union_busy_us_per_step": float(np.median([r["union_busy_us"] / steps for r in timed])),
            "span_us_per_step": float(np.median([r["span_us"] / steps for r in timed])),
            "mean_dispatch_us_under_overlap": float(np.median([r["mean_dispatch_us"] for r in timed])),
            "two_resident_frac_of_busy_time": float(np.median([r["two_or_more_resident_frac"] for r in timed])),
            "kernel_ms_overlapped": float(np.median([r["union_busy_us"] / steps for r in timed])) / 1e3,
            "regions": timed,
        })
        if line:
            rec["union_busy_over_bench_ms_per_step"] = rec["union_busy_us_per_step"] / (1e3 * line["ms_per_step"])
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "regions"}, indent=1))


if __name__ == "__main__":
    main()
