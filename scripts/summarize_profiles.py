#!/usr/bin/env python3
"""Turn the rocprofv3 databases of scripts/record_pass.sh (gpurun_out/rec/) into the small summaries kept under profiles/.

    python scripts/summarize_profiles.py [gpurun_out/rec] [profiles] [r1]
"""
import csv
import json
import os
import shutil
import sqlite3
import statistics
import sys

REC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rec"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r1"
FUSED = "lm_fused_kernel"


def rows(db, sql):
    con = sqlite3.connect(os.path.join(REC, db))
    cur = con.execute(sql)
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur]


def kernel_stats():
    """--kernel-trace --stats: per-kernel call count / total / mean / min / max (ns), like rocprofv3's kernel_stats.csv."""
    ks = rows("kt/kt_results.db", "select name, start, end from kernels")
    by = {}
    for k in ks:
        by.setdefault(k["name"], []).append(k["end"] - k["start"])
    total = sum(sum(v) for v in by.values())
    path = os.path.join(OUT, f"{TAG}_fused_kernel_stats.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v),
                        round(statistics.pstdev(v), 3)])  # fmt: skip
    fused = [v for n, v in by.items() if FUSED in n]
    return {"calls": len(fused[0]), "avg_ns": sum(fused[0]) / len(fused[0]), "min_ns": min(fused[0])} if fused else None


def counter(db, name):
    r = rows(db, f"select kernel_name, counter_name, value, duration from counters_collection where counter_name = '{name}'")
    return [x for x in r if FUSED in x["kernel_name"]]


def main():
    os.makedirs(OUT, exist_ok=True)
    summary = {"kernel_trace": kernel_stats()}
    fetch = [x["value"] for x in counter("pmc_fetch/pmc_results.db", "FETCH_SIZE")]
    write = [x["value"] for x in counter("pmc_write/pmc_results.db", "WRITE_SIZE")]
    for key, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        summary[key] = {"dispatches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
    # scripts/pmc_probe.py: variants A..D, three rounds; keep the last round
    sq = {}
    for cname in ("SQ_INSTS_VALU", "SQ_WAVES"):
        for x in rows("pmc_valu/pmc_results.db", "select dispatch_id, kernel_name, counter_name, value, duration from "
                      f"counters_collection where counter_name = '{cname}' order by dispatch_id"):  # fmt: skip
            if "cppf" in x["kernel_name"]:
                sq.setdefault(x["dispatch_id"], {"name": x["kernel_name"][:100], "dur_ns": x["duration"]})[cname] = x["value"]
    disp = [sq[k] for k in sorted(sq)][-4:]
    for d, label in zip(disp, ("A: K=10, no collision", "B: K=20, no collision", "C: K=10 + collision (bench launch, no summary)",
                               "D: collision_masks alone")):  # fmt: skip
        d["variant"] = label
        d["valu_per_row"] = d["SQ_INSTS_VALU"] * 64 / (1024 * 256) if "SQ_INSTS_VALU" in d else None
    summary["sq_counters_scripts_pmc_probe_last_round"] = disp
    with open(os.path.join(OUT, f"{TAG}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    bench = json.load(open(os.path.join(REC, "bench.json")))
    cfg = bench["config"]
    key = f"{cfg['robot']}_S{cfg['seeds_per_gpu']}_W{cfg['waypoints']}_K{cfg['lm_iterations_per_step']}_coll{int(cfg['collision_fused'])}"
    fk, wk = summary["FETCH_SIZE"]["mean_KB"], summary["WRITE_SIZE"]["mean_KB"]
    traffic = {
        "_how": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 ; rocprofv3 --pmc "
        "WRITE_SIZE -- (same): scripts/record_pass.sh.  Mean over the lm_fused_kernel dispatches.  FETCH_SIZE (KB) is doubled "
        "(gfx950 reports half the bytes of a coalesced stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE (KB) is exact.",
        key: {
            "FETCH_SIZE_KB": round(fk, 2),
            "WRITE_SIZE_KB": round(wk, 2),
            "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024, -3)),
            "algorithmic_bytes_per_launch": int(cfg["seeds_per_gpu"] * cfg["waypoints"] * (8 * cfg["ndof"] + 28 + 6)),
            "note": "read side = x_in + the shared [W,7] target; write side = x_out + the packed per-row outputs + the [S,8] summary.",
        },
    }
    with open(os.path.join(OUT, f"{TAG}_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=2)
    for src, dst in (("bench.json", "bench.json"), ("bench_streams1.json", "bench_streams1.json"), ("bench_dist1.json", "bench_dist1.json"),
                     ("bench_C2.json", "bench_C2.json"), ("bench_C3.json", "bench_C3.json"), ("bench_C5.json", "bench_C5.json"),
                     ("kbench.txt", "kbench.txt"), ("pytest_gpu.txt", "pytest_gpu.txt")):  # fmt: skip
        if os.path.exists(os.path.join(REC, src)):
            shutil.copy(os.path.join(REC, src), os.path.join(OUT, f"{TAG}_{dst}"))
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
