#!/usr/bin/env python3
"""Turn the rocprofv3 databases of scripts/record_pass.sh (gpurun_out/rec/) into the small summaries kept under profiles/.

    python scripts/summarize_profiles.py [gpurun_out/rec] [profiles] [r1]
"""
import csv
import json
import os
import shutil
import statistics
import sys

REC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rec"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r1"
FUSED = "lm_fused_kernel"


def read_csv(rel):
    with open(os.path.join(REC, rel), newline="") as f:
        return list(csv.DictReader(f))


def kernel_stats(grid):
    """--kernel-trace --stats: rocprofv3's own kernel_stats.csv is kept as is; the bench launches are the dispatches of
    the fused kernel with the full grid (the input generator also runs it, on S rows at a time)."""
    shutil.copy(os.path.join(REC, "kt/kt_kernel_stats.csv"), os.path.join(OUT, f"{TAG}_fused_kernel_stats.csv"))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in read_csv("kt/kt_kernel_trace.csv")
         if FUSED in r["Kernel_Name"] and int(r["Grid_Size_X"]) == grid]  # fmt: skip
    return {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d), "stdev_ns": statistics.pstdev(d)}


def counter(rel, name, grid):
    return [float(r["Counter_Value"]) for r in read_csv(rel)
            if r["Counter_Name"] == name and FUSED in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]  # fmt: skip


def main():
    os.makedirs(OUT, exist_ok=True)
    bench = json.load(open(os.path.join(REC, "bench.json")))
    cfg = bench["config"]
    grid = cfg["seeds_per_gpu"] * cfg["waypoints"]
    summary = {"kernel_trace": kernel_stats(grid)}
    fetch = counter("pmc_fetch/pmc_counter_collection.csv", "FETCH_SIZE", grid)
    write = counter("pmc_write/pmc_counter_collection.csv", "WRITE_SIZE", grid)
    for key, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        summary[key] = {"dispatches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
    # scripts/pmc_probe.py: variants A..E, three rounds; keep the last round.  Two counter passes over the same script.
    labels = ("A: K=10, no collision (random inputs)", "B: K=20, no collision (random inputs)",
              "C: K=10 + collision (random inputs, no summary)", "D: collision_masks alone (random inputs)",
              "E: the bench launch: K=10 + collision + per-seed summary, problem inputs")  # fmt: skip
    disp = [{"variant": lab} for lab in labels]
    for rel in ("pmc_valu/pmc_counter_collection.csv", "pmc_valu2/pmc_counter_collection.csv"):
        if not os.path.exists(os.path.join(REC, rel)):
            continue
        sq = {}
        for r in read_csv(rel):
            if "cppf" in r["Kernel_Name"] and int(r["Grid_Size"]) == 1024 * 256:
                e = sq.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:100],
                                                          "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})  # fmt: skip
                e[r["Counter_Name"]] = float(r["Counter_Value"])
        last = [sq[k] for k in sorted(sq)][-len(labels):]
        for d, e in zip(disp, last):
            for k, v in e.items():
                d[k if k not in ("dur_ns",) or k not in d else "dur_ns_pass2"] = v
    for d in disp:
        d["valu_per_row"] = d["SQ_INSTS_VALU"] * 64 / (1024 * 256) if "SQ_INSTS_VALU" in d else None
    summary["sq_counters_scripts_pmc_probe_last_round"] = disp
    with open(os.path.join(OUT, f"{TAG}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    key = f"{cfg['robot']}_S{cfg['seeds_per_gpu']}_W{cfg['waypoints']}_K{cfg['lm_iterations_per_step']}_coll{int(cfg['collision_fused'])}"
    fk, wk = summary["FETCH_SIZE"]["mean_KB"], summary["WRITE_SIZE"]["mean_KB"]
    traffic = {
        "_how": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --streams 1 ; rocprofv3 --pmc "
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
                     ("bench_random_inputs.json", "bench_random_inputs.json"), ("valu_issue_rate_calibration.txt", "valu_issue_rate_calibration.txt"),
                     ("bench_C2.json", "bench_C2.json"), ("bench_C3.json", "bench_C3.json"), ("bench_C5.json", "bench_C5.json"),
                     ("kbench.txt", "kbench.txt"), ("launch_model.txt", "launch_model.txt"), ("pytest_gpu.txt", "pytest_gpu.txt")):  # fmt: skip
        if os.path.exists(os.path.join(REC, src)):
            shutil.copy(os.path.join(REC, src), os.path.join(OUT, f"{TAG}_{dst}"))
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
