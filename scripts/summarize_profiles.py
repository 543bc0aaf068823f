#!/usr/bin/env python3
"""Turn the rocprofv3 databases of scripts/record_pass.sh (gpurun_out/rec/) into the small summaries kept under profiles/.

    python scripts/summarize_profiles.py [gpurun_out/rec] [profiles] [r4] [counters|copy|all]

"counters" needs only the profiler passes (it runs BEFORE the bench records of a pass, on the GPU box, so that every bench.py record
of the pass finds the executed-instruction counters of its own launch shape and library build under profiles/); "copy" copies the
text / JSON outputs of the pass; "all" (default) does both.

Writes  <tag>_fused_kernel_stats.csv (rocprofv3's own kernel_stats), <tag>_pmc_summary.json (kernel-trace duration of the bench
launch, FETCH / WRITE_SIZE, and per labelled variant of scripts/pmc_probe.py every SQ counter of the four counter passes with the
derived per-row / per-iteration figures), <tag>_traffic.json and <tag>_issue.json (what bench.py's roofline object reads), the
MFMA comparison <tag>_mfma_quad.json, and copies of the text / JSON outputs of the pass.
"""
import csv
import json
import os
import shutil
import statistics
import sys

REC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rec"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r4"
MODE = sys.argv[4] if len(sys.argv) > 4 else "all"
FUSED = "lm_fused_kernel"
PROBE_KERNELS = ("lm_fused_kernel", "lm_quad_kernel", "collision_kernel")


def read_csv(rel):
    with open(os.path.join(REC, rel), newline="") as f:
        return list(csv.DictReader(f))


def kernel_stats(grid, build_id):
    """--kernel-trace --stats: rocprofv3's own kernel_stats.csv is kept as is behind ONE comment line naming the library build it
    was taken with (bench.py's roofline.kernel_ms_profile reads it only for that build); the bench launches are the dispatches of
    the fused kernel with the full grid (the input generator also runs it, on S rows at a time)."""
    with open(os.path.join(REC, "kt/kt_kernel_stats.csv")) as src, open(os.path.join(OUT, f"{TAG}_fused_kernel_stats.csv"), "w") as dst:
        dst.write(f"# build_id={build_id}\n")
        dst.write(src.read())
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in read_csv("kt/kt_kernel_trace.csv")
         if FUSED in r["Kernel_Name"] and int(r["Grid_Size_X"]) == grid]  # fmt: skip
    return {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d), "stdev_ns": statistics.pstdev(d)}


def counter(rel, name, grid):
    return [float(r["Counter_Value"]) for r in read_csv(rel)
            if r["Counter_Name"] == name and FUSED in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]  # fmt: skip


def probe_dispatches(rel, n_variants):
    """The last round of scripts/pmc_probe.py: one dict of counters per labelled variant, in dispatch order."""
    by_id = {}
    for r in read_csv(rel):
        if any(k in r["Kernel_Name"] for k in PROBE_KERNELS):
            e = by_id.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:110], "grid": int(r["Grid_Size"]),
                                                          "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})  # fmt: skip
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by_id[k] for k in sorted(by_id)][-n_variants:]


ROWS = {"A": 262144, "B": 262144, "C": 262144, "D": 262144, "E": 262144, "F": 262144, "G": 131072, "H": 8192, "I": 8192, "J": 16384,
        "K": 16384, "L": 16384, "M": 262144, "N": 2097152, "O": 262144, "P": 262144, "Q": 262144, "R": 8192}  # fmt: skip
ITERS = {"A": 10, "B": 20, "C": 10, "D": 0, "E": 10, "F": 10, "G": 10, "H": 10, "I": 10, "J": 20, "K": 20, "L": 20, "M": 10, "N": 10,
         "O": 10, "P": 10, "Q": 10, "R": 10}  # fmt: skip
ISSUE_KEYS = {  # variant -> key of bench.py's workload_key(robot, S, W, K, collide, inputs [+ "_b<steps per launch>"])
    "E": "panda_S1024_W256_K10_coll1",
    "C": "panda_S1024_W256_K10_coll1_random",
    "G": "panda_S128_W64_K10_coll0_problem_b16",
    "M": "fetch_S512_W256_K10_coll1_problem_b2",
    "N": "chain12_S4096_W512_K10_coll1",
    "O": "panda_S128_W256_K10_coll1_problem_b8",
    "P": "panda_S256_W256_K10_coll1_problem_b4",
    "Q": "panda_S512_W256_K10_coll1_problem_b2",
}


def copy_outputs():
    os.makedirs(OUT, exist_ok=True)
    for src in ("bench.json", "bench_dist1.json", "bench_2ranks_rehearsal.json", "bench_solver_f64.json", "bench_solver_f32.json", "bench_random_solver_f32.json", "gate_census_problem.txt", "gate_census_random.txt", "bench_C2.json", "bench_C3.json",
                "bench_C5.json", "shard_streams.txt", "hwq_sweep.txt", "shard_bench.txt", "shard_bench_mfma.txt", "mfma_chain12.txt", "ksweep.txt", "dp_bench.txt", "coupled_bench.txt", "coupled_dp_kernels.txt", "lone_wave_micro.txt", "kbench.txt",
                "kbench_small.txt", "launch_model.txt", "rtc_bench.txt", "pytest_gpu.txt", "valu_issue_rate_calibration.txt", "bench_driverflags.json", "bench_shard128_driverflags.json",
                "bench_shard128_2000steps.json", "bench_2ranks_driverflags.json", "bench_2000steps.json"):  # fmt: skip
        if os.path.exists(os.path.join(REC, src)):
            shutil.copy(os.path.join(REC, src), os.path.join(OUT, f"{TAG}_{src}"))


def main():
    os.makedirs(OUT, exist_ok=True)
    # the headline workload (BASELINE config 4) and the library build every record of this pass is keyed by
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cppflow_amd import build as hip_build

    cfg = {"robot": "panda", "seeds_per_gpu": 1024, "waypoints": 256, "lm_iterations_per_step": 10, "collision_fused": True, "ndof": 7}
    grid = cfg["seeds_per_gpu"] * cfg["waypoints"]
    build_id = hip_build.built_id()
    assert build_id == hip_build.source_hash(), "the library on disk is not the one these sources produce"
    summary = {"library_build_id": build_id, "kernel_trace": kernel_stats(grid, build_id)}
    fetch = counter("pmc_fetch/pmc_counter_collection.csv", "FETCH_SIZE", grid)
    write = counter("pmc_write/pmc_counter_collection.csv", "WRITE_SIZE", grid)
    for key, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        summary[key] = {"dispatches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
    labels = json.load(open(os.path.join(REC, "pmc_labels.json")))
    disp = [{"variant": lab} for lab in labels]
    for p in ("pmc_p1", "pmc_p2", "pmc_p3", "pmc_p4"):
        rel = f"{p}/pmc_counter_collection.csv"
        if not os.path.exists(os.path.join(REC, rel)):
            continue
        for d, e in zip(disp, probe_dispatches(rel, len(labels))):
            for k, v in e.items():
                if k == "dur_ns":
                    d.setdefault("dur_ns_by_pass", {})[p] = v
                elif k in ("kernel", "grid"):
                    assert d.setdefault(k, v) == v, (d["variant"], k, d[k], v)  # the four passes see the same dispatches
                else:
                    d[k] = v
    issue, SIMDS, GHZ = {}, 1024, 2.4
    for d in disp:
        tag = d["variant"][0]
        rows, its = ROWS[tag], ITERS[tag]
        if "SQ_INSTS_VALU" in d:
            d["valu_per_row"] = d["SQ_INSTS_VALU"] * 64 / rows
            d["valu_issue_us_at_peak"] = d["SQ_INSTS_VALU"] * 2 / SIMDS / (GHZ * 1e3)
            d["valu_issue_frac"] = d["valu_issue_us_at_peak"] / (d["dur_ns_by_pass"]["pmc_p1"] * 1e-3)
        if "SQ_INSTS_VALU_FMA_F32" in d and "SQ_INSTS_VALU" in d:
            f32 = 2 * d["SQ_INSTS_VALU_FMA_F32"] + d["SQ_INSTS_VALU_ADD_F32"] + d["SQ_INSTS_VALU_MUL_F32"] + d["SQ_INSTS_VALU_TRANS_F32"]
            f64 = 2 * d.get("SQ_INSTS_VALU_FMA_F64", 0) + d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0)
            d["executed_flops_per_row"] = (f32 + f64) * 64 / rows
            d["flops_per_valu_lane_op"] = (f32 + f64) / d["SQ_INSTS_VALU"]
            d["executed_tflops"] = (f32 + f64) * 64 / (d["dur_ns_by_pass"]["pmc_p1"] * 1e-9) / 1e12
        if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
            d["issue_stall_frac_of_wave_cycles"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
        if tag in ISSUE_KEYS and "flops_per_valu_lane_op" in d:
            issue[ISSUE_KEYS[tag]] = {
                "valu_insts_per_launch": d["SQ_INSTS_VALU"],
                "flops_per_valu_lane_op": round(d["flops_per_valu_lane_op"], 4),
                "fma_f32": d["SQ_INSTS_VALU_FMA_F32"], "add_f32": d["SQ_INSTS_VALU_ADD_F32"], "mul_f32": d["SQ_INSTS_VALU_MUL_F32"],
                "trans_f32": d["SQ_INSTS_VALU_TRANS_F32"], "kernel_us_profiled": d["dur_ns_by_pass"]["pmc_p1"] * 1e-3,
                "variant": d["variant"],
            }  # fmt: skip
    by = {d["variant"][0]: d for d in disp}
    if "A" in by and "B" in by and "SQ_INSTS_VALU" in by["A"]:
        summary["row_shape_valu_per_row_iteration"] = (by["B"]["SQ_INSTS_VALU"] - by["A"]["SQ_INSTS_VALU"]) * 64 / ROWS["A"] / 10
    summary["sq_counters_scripts_pmc_probe_last_round"] = disp
    with open(os.path.join(OUT, f"{TAG}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    issue["_build_id"] = build_id
    issue["_how"] = ("rocprofv3 --pmc SQ_INSTS_VALU ... / --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 "
                     "-- python3 scripts/pmc_probe.py (scripts/record_pass.sh): wave-instruction counts of ONE launch of the matching workload.  "
                     "flops_per_valu_lane_op = (2 FMA + ADD + MUL + TRANS) / VALU: executed flops per VALU lane-operation, measured, not modelled.")  # fmt: skip
    with open(os.path.join(OUT, f"{TAG}_issue.json"), "w") as f:
        json.dump(issue, f, indent=1)
    # the MFMA question: quad shape, J J^T on the VALU (rotated operands, FMAs) vs v_mfma_f32_4x4x1
    mf = {}
    for a, b, what in (("H", "I", "C2: Panda 128 x 64 = 8192 rows, K = 10, no collision"), ("K", "L", "Panda 64 x 256 = 16384 rows, K = 20, no collision")):
        if a in by and b in by and "SQ_INSTS_VALU" in by[a]:
            keep = ("dur_ns_by_pass", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
                    "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES",
                    "SQ_INSTS_VALU_MFMA_F32", "SQ_WAIT_ANY", "valu_per_row", "issue_stall_frac_of_wave_cycles")  # fmt: skip
            its = ITERS[a]
            mf[what] = {
                "valu": {k: by[a].get(k) for k in keep},
                "mfma": {k: by[b].get(k) for k in keep},
                "us_per_launch_valu": by[a]["dur_ns_by_pass"]["pmc_p1"] * 1e-3,
                "us_per_launch_mfma": by[b]["dur_ns_by_pass"]["pmc_p1"] * 1e-3,
                "valu_insts_per_row_iteration_valu": by[a]["SQ_INSTS_VALU"] * 64 / ROWS[a] / its / 4 * 4,
                "valu_insts_per_row_iteration_mfma": by[b]["SQ_INSTS_VALU"] * 64 / ROWS[b] / its / 4 * 4,
                "note": "per ROW, i.e. summed over the four lanes of the quad: divide by 4 for the instructions one wavefront issues per iteration",
            }
    with open(os.path.join(OUT, f"{TAG}_mfma_quad.json"), "w") as f:
        json.dump(mf, f, indent=1)
    key = f"{cfg['robot']}_S{cfg['seeds_per_gpu']}_W{cfg['waypoints']}_K{cfg['lm_iterations_per_step']}_coll{int(cfg['collision_fused'])}"
    fk, wk = summary["FETCH_SIZE"]["mean_KB"], summary["WRITE_SIZE"]["mean_KB"]
    traffic = {
        "_build_id": build_id,
        "_how": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-siblings --streams 1 ; "
        "rocprofv3 --pmc WRITE_SIZE -- (same): scripts/record_pass.sh.  Mean over the lm_fused_kernel dispatches.  FETCH_SIZE (KB) is doubled "
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
    print(json.dumps({k: v for k, v in summary.items() if k != "sq_counters_scripts_pmc_probe_last_round"}, indent=1))
    for d in disp:
        print(d["variant"][:60].ljust(62), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()
                                            if k in ("valu_per_row", "valu_issue_frac", "flops_per_valu_lane_op", "executed_tflops", "issue_stall_frac_of_wave_cycles")})  # fmt: skip


if __name__ == "__main__":
    if MODE in ("counters", "all"):
        main()
    if MODE in ("copy", "all"):
        copy_outputs()
