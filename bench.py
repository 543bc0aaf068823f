#!/usr/bin/env python3
"""bench.py -- LM-IK iterations/s of the fused MI355X hot path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch of synthetic seeds: a single launch of the fused kernel
(cppf_lm_pose_steps) doing `--lm-steps` K iterations of { pose-only LM step ; clamp to joint limits } on every
(seed, waypoint) row, then the pose-error metrics, self / environment collision masks, joint-limit mask and search cost of
the result, and the per-seed summary reduction (8 floats per seed: the x_is_valid maxima, collision counts, summed cost).
For N > 1 the summaries of `--gather-every` consecutive steps are all-gathered over RCCL on an auxiliary stream and CONSUMED:
every rank runs x_is_valid's seed selection (cppf_select_valid_seed_gathered, cppflow/optimization_utils.py:856-909) over all
ranks' seeds for every step, and a ring slot is reused only after that selection has completed.  value = rows * K * steps /
wall-time over all ranks.

Launching.  `python bench.py --gpus N` with WORLD_SIZE unset starts N fresh rank processes itself -- BEFORE this process
touches the GPU -- relays rank 0's single JSON line and exits with the children's status; under torch.distributed.run
(WORLD_SIZE set) it is one rank.

Scaling.  N = 1: BASELINE.json configs[3] on one GPU -- Panda (7-DoF), 1024 seeds x 256 waypoints, the two cuboids of
panda__2cubes -- the configuration the metric is quoted on.  N > 1 defaults to STRONG scaling, the configuration
BASELINE.json names ("1024 seeds x 256 waypoints, seed-sharded across 2/4/8 MI355X"): the same 1024 seeds split by
`distributed.seed_shard`; the weak-scaling figure (1024 seeds per GPU) is measured in the same run and reported as the
sibling key `weak_scaling`.  `--scaling weak` makes the weak figure the headline instead.

Inputs are already resident in HBM when the timed region starts: the target path is the named reference
problem's (panda__2cubes resampled to 256 waypoints; committed fixture), the seeds are synthetic -- per seed an IK branch
tracking the path, x0 = clamp(q*_s + 0.1 randn) (the construction of the reference's tests/optimization_test.py:82).
`--inputs random` switches to the 8d fall-back (independent q* ~ U(limits) per waypoint, tests/optimization_test.py:136-137),
the worst case for the wave-uniform collision broad phase; at N = 1 that figure is also reported as the sibling key
`random_inputs`, and the one-stream figure as `one_stream`.

Timing: `--prewarm-ms` (60) of untimed launches bring the GPU to its sustained clocks, then W untimed warm-up steps, then
exactly K steps: barrier + synchronize, clock, the K steps (+ the exchange of a partly filled bucket), this rank's synchronize,
clock; the maximum over ranks is reported.  The group's CLOSING barrier comes after the clock has stopped (its cost is recorded in
config.timed_region.closing_barrier_us): at the driver's `--steps 20` a sharded region is ~100 us long, and an 8-rank barrier
inside it would be most of what is timed.  The region is measured `--repeats` (5) times back to back and the MEDIAN is reported; a
region shorter than 0.4 ms `--short-region-repeats` (16) more times (behind a pre-warm and warm-up of their own) -- every repetition exactly K steps (all of them in
config.timed_region.ms_per_step_all).  Consecutive steps are independent batches (a ring of output-buffer sets) alternating
between `--streams` HIP streams.

Steps per launch (`--batch`, cppf_lm_batch_*).  A launch of the fused kernel carries B consecutive steps -- B independent problems
laid end to end in one grid, each with its own outputs, bit for bit what B separate launches produce.  B = how many of this rank's
steps make one full-width launch (262 144 rows, four wavefronts per SIMD), at most 16: ONE at N = 1 (the 1024-seed step IS a
full-width launch; it goes through the same batch entry point), 2 / 4 / 8 for the 131 072 / 65 536 / 32 768-row shards of N = 2 / 4 /
8 -- so every GPU issues launches of the same width at every N instead of small launches whose only overlap is the four hardware
queues.  A timed region of K steps is floor(K / B) full launches and one launch of the K mod B steps left over.

Prints ONE JSON line (rank 0).
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

torch = None  # imported by main() AFTER the launcher decision: the parent of an N > 1 run never loads a GPU runtime


def _ensure_torch():
    """helpers below are also imported by scripts/ (which never go through main()'s launcher decision)"""
    global torch
    if torch is None:
        import torch as _torch

        torch = _torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak == fp32-input MFMA peak
N_SIMD = 1024  # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md: max clock; one VALU wave-instruction occupies a SIMD for 2 cycles


def algorithmic_flops_per_row_iter(d: int) -> float:
    """FK 130*(d + n_fixed) + Jacobian 12d + pose error ~100 + scaling 6(d+1) + J^T J upper triangle
    12*d(d+1)/2 + J^T e 12d + Cholesky d^3/3 + 2d^2 + update/clamp 3d   (d=7: ~1.9 kFLOP; the figure the survey states)."""
    n_fixed = 1
    return (
        130.0 * (d + n_fixed) + 12 * d + 100 + 6 * (d + 1) + 12 * d * (d + 1) / 2 + 12 * d + d**3 / 3 + 2 * d * d + 3 * d
    )


def algorithmic_flops_collision(L: int, P: int, O: int) -> float:
    """capsule end points 36 L + pairs 90 P_s + capsule-cuboid 150 L O."""
    return 36.0 * L + 90.0 * P + 150.0 * L * O


def algorithmic_bytes_per_row(d: int, collide: bool) -> float:
    """read x 4d + read target 28 + write x 4d (+2 mask bytes + 4 cost bytes when collision is fused)."""
    return 8.0 * d + 28.0 + (6.0 if collide else 0.0)


def _profile_record(fname, key):
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(key)


def workload_key(robot, S, W, K, collide, inputs="problem"):
    return f"{robot}_S{S}_W{W}_K{K}_coll{int(collide)}" + ("" if inputs == "problem" else f"_{inputs}")


PROFILE_ROUND = "r4"  # the committed record pass these lookups read (scripts/record_pass.sh -> profiles/r4_*)


def traffic_from_profiles(robot, S, W, K, collide, build_id):
    """HBM bytes per launch of the fused kernel from the rocprofv3 PMC passes committed under profiles/ (separate
    --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  bench.py cannot profile itself, so the figure is the recorded one for the matching workload AND library
    build (the record carries the build id of the library it was taken with), else None."""
    rec = _profile_record(f"{PROFILE_ROUND}_traffic.json", workload_key(robot, S, W, K, collide))
    if rec and _profile_record(f"{PROFILE_ROUND}_traffic.json", "_build_id") == build_id:
        return rec.get("hbm_bytes_per_launch")
    return None


def issue_record_from_profiles(robot, S, W, K, collide, inputs, build_id):
    """({"valu_insts_per_launch": SQ_INSTS_VALU of one fused launch (wave-instructions), "flops_per_valu_lane_op": executed
    flops per VALU lane-operation (FMA = 2, mul / add / sub = 1, everything else 0), ...}, note) recorded by `rocprofv3 --pmc` for
    the matching workload (scripts/record_pass.sh -> profiles/r3_issue.json).  A record is only used when it was taken with THIS
    build of the library (`_build_id` in the file == cppf_build_id()): after any kernel change the counts are stale, and a stale
    instruction count divided by a live kernel time is not a measurement.  Returns (None, why) otherwise."""
    fname = f"{PROFILE_ROUND}_issue.json"
    rec = _profile_record(fname, workload_key(robot, S, W, K, collide, inputs))
    if rec is None:
        return None, f"no SQ_INSTS_VALU record for this workload in profiles/{fname}"
    have = _profile_record(fname, "_build_id")
    if have != build_id:
        return None, f"profiles/{fname} was recorded with library build {have}, this run loaded {build_id}"
    return rec, ""


def kernel_profile_from_profiles(kernel_substr, build_id):
    """Average duration (ms) and call count of the dominant kernel in the committed `rocprofv3 --kernel-trace --stats` summary of
    this same command (profiles/r3_fused_kernel_stats.csv; its first line names the build it was taken with), or (None, why)."""
    import csv

    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_fused_kernel_stats.csv")
    if not os.path.exists(path):
        return None, f"profiles/{PROFILE_ROUND}_fused_kernel_stats.csv missing"
    with open(path) as f:
        first = f.readline()
        if not first.startswith("# build_id="):
            return None, "stats file carries no build id"
        have = first.strip().split("=", 1)[1]
        rows = list(csv.DictReader(f))
    if have != build_id:
        return None, f"stats recorded with library build {have}, this run loaded {build_id}"
    best = None
    for r in rows:
        if kernel_substr in r["Name"] and (best is None or int(r["Calls"]) > int(best["Calls"])):
            best = r
    if best is None:
        return None, f"no row matching {kernel_substr!r}"
    return {"ms": float(best["AverageNs"]) * 1e-6, "calls": int(best["Calls"]), "min_ms": float(best["MinNs"]) * 1e-6,
            "max_ms": float(best["MaxNs"]) * 1e-6}, ""


def make_inputs(robot, S, W, device, seed):
    _ensure_torch()
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo = torch.tensor([l for l, _ in robot.actuated_joints_limits], dtype=torch.float32)
    hi = torch.tensor([u for _, u in robot.actuated_joints_limits], dtype=torch.float32)
    q_star = lo + (hi - lo) * torch.rand((W, robot.ndof), generator=g)
    target = robot.forward_kinematics(q_star.to(device))  # [W,7]
    g2 = torch.Generator(device="cpu").manual_seed(1000 + seed)
    x0 = q_star[None] + 0.1 * torch.randn((S, W, robot.ndof), generator=g2)
    x0 = torch.minimum(torch.maximum(x0, lo), hi).reshape(S * W, robot.ndof).contiguous()
    return x0.to(device), target.contiguous()


PROBLEM_PATHS = {  # tests/golden/reference_paths.npz: the target paths of the problems BASELINE.json's configs name
    ("panda", 64): "panda__1cube_first64",
    ("fetch", 256): "fetch__hello_first256",
    ("panda", 256): "panda__2cubes_resampled256",
}


def make_inputs_problem(robot, S, W, device, seed):
    """SURVEY.md 8(d) inputs: the target path of the reference problem the configuration names (committed fixture; the 12-DoF
    chain has no reference problem: target = FK of a smooth random walk q*_{t+1} = clamp(q*_t + 0.02 randn)) and, per seed, a
    distinct IK branch q*_s that tracks the path (waypoint 0 solved by damped LM from a U(limits) start, every later waypoint
    warm-started from its predecessor, a branch that loses the path continuing on one that did not -- what IKFlow + dp_search
    hand to the optimiser), then x0 = clamp(q*_s + 0.1 randn)
    (the construction of the reference's tests/optimization_test.py:82).  Returns (x0 [S*W,d], target [W,7], description)."""
    _ensure_torch()
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo = torch.tensor([l for l, _ in robot.actuated_joints_limits], dtype=torch.float32)
    hi = torch.tensor([u for _, u in robot.actuated_joints_limits], dtype=torch.float32)
    d = robot.ndof
    key = PROBLEM_PATHS.get((robot.name, W))
    if key is not None:
        z = np.load(os.path.join(ROOT, "tests", "golden", "reference_paths.npz"))
        target = torch.tensor(z[key], dtype=torch.float32, device=device).contiguous()
        what = f"target path = {key} (reference problem, tests/golden/reference_paths.npz)"
    else:
        q = torch.empty((W, d), dtype=torch.float32)
        q[0] = lo + (hi - lo) * torch.rand(d, generator=g)
        steps = 0.02 * torch.randn((W, d), generator=g)
        for t in range(1, W):
            q[t] = torch.minimum(torch.maximum(q[t - 1] + steps[t], lo), hi)
        target = robot.forward_kinematics(q.to(device)).contiguous()
        what = "target path = FK of a smooth random walk (q*_{t+1} = clamp(q*_t + 0.02 randn))"
    lo_d, hi_d = lo.to(device), hi.to(device)
    branch = torch.empty((S, W, d), dtype=torch.float32, device=device)
    # Fetch: the lift joint is a pure z translation at the root of the chain (torso_lift_link is unrotated w.r.t. the world,
    # cppflow/data_type_utils.py:65-73), so a seed is a lift height -- drawn once per seed from the middle 80 % of its range, as a
    # sampler of whole-body configurations would -- and an ARM branch tracking the path lowered by that height (the 7-joint chain of
    # fetch_arm).  Tracking the path with the pose-only LM step on all 8 joints instead lets the lift joint, whose Jacobian column
    # is a whole metre per unit, take every vertical motion: the branches drift onto its limits, the clamp of
    # cppflow/optimization.py:259 pins them there, and half the rows of a batch built that way can no longer converge (round 3's
    # C3 inputs: 55 %) -- a property of those inputs, not of any kernel.
    ik_robot, lift = robot, None
    if robot.name == "fetch":
        from cppflow_amd.robots import get_robot as _get_robot

        ik_robot = _get_robot("fetch_arm")
        lift = (lo[0] + (hi[0] - lo[0]) * (0.1 + 0.8 * torch.rand(S, generator=g))).to(device)
    d_ik = ik_robot.ndof
    lo_ik = torch.tensor([l for l, _ in ik_robot.actuated_joints_limits], dtype=torch.float32)
    hi_ik = torch.tensor([u for _, u in ik_robot.actuated_joints_limits], dtype=torch.float32)

    def seeds_target(w):
        """[S, 7]: waypoint w as every seed's IK problem sees it (row r of a launch with W = n uses target row r)"""
        t = target[w : w + 1].repeat(S, 1)
        if lift is not None:
            t[:, 2] -= lift
        return t.contiguous()

    # waypoint 0: damped LM from random starts, re-drawing the seeds that did not reach the pose (up to 12 rounds)
    x = torch.empty((S, d_ik), dtype=torch.float32, device=device)
    todo = torch.ones(S, dtype=torch.bool, device=device)
    t0 = seeds_target(0)
    for _ in range(12):
        start = (lo_ik + (hi_ik - lo_ik) * (0.1 + 0.8 * torch.rand((S, d_ik), generator=g))).to(device).contiguous()
        r = ik_robot.lm_pose_steps(start, t0, 1e-2, 3.5, 0.35, n_steps=60)
        r = ik_robot.lm_pose_steps(r["x"], t0, 1e-6, 3.5, 0.35, n_steps=10, want_errors=True)
        ok = (r["pos_err_m"] < 1e-4) & (r["rot_err_rad"] < 1.75e-3)
        take = todo & ok
        x[take] = r["x"][take]
        todo &= ~ok
        if not bool(todo.any()):
            break
    x[todo] = r["x"][todo]
    x = x.contiguous()
    gd = torch.Generator(device=device).manual_seed(seed + 17)
    for w in range(W):
        r = ik_robot.lm_pose_steps(x, seeds_target(w), 1e-6, 3.5, 0.35, n_steps=8, want_errors=True)
        x = r["x"]
        # a branch that loses the path (runs into a joint limit) continues on a branch that did not (with the donor's lift height)
        ok = (r["pos_err_m"] < 1e-4) & (r["rot_err_rad"] < 1.75e-3)
        donors = torch.nonzero(ok).reshape(-1)
        if 0 < donors.numel() < S:
            pick = donors[torch.randint(donors.numel(), (S,), generator=gd, device=device)]
            x = torch.where(ok[:, None], x, x[pick]).contiguous()
            if lift is not None:
                lift = torch.where(ok, lift, lift[pick]).contiguous()
        branch[:, w] = x if lift is None else torch.cat([lift[:, None], x], dim=1)
    noise = 0.1 * torch.randn((S, W, d), generator=g)
    x0 = torch.minimum(torch.maximum(branch + noise.to(device), lo_d), hi_d).reshape(S * W, d).contiguous()
    return x0, target, what + "; seeds = per-seed IK branch tracking the path + 0.1 randn"


def _cpu_inputs(chain, W, d, S_cpu, seed=0):
    rng = np.random.RandomState(seed)
    q_star = rng.uniform(chain.lo, chain.hi, size=(W, d)).astype(np.float32)
    x0 = np.clip(q_star[None] + 0.1 * rng.randn(S_cpu, W, d), chain.lo, chain.hi).reshape(S_cpu * W, d).astype(np.float32)
    return q_star, x0


def cpu_baseline_torch(robot_name, obstacles, d, W, K, budget_s=12.0):
    """The reference-equivalent CPU path: oracle/ref_torch.py issues the reference's own torch op sequence
    (cppflow/optimization.py:73-92: in-place row scaling, bmm x2, eye.repeat, torch.linalg.solve, python-loop clamp;
    cppflow/collision_detection.py:27-69: distance tensors -> min -> "< 0") with batched-torch kinematics standing in for
    the un-vendored jrl.  fp32, torch's default CPU threads, on a bounded sample of the same workload."""
    _ensure_torch()
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import ref_torch

    spec = ROBOT_SPECS[robot_name]()
    chain = canonicalize(spec)
    rb = ref_torch.TorchRobot(spec, device="cpu", dtype=torch.float32)
    cub = [torch.tensor(c) for c, _ in obstacles]
    Ts = [torch.tensor(T) for _, T in obstacles]
    eps_r, eps_p = float(np.deg2rad(1.5)), 0.03

    def run(S_cpu):
        q_star, x0 = _cpu_inputs(chain, W, d, S_cpu)
        target = rb.forward_kinematics(torch.tensor(q_star)).repeat(S_cpu, 1)
        x = torch.tensor(x0)
        t0 = time.perf_counter()
        x = ref_torch.lm_pose_steps(rb, x, target, K)
        ref_torch.calculate_pose_error_m_rad(rb, x, target)
        ref_torch.q_costs_external(rb, x.reshape(S_cpu, W, d), cub, Ts, eps_r, eps_p)
        return time.perf_counter() - t0

    # torch's default (all cores) is far from optimal for these small batched ops on a many-core host: probe a few
    # thread counts on a small sample and time the bounded sample with the best one
    run(1)
    cores = os.cpu_count() or 1
    best_threads, t_probe = None, None
    for th in sorted({min(cores, c) for c in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(th)
        tt = run(32)
        if t_probe is None or tt < t_probe:
            best_threads, t_probe = th, tt
    torch.set_num_threads(best_threads)
    S_cpu = int(max(32, min(8192, 32 * budget_s / max(t_probe, 1e-6))))
    t = run(S_cpu)
    threads = torch.get_num_threads()
    return {
        "value": S_cpu * W * K / t,
        "unit": "LM-IK iterations/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{S_cpu} seeds x {W} waypoints x {K} LM iterations + pose metrics + collision masks + cost; torch-CPU "
        f"restatement of the reference's op sequence (oracle/ref_torch.py; jrl is not vendored so the reference itself "
        f"cannot run), fp32, {threads} torch threads of {os.cpu_count()} host cores, {t:.2f} s",
    }


def cpu_baseline_c(robot_name, obstacles, d, W, K, budget_s=6.0):
    """The C restatement (oracle/lmik_oracle.c, canonical fp32 build, LU solve in reference order), OpenMP over rows."""
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import oracle as orc

    orc.build()
    cores = os.cpu_count() or 1
    chain = canonicalize(ROBOT_SPECS[robot_name]())
    o = orc.Oracle(chain, f32=True, threads=cores)
    lo_b = np.array([np.float32(T[:3, 3]) + np.float32(c[:3]) for c, T in obstacles], dtype=np.float64).reshape(-1, 3)
    hi_b = np.array([np.float32(T[:3, 3]) + np.float32(c[3:]) for c, T in obstacles], dtype=np.float64).reshape(-1, 3)

    def run(S_cpu):
        q_star, x0 = _cpu_inputs(chain, W, d, S_cpu)
        tgt = np.tile(o.fk(q_star.astype(np.float64)), (S_cpu, 1))
        x0 = x0.astype(np.float64)
        t0 = time.perf_counter()
        x = o.lm_steps(x0, tgt, K, 1e-6, 3.5, 0.35, solver=0)
        o.pose_metrics(x, tgt)
        o.masks(x, lo_b, hi_b, chain.lo, chain.hi)
        return time.perf_counter() - t0

    t_probe = run(64)
    S_cpu = int(max(64, min(32768, 64 * budget_s / max(t_probe, 1e-6))))
    t = run(S_cpu)
    return {
        "value": S_cpu * W * K / t,
        "unit": "LM-IK iterations/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{S_cpu} seeds x {W} waypoints x {K} LM iterations + pose metrics + collision masks; scalar C restatement "
        f"(oracle/lmik_oracle.c, fp32 canonical build), OpenMP {cores} threads, {t:.2f} s",
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--robot", default="panda")
    ap.add_argument("--seeds", type=int, default=1024,
                    help="seeds of the configuration: the TOTAL that is sharded over the GPUs under strong scaling, per GPU under weak")
    ap.add_argument("--waypoints", type=int, default=256)
    ap.add_argument("--lm-steps", type=int, default=10, help="K fused LM iterations per launch")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="auto = strong for N > 1 (BASELINE.json configs[3]: the same seeds sharded over the GPUs); the other mode "
                    "is measured in the same run and reported as a sibling key")  # fmt: skip
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps between barrier + synchronize pairs) is measured this many times "
                    "back to back; value / ms_per_step are the MEDIAN repetition (min and max in config.timed_region)")
    ap.add_argument("--short-region-repeats", type=int, default=16,
                    help="a timed region shorter than 0.4 ms is measured this many times more (same exactly---steps regions; the median "
                    "of all repetitions is reported); 0 = never")
    ap.add_argument("--kernel-reps", type=int, default=400,
                    help="isolated launches behind roofline.kernel_ms (median of HIP-event pairs after a pre-warm)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed launches before the W warm-up steps, to reach sustained clocks (0 disables)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: steps per all-gather of the per-seed summaries (the summaries of G steps travel in one collective); "
                    "0 = 8, 32 for shards of <= 65 536 rows, 64 for <= 32 768: a collective's latency is paid once per G steps")  # fmt: skip
    ap.add_argument("--batch", type=int, default=0,
                    help="steps per launch (cppf_lm_batch_*: B independent problems in one grid); 0 = as many of this rank's steps as "
                    "make one full-width launch of 262 144 rows, at most 16 (1 at N = 1 / C4)")  # fmt: skip
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the independent steps alternate between (0 = 2, or 4 for strong-scaling shards that cannot "
                    "fill the chip with two launches in flight)")  # fmt: skip
    ap.add_argument("--graphs", choices=["auto", "on", "off"], default="auto",
                    help="replay each stream's run of consecutive steps as one captured hipGraph instead of one host call per step "
                         "(auto: on for shards of <= 65536 rows, where a step is shorter than the host's launch call)")
    ap.add_argument("--shape", choices=["auto", "row", "quad"], default="auto", help="kernel shape (cppf_lm_params.shape)")
    ap.add_argument("--solver", choices=["auto", "f32", "f64"], default="auto",
                    help="precision of the damped solve (cppf_lm_params.solver): auto = the reference's dtype with the conditioning gate "
                    "(rows whose fp32 solve is estimated to be off by > 1e-5 in task space redo it in double precision; the default and "
                    "the headline); f32 = no gate; f64 = every row in double precision")  # fmt: skip
    ap.add_argument("--inputs", choices=["problem", "random"], default="problem",
                    help="problem: the named reference problem's target path + per-seed IK branches (SURVEY 8d); "
                    "random: independent random configurations per waypoint (the 8d fall-back, worst case for the broad phase)")  # fmt: skip
    ap.add_argument("--no-collide", action="store_true", help="FK+Jacobian+LM only (BASELINE configs[1] style)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-siblings", action="store_true", help="skip the one_stream / random_inputs / weak_scaling sibling measurements")
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default=None,
                    help="BASELINE.json configs[1..4] geometry (default = C4, the configuration the metric is quoted on)")  # fmt: skip
    args = ap.parse_args(argv)
    if args.config is not None:
        preset = {  # robot, seeds, waypoints, collision fused
            "C2": ("panda", 128, 64, False),  # FK+Jacobian+LM only
            "C3": ("fetch", 512, 256, True),  # + collision fused (fetch__hello has no obstacles: self-collision only)
            "C4": ("panda", 1024, 256, True),
            "C5": ("chain12", 4096, 512, True),
        }[args.config]
        args.robot, args.seeds, args.waypoints = preset[0], preset[1], preset[2]
        args.no_collide = not preset[3]
    return args


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not imported torch,
    let alone touched the GPU), relay rank 0's JSON line, return the first non-zero exit status (0 if every rank succeeded).
    Never replaces a running process: children are started with subprocess and waited for."""
    import tempfile

    n = args.gpus
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))  # fmt: skip
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))  # fmt: skip
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:  # a rank died: the others would wait in a collective for ever
                    other.terminate()
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    out0.seek(0)
    for ln in out0.read().splitlines():  # stdout carries the ONE JSON line; anything else a library printed goes to stderr
        (sys.stdout if ln.startswith('{"metric"') else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return rc


class HostStagedGather:
    """Rehearsal transport (CPPF_BENCH_SHARE_GPU=1: several ranks share ONE GPU, which RCCL refuses): the all-gather goes
    through gloo with the payload staged on the host.  Same call sites, same dependency structure, meaningless timing."""

    def __init__(self, dist):
        self.dist = dist

    def all_gather(self, out, inp):
        host_in = inp.cpu()  # synchronises on the current (auxiliary) stream
        host_out = torch.empty((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]), dtype=out.dtype)
        self.dist.all_gather_into_tensor(host_out, host_in)
        out.copy_(host_out.view(out.shape))


class RcclGather:
    def __init__(self, dist):
        self.dist = dist

    def all_gather(self, out, inp):
        # the concatenated form (world * G rows of [S, 8]); `out` [world, G, S, 8] is the same memory
        work = self.dist.all_gather_into_tensor(out.view((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:])), inp, async_op=True)
        work.wait()  # stream-side: the current (auxiliary) stream waits for the communicator's stream


class CAbiGather:
    """RCCL through the library's own C ABI (cppf_comm_init_rank / cppf_allgather_bytes): one ncclAllGather enqueued on the
    auxiliary stream itself -- no second stream, no c10d bookkeeping (3 us of host time and 6 us on the stream against 28 / 33 us
    for torch.distributed's call on one rank, scripts/gather_latency.py).  Built by `pick_cabi_or_c10d`, which ships the
    communicator's unique id to the other ranks through the torch.distributed group that is up anyway."""

    def __init__(self, comm, world):
        from cppflow_amd import _hip

        self._hip, self.comm, self.world = _hip, comm, world

    def all_gather(self, out, inp):
        nbytes = inp.numel() * inp.element_size()
        assert out.numel() * out.element_size() == nbytes * self.world
        self._hip.check(self._hip.lib().cppf_allgather_bytes(self.comm, inp.data_ptr(), out.data_ptr(), nbytes,
                                                              torch.cuda.current_stream(inp.device).cuda_stream))

    def close(self):
        self._hip.lib().cppf_comm_destroy(self.comm)


def pick_cabi_or_c10d(dist, rank, world, dev_index, device):
    """The C-ABI communicator if EVERY rank can bring it up and its all-gather equals torch.distributed's own on a probe, else
    the c10d call on every rank.  Returns (transport, record).

    Every rank executes the SAME sequence of collectives on the c10d group whatever happens to it locally (a rank
    that failed early used to jump to the fall-back's all-reduce while the others still sat in the broadcast -- mismatched
    collectives, a hang until the c10d timeout): local failures become flags, and after each stage all ranks MIN-reduce their
    flag and leave together.  Stage 0 (no collective): can this rank load RCCL through the library (cppf_comm_available)?  1: rank
    0 draws the unique id and ALWAYS broadcasts (status, id).  2: all ranks agree to go on, then call cppf_comm_init_rank.  3:
    agree again, then the probe (both all-gathers on every rank).  4: agree on the comparison.  A successful probe is the
    validation of this transport on the hardware the run is on."""
    import ctypes

    from cppflow_amd import _hip

    def all_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    rec = {"requested": "cabi", "stages": []}
    why, lib = "", None
    try:
        lib = _hip.lib()
        ok = lib.cppf_comm_available() == 0
        if not ok:
            why = lib.cppf_last_error().decode("utf-8", "replace")
    except Exception as e:  # noqa: BLE001 -- any local failure becomes a flag
        ok, why = False, repr(e)
    box = [None]
    if rank == 0:
        uid = (ctypes.c_char * 128)()
        st = False
        if ok:
            try:
                st = lib.cppf_comm_unique_id(uid) == 0
            except Exception as e:  # noqa: BLE001
                why = repr(e)
        box = [(bool(st), bytes(uid))]
    dist.broadcast_object_list(box, src=0)
    ok = ok and bool(box[0][0])
    go = all_ok(ok)
    rec["stages"].append({"stage": "load RCCL through the C ABI + unique id from rank 0 (torch.distributed broadcast)", "ok": go})
    comm = ctypes.c_void_p()
    if go:
        try:
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0][1])
            ok = lib.cppf_comm_init_rank(uid, rank, world, dev_index, ctypes.byref(comm)) == 0 and lib.cppf_comm_world(comm) == world
            if not ok:
                why = lib.cppf_last_error().decode("utf-8", "replace")
        except Exception as e:  # noqa: BLE001
            ok, why = False, repr(e)
        go = all_ok(ok)
        rec["stages"].append({"stage": "cppf_comm_init_rank on every rank", "ok": go})
    if go:
        cabi = CAbiGather(comm, world)
        probe = torch.full((1, 4, 8), float(rank + 1), dtype=torch.float32, device=device)
        got = torch.zeros((world, 1, 4, 8), dtype=torch.float32, device=device)
        want = torch.zeros_like(got)
        try:
            cabi.all_gather(got, probe)
        except Exception as e:  # noqa: BLE001
            ok, why = False, repr(e)
        dist.all_gather_into_tensor(want.view(world, 4, 8), probe)  # (every rank, whatever the C-ABI call did)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(got, want))
        if not ok and not why:
            why = "probe mismatch"
        go = all_ok(ok)
        rec["stages"].append({"stage": "probe: cppf_allgather_bytes == torch.distributed all_gather_into_tensor", "ok": go})
        if go:
            rec.update(transport="RCCL through the C ABI (cppf_allgather_bytes on the launch stream)", world_seen=int(lib.cppf_comm_world(comm)),
                       unique_id_via="cppf_comm_unique_id on rank 0 -> torch.distributed broadcast_object_list")
            return cabi, rec
    if comm.value:
        try:
            lib.cppf_comm_destroy(comm)
        except Exception:  # noqa: BLE001
            pass
    print(f"bench: rank {rank}: C-ABI RCCL transport not used ({why or 'failed on another rank'}); every rank uses torch.distributed's all-gather",
          file=sys.stderr)
    rec.update(transport="nccl (RCCL) through torch.distributed", world_seen=dist.get_world_size(), unique_id_via="torch.distributed (c10d store)",
               fallback_reason=why or "failed on another rank")
    return RcclGather(dist), rec


def device_census(dist, rank, dev_index):
    """[(rank, device ordinal, PCI bus id, name)] of every rank, gathered through the process group: lets a reader of the JSON
    check that N ranks sat on N different GPUs."""
    p = torch.cuda.get_device_properties(dev_index)
    bus = None
    if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    mine = {"rank": rank, "device": dev_index, "pci_bus_id": bus, "uuid": str(getattr(p, "uuid", "")) or None, "name": p.name}
    if dist is None:
        return [mine]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return everyone


class NoGather:
    """diagnostic (CPPF_BENCH_TRANSPORT=none): everything of the exchange step except the collective itself"""

    def all_gather(self, out, inp):
        out.view((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]))[: inp.shape[0]].copy_(inp)


FULL_WIDTH_ROWS = 262144  # four wavefronts per SIMD of the row shape: the launch width of the N = 1 workload


def launch_plan(rows_main, steps, batch_arg=0, gather_every=0, streams_arg=0, quad=False, max_batch=16):
    """How a rank issues a region of `steps` steps of `rows_main` rows each -> (steps per launch, steps per collective (bucket), the
    requested bucket before clamping, launch streams).  Pure host logic (tests/test_bench_runner_order.py holds it to the measured
    choices for the driver's flags at N = 1 / 2 / 4 / 8)."""
    # steps per launch: as many of this rank's steps as make one full-width launch (1 at N = 1 / C4; 2 / 4 / 8 for the shards of 2 /
    # 4 / 8 GPUs), so that every GPU issues launches of the same width at every N
    batch = batch_arg if batch_arg > 0 else max(1, min(max_batch, FULL_WIDTH_ROWS // max(rows_main, 1)))
    if quad:
        batch = 1
    rows_launch = rows_main * batch
    # steps per collective: the all-gather's latency (tens of microseconds across a node) is paid once per bucket; never more than the
    # timed region holds (at the driver's --steps 20 at least one FULL exchange must lie inside the region), a multiple of the batch
    G_req = gather_every if gather_every > 0 else (64 if rows_main <= 32768 else (32 if rows_main <= 65536 else 8))
    # (a region of K steps holds at least two full buckets when it can: at the driver's --steps 20 and 8 steps per launch that is one
    # exchange behind every launch -- two launches on two streams in flight -- instead of one bucket that serialises two launches)
    G = max(batch, (min(G_req, max(steps // 2, 1)) // batch) * batch)
    # ... and in a region of only a few buckets every launch is followed by its own exchange (one bucket = one launch), so that
    # consecutive launches alternate between the streams like those of an N = 1 run: with 8-step buckets the driver's 20-step region
    # of a 512-seed shard (4 + 4 + 2 launches of two steps) put four launches in a row on one stream and ended on two that ran alone
    # -- 23.9 against 19.4 us per step; 45.1 against 36.8 for 1024 seeds per rank (profiles/r4_short_region_buckets.txt)
    # (a bucket stays two launches where that is still at most 8 steps of a shard of <= 65 536 rows: 10.65 against 11.5 us per step for
    # 256 seeds per rank)
    if gather_every <= 0 and steps < 4 * G:
        G = batch if (rows_main > 65536 or 2 * batch > 8) else 2 * batch
    # launches of <= 2 wavefronts per SIMD: two in flight cannot fill the chip, four can (profiles/r2_hwq_sweep.txt)
    n_streams = streams_arg if streams_arg > 0 else (4 if rows_launch <= 131072 else 2)
    return batch, G, G_req, n_streams


class Runner:
    """One workload (a batch of S seeds x W waypoints on this rank) and the machinery that steps it: a ring of output-buffer
    sets, `n_streams` launch streams, and launches of B consecutive steps each (cppf_lm_batch_*: B independent problems in one grid;
    B = 1 is one step per launch through the same entry point).

    A LAUNCH GROUP is B consecutive ring slots; a launch always starts at a group's first slot and carries 1 .. B of its steps (the
    K mod B steps left over at the end of a region go out as one shorter launch, after which the ring moves on to the next group).
    Without a transport (N = 1) consecutive groups alternate between the streams.  With one, the ring is `n_streams` BUCKETS of G
    steps (G a multiple of B); a bucket's launches all go to ONE stream and its exchange step -- the all-gather of the G [S,8]
    summaries and the seed selection over every rank's seeds -- is enqueued on that same stream right behind them, so producer ->
    collective -> consumer -> reuse of the bucket's buffers are ordered by the stream itself.  No cross-stream event anywhere:
    measured on a 32 768-row shard, making an auxiliary stream wait on events of four launch streams cost 22.5 us per step against
    7.2 us without the waits (the kernels stopped overlapping), while buckets on their own streams keep the full overlap."""

    def __init__(self, robot, x0, target, K, collide, n_streams, G, transport, world, shape, device, solver=0, graphs=False, batch=1):
        from cppflow_amd import _hip
        from cppflow_amd.data_types import DEFAULT_CONSTRAINTS

        self.robot, self.x0, self.target, self.K, self.collide, self.device = robot, x0, target, K, collide, device
        n, W = x0.shape[0], target.shape[0]
        self.n, self.S, self.W, self.world = n, n // W, W, world
        self.transport = transport if collide else None
        self.n_streams = max(1, n_streams)
        self.use_graphs = bool(graphs)
        self.B = B = max(1, min(int(batch), _hip.MAX_BATCH))
        # the batch entry point is the row shape; an explicit --shape quad keeps the plain per-step launches (B = 1)
        self.use_batch_api = shape != _hip.SHAPE_QUAD
        if not self.use_batch_api:
            self.B = B = 1
        # bucket mode: a ring of `n_streams` buckets of G consecutive steps (G a multiple of B), one stream per bucket
        self.buckets = self.transport is not None or self.use_graphs
        self.G = G = (max(B, (max(1, G) // B) * B) if self.buckets else B)
        self.NBUF = NBUF = self.n_streams * G if self.buckets else max(4, self.n_streams) * B
        prm = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE
        self.prm = prm
        self.x_outs = [torch.empty_like(x0) for _ in range(NBUF)]
        self.packeds = [torch.empty(robot.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=device) if collide else None
                        for _ in range(NBUF)]  # fmt: skip
        self.summ_all = torch.empty((NBUF, self.S, 8), dtype=torch.float32, device=device) if collide else None
        self.errs = None if collide else [(torch.empty(n, device=device), torch.empty(n, device=device)) for _ in range(NBUF)]
        self.shape, self.solver = shape, solver

        def item(b):
            it = dict(x=x0, target=target, x_out=self.x_outs[b])
            if collide:
                it.update(packed_out=self.packeds[b], summary_out=self.summ_all[b])
            else:  # FK + Jacobian + LM only (BASELINE configs[1]): the result and its pose errors, no collision stage
                it.update(errors_out=self.errs[b])
            return it

        # launches[g][c - 1]: the launch of the first c steps of group g (c = B: the group; c < B: what is left at the end of a region)
        self.launches = []
        for g in range(NBUF // B):
            if self.use_batch_api:
                self.launches.append([robot.lm_batch_plan([item(g * B + j) for j in range(c)], n_steps=K, solver=solver, **prm)
                                      for c in range(1, B + 1)])
            elif collide:
                self.launches.append([robot.lm_launch_plan(x0, target, n_steps=K, x_out=self.x_outs[g], packed_out=self.packeds[g],
                                                           summary_out=self.summ_all[g], shape=shape, solver=solver, **prm)])
            else:
                self.launches.append([robot.lm_launch_plan(x0, target, n_steps=K, x_out=self.x_outs[g], errors_out=self.errs[g],
                                                           shape=shape, solver=solver, **prm)])
        first = self.launches[0][0].outputs
        self.outputs = first[0] if isinstance(first, list) else first  # ring slot 0's output views
        if self.transport is not None:
            self.gathered = [torch.empty((world, G, self.S, 8), dtype=torch.float32, device=device) for _ in range(self.n_streams)]
            self.selected = [torch.empty((G, 4), dtype=torch.int32, device=device) for _ in range(self.n_streams)]
            self.constraints = DEFAULT_CONSTRAINTS
        else:
            self.gathered = self.selected = None
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n_streams)]
        self.start_bucket = int(os.environ["CPPF_BENCH_START_BUCKET"]) if os.environ.get("CPPF_BENCH_START_BUCKET") else None  # (developer override)
        self.start_bucket_calibration_us_per_step = None
        self.region_prewarm = int(os.environ.get("CPPF_BENCH_REGION_PREWARM", "0"))
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream(device))
        self.step_no = 0  # always a multiple of B: the ring position of the next launch
        self.graphs = None
        if self.use_graphs:
            # One hipGraph per bucket = its G / B launches in stream order, captured on the bucket's own stream (every launch once
            # eagerly first: nothing lazy may happen inside a capture).  A replay costs the host one call per G steps.
            for g in range(NBUF // B):
                self.launches[g][B - 1].launch_on(self.streams[(g * B) // G])
            torch.cuda.synchronize()
            try:
                graphs = []
                for bucket in range(self.n_streams):
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=self.streams[bucket], capture_error_mode="thread_local"):
                        for g in range(bucket * G // B, (bucket + 1) * G // B):
                            self.launches[g][B - 1].launch_on(self.streams[bucket])
                    graphs.append(gr)
                torch.cuda.synchronize()
                self.graphs = graphs
            except RuntimeError as e:  # capture refused on this box: the eager path does the same work, one host call per launch
                print(f"bench: hipGraph capture failed ({e}); continuing with eager launches", file=sys.stderr)
                torch.cuda.synchronize()
                self.graphs = None

    def stream_of(self, b):
        return self.streams[b // self.G] if self.buckets else self.streams[(b // self.B) % self.n_streams]

    def launch(self):
        """one full launch (B steps) on torch's current stream (group 0)"""
        self.launches[0][self.B - 1].launch()

    def run_steps(self, n):
        """`n` steps = floor(n / B) launches of B steps and one of n mod B; whole buckets go out as one graph replay each when
        graphs are on.  A bucket's exchange step follows its last launch ON ITS STREAM; the host issues it one launch late -- after
        the next bucket's first launch has gone to ITS stream -- so that the second stream's kernels are not held back by the
        ~10 us of host time the collective and the selection launch take (a 20-step region of a 32 768-row shard is ~100 us)."""
        B, G = self.B, self.G
        pending = None  # a complete bucket whose exchange step has not been issued yet

        def flush():
            nonlocal pending
            if pending is not None:
                self.exchange(pending)
                pending = None

        while n > 0:
            b = self.step_no % self.NBUF
            if self.graphs is not None and b % G == 0 and n >= G:
                bucket = b // G
                if pending == bucket:
                    flush()
                with torch.cuda.stream(self.streams[bucket]):
                    self.graphs[bucket].replay()
                self.step_no += G
                n -= G
                flush()
                if self.transport is not None:
                    pending = bucket
                continue
            c = min(B, n)
            if pending is not None and self.buckets and pending == b // G:
                flush()  # (a one-bucket ring: the launch below would overwrite the summaries still to be gathered)
            self.launches[b // B][c - 1].launch_on(self.stream_of(b))
            self.step_no += B  # (a shorter launch leaves the rest of its group unused: the ring moves on to the next group)
            n -= c
            flush()
            if self.buckets and self.transport is not None and (b + B) % G == 0:
                pending = b // G  # the bucket is complete: gather its G summaries from every rank and consume them
        flush()

    def exchange(self, bucket):
        G = self.G
        with torch.cuda.stream(self.streams[bucket]):
            self.transport.all_gather(self.gathered[bucket], self.summ_all[bucket * G : (bucket + 1) * G])
            # the consumer (cppflow/optimization_utils.py:856-909 over ALL ranks' seeds, one row of `selected` per step)
            self.robot.select_valid_seed(self.gathered[bucket], self.constraints, out=self.selected[bucket])

    def drain(self):
        if self.buckets and self.step_no % self.G != 0:
            if self.transport is not None:  # a partly filled bucket: its summaries are exchanged too before the clock stops
                self.exchange((self.step_no % self.NBUF) // self.G)
            self.step_no += self.G - self.step_no % self.G  # the next step starts a fresh bucket

    def timed(self, steps, warmup, prewarm_ms, barrier, repeats=1, closing_barrier=None):
        """`prewarm_ms` of untimed launches (sustained clocks, full pipeline), W untimed warm-up steps, then `repeats` times:
        opening barrier + synchronize, clock, exactly `steps` steps (+ the exchange of a partly filled bucket), THIS RANK's
        synchronize, clock.  The group's closing barrier (`closing_barrier`, N > 1) comes after the clock: the caller takes the
        maximum over ranks of the elapsed times, which is what a barrier-closed region measures minus the barrier's own latency.
        Returns (elapsed seconds of every repetition, seconds the closing barrier took each time)."""
        # the pre-warm is time-based, so it must not contain collectives (ranks would issue different numbers of them):
        # bare full launches round-robin over the groups and streams
        t_pre = time.perf_counter()
        ngroups = self.NBUF // self.B
        while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
            for i in range(max(1, 48 // self.B)):
                g = i % ngroups
                self.launches[g][self.B - 1].launch_on(self.stream_of(g * self.B))
            torch.cuda.synchronize()
        self.run_steps(warmup)
        self.drain()
        # Which stream a region starts on -- and which streams it uses -- matters when the region is a handful of launches: with two
        # buckets of 8 steps on two streams, a 20-step region of a 32 768-row shard (launches of 8 / 8 / 4 steps) takes 6.9 us per step
        # when it starts on the first stream and 5.8 when it starts on the second, every time within one process
        # (profiles/r4_start_bucket.txt; the ring position used to alternate between repetitions, and the timings with it).  So, once
        # per run and untimed: three regions per candidate, the best one is how every timed region runs (the ring may restart
        # anywhere: everything issued before has completed).  Every rank runs the same number of calibration regions -- they contain
        # collectives -- and decides for itself.
        def region():
            """one repetition: opening barrier, clock, exactly `steps` steps + drain, this rank's synchronize, clock, closing barrier"""
            barrier()
            if self.region_prewarm > 0:
                # the group's barrier leaves the GPU idle for 50 - 150 us and the clocks drop with it: a few bare launches (no
                # collectives; the same fixed number on every rank) and a synchronize bring them back before the clock starts
                for i in range(self.region_prewarm):
                    g = i % ngroups
                    self.launches[g][self.B - 1].launch_on(self.stream_of(g * self.B))
                torch.cuda.synchronize()
            if self.start_bucket is not None:
                self.step_no = self.start_bucket * self.G
            t0 = time.perf_counter()
            self.run_steps(steps)
            self.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tc = None
            if closing_barrier is not None:
                closing_barrier()
                tc = time.perf_counter() - t1
            return t1 - t0, tc

        if self.start_bucket is None and self.buckets and self.transport is not None and self.n_streams >= 2 and steps <= 64:
            # candidates: (streams of the buckets, start bucket).  Two buckets and eager launches: every ORDERED pair out of a pool of
            # six streams, starting on the first -- which hardware queue a stream lands on, and what else shares it, is decided when
            # the runtime creates it, and the same 8 / 8 / 4-step region took 5.8 ... 7.0 us per step over three processes on ONE box
            # with the two streams this Runner happened to get (profiles/r4_start_bucket.txt).  Otherwise: the start bucket only.
            if self.n_streams == 2 and self.graphs is None:
                pool = list(self.streams) + [torch.cuda.Stream(device=self.streams[0].device) for _ in range(4)]
                cands = [((i, j), 0) for i in range(len(pool)) for j in range(len(pool)) if i != j]
            else:
                pool = list(self.streams)
                cands = [(tuple(range(self.n_streams)), sb) for sb in range(self.n_streams)]
            med = []
            for idx, sb in cands:
                self.streams = [pool[i] for i in idx]
                self.start_bucket = sb
                med.append(float(np.median([region()[0] for _ in range(3)])))  # (the very repetition that is timed below)
            best = int(np.argmin(med))
            self.streams = [pool[i] for i in cands[best][0]]
            self.start_bucket = cands[best][1]
            self.start_bucket_calibration_us_per_step = {"candidates": len(cands), "chosen_streams": list(cands[best][0]), "chosen_start_bucket": cands[best][1],
                                                         "best": 1e6 * med[best] / steps, "median": 1e6 * float(np.median(med)) / steps,
                                                         "worst": 1e6 * max(med) / steps, "first": 1e6 * med[0] / steps}
        out, closing = [], []
        for _ in range(max(1, repeats)):
            dt, tc = region()
            out.append(dt)
            if tc is not None:
                closing.append(tc)
        return out, closing

    def host_enqueue_us(self):
        """diagnostic: host cost of issuing one step (64 steps into an empty queue, no waiting on the GPU)"""
        torch.cuda.synchronize()
        nh = max(64, 4 * self.G) if self.graphs is None else 4 * self.G
        nh = (nh // self.B) * self.B
        th = time.perf_counter()
        self.run_steps(nh)
        t = (time.perf_counter() - th) / nh
        self.drain()
        torch.cuda.synchronize()
        return 1e6 * t

    def kernel_ms(self, reps, prewarm=300):
        """Isolated launch duration: HIP events bracketing single launches on the launch stream (torch's current stream IS
        the stream the kernel is launched on), one launch in flight at a time, no collective inside the bracket.  `prewarm`
        untimed launches first (an idle gap drops the clocks: the first launches after one run 30-40 % long), then the MEDIAN of
        `reps` pairs -- one host hiccup of a millisecond moves a mean of 50 by 20 us and a median not at all.  Returns a dict.
        (A launch is B steps: `launch_steps` says how many.)"""
        for _ in range(prewarm):
            self.launch()
        torch.cuda.synchronize()
        for _ in range(prewarm // 4):
            self.launch()
        kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in kev:
            a.record()
            self.launch()
            b.record()
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) for a, b in kev])
        return {"median": float(np.median(t)), "mean": float(t.mean()), "min": float(t.min()), "p10": float(np.quantile(t, 0.1)),
                "p90": float(np.quantile(t, 0.9)), "max": float(t.max()), "n": int(reps), "launch_steps": self.B}


def dryrun(args, world, rank):
    """CPPF_BENCH_DRYRUN=1 (tests/test_bench_launcher.py, no GPU): the rank choreography of an N > 1 run without a single
    kernel -- gloo process group, seed sharding, one all-gather of [S_local, 8] stand-in summaries (the seed indices),
    the host-side seed selection over all ranks' seeds, max-over-ranks, ONE JSON line from rank 0.  It measures nothing and
    says so (`value` null, `data` "dryrun")."""
    import torch.distributed as dist

    from cppflow_amd.data_types import DEFAULT_CONSTRAINTS
    from cppflow_amd.distributed import allgather_seed_summaries, drop_padding, padded_shard_size, seed_shard, shard_counts
    from cppflow_amd.evaluation_utils import seed_metrics_are_below_threshold

    if os.environ.get("CPPF_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(7)
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)  # gloo / RCCL print banners on stdout; it carries exactly ONE JSON line
    os.dup2(2, 1)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    S, W = args.seeds, args.waypoints
    b, e = seed_shard(S, rank, world)
    S_pad = padded_shard_size(S, W, world)
    mine = torch.zeros((S_pad, 8), dtype=torch.float32)
    mine[: e - b, 7] = torch.arange(b, e, dtype=torch.float32)  # stand-in for the summed cost: the global seed index
    mine[e - b :] = float("inf")  # filler seeds can never be selected
    allseeds = drop_padding(allgather_seed_summaries(mine), S_pad, shard_counts(S, world))
    valid = [i for i in range(allseeds.shape[0]) if seed_metrics_are_below_threshold(DEFAULT_CONSTRAINTS, allseeds[i, :4])[0]]
    # the same self-verification as the real run: every rank's selection gathered and compared, and against the single-process
    # answer (here: seed 0 is the first valid one, all S are valid, and the stand-in cost makes seed 0 the cheapest)
    mine_sel = [valid[0] if valid else -1, len(valid), int(torch.argmin(allseeds[:, 7]).item()), 0]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine_sel)
    assert all(e == everyone[0] for e in everyone), everyone
    census = [None] * world
    dist.all_gather_object(census, {"rank": rank, "device": None, "pci_bus_id": None, "name": "cpu (dry run)"})
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps({"metric": "LM-IK iterations/sec (seeds x waypoints)", "value": None, "unit": "LM-IK iterations/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "strong",
                          "data": "dryrun (no GPU: launcher and collective choreography only)",
                          "config": {"world_size": dist.get_world_size(), "seeds_total": S, "seeds_per_gpu_padded": S_pad,
                                     "gathered_seed_ids": [int(v) for v in allseeds[:, 7]], "n_valid": len(valid),
                                     "max_rank": int(t.item())},
                          "rccl": {"requested": "dryrun", "transport": "gloo (dry run)", "world_seen": dist.get_world_size(),
                                   "unique_id_via": "torch.distributed (c10d store)", "ranks": census},
                          "selection_check": {"selected_by_rank": everyone, "identical_on_every_rank": True,
                                              "single_process_selection": [0, S, 0, 0],
                                              "equals_single_process": everyone[0] == [0, S, 0, 0]}}), flush=True)  # fmt: skip
        os.dup2(2, 1)
    dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:  # checked before anything touches the GPU
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` (it starts its own "
                 f"ranks) or launch N ranks with torch.distributed.run and pass the same --gpus N")  # fmt: skip

    # Independent launches on different HIP streams only overlap when the streams map to different hardware queues; the
    # runtime's default of 4 queues per process pairs them up (measured, scripts/hwq_sweep.sh: a 32 768-row shard steps in
    # 13.1 us on 4 streams with the default and in 7.1 us with 16 queues; the full-size launch is unaffected).  Must be set before
    # the HIP runtime initialises, i.e. before torch is imported.
    if os.environ.get("CPPF_BENCH_KEEP_HWQ", "0") != "1":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    global torch
    import torch

    if os.environ.get("CPPF_BENCH_DRYRUN", "0") == "1":
        return dryrun(args, world, rank)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # CPPF_BENCH_SHARE_GPU=1: rehearsal of the N > 1 code path on a one-GPU box (every rank on device 0, host-staged gloo)
    share_gpu = os.environ.get("CPPF_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants GPU {dev_index} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    saved_stdout_fd = None
    transport = None
    rccl_rec = None
    # CPPF_BENCH_FORCE_DIST=1 initialises the RCCL process group even for one rank (rehearses the N > 1 code path)
    force_dist = os.environ.get("CPPF_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")  # the tiny collective should not queue behind a full-chip kernel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when the communicator comes up; stdout carries exactly ONE JSON line, so
        # fd 1 points at stderr until the result is printed
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            transport = HostStagedGather(dist)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
            which = os.environ.get("CPPF_BENCH_TRANSPORT", "cabi")
            if which == "cabi":
                transport, rccl_rec = pick_cabi_or_c10d(dist, rank, world, dev_index, device)
            else:
                transport = {"c10d": lambda: RcclGather(dist), "none": lambda: NoGather()}[which]()
                rccl_rec = {"requested": which, "transport": {"c10d": "nccl (RCCL) through torch.distributed", "none": "none (diagnostic)"}[which],
                            "world_seen": dist.get_world_size(), "unique_id_via": "torch.distributed (c10d store)"}
        assert dist.get_world_size() == world

    from cppflow_amd import _hip
    from cppflow_amd.distributed import allgather_seed_outputs, seed_shard
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
    from cppflow_amd.robots import get_robot
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    robot = get_robot(args.robot)
    d, W, K = robot.ndof, args.waypoints, args.lm_steps
    collide = not args.no_collide
    obstacles = obstacle_arrays(PANDA_2CUBES_OBSTACLES) if (collide and args.config != "C3") else []
    robot.set_obstacles([c for c, _ in obstacles], [T for _, T in obstacles])
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    if os.environ.get("CPPF_BENCH_GATE_REL_PPM"):  # developer sweep of the lean iterations' relative gate (cppflow_hip_debug.h)
        robot.debug_set("gate_rel_ppm", int(os.environ["CPPF_BENCH_GATE_REL_PPM"]))
    if os.environ.get("CPPF_BENCH_SPREAD_KB"):  # developer sweep of the residency claim of small launches (cppflow_hip_debug.h)
        robot.debug_set("spread_kb", int(os.environ["CPPF_BENCH_SPREAD_KB"]))
    shape = {"auto": _hip.SHAPE_AUTO, "row": _hip.SHAPE_ROW, "quad": _hip.SHAPE_QUAD}[args.shape]
    solver = {"auto": _hip.SOLVER_AUTO, "f32": _hip.SOLVER_F32, "f64": _hip.SOLVER_F64}[args.solver]

    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    S_cfg = args.seeds
    if scaling == "strong":
        assert S_cfg % world == 0 and (S_cfg // world) >= 1, f"--seeds {S_cfg} must be a multiple of --gpus {world} under strong scaling"
    S_main = S_cfg // world if scaling == "strong" else S_cfg

    full_inputs = {}

    def inputs_for(S_local, mode, kind):
        """(x0 [S_local*W, d], target, description).  Strong scaling: every rank builds the SAME S_cfg seeds (seed 0) and keeps
        its `seed_shard`; weak scaling: rank r builds its own S_cfg seeds (seed r)."""
        gen_seed = 0 if mode == "strong" else rank
        if kind == "problem":
            x_all, tgt, desc = make_inputs_problem(robot, S_cfg, W, device, seed=gen_seed)
        else:
            x_all, tgt = make_inputs(robot, S_cfg, W, device, seed=gen_seed)
            desc = "per waypoint q* ~ U(limits), target = FK(q*), seeds = clamp(q* + 0.1 randn) (SURVEY 8d fall-back inputs)"
        if mode == "strong" or world == 1:
            full_inputs[kind] = (x_all, tgt)  # all S_cfg seeds: the single-process reference of the selection check
        if mode == "strong" and world > 1:
            b, e = seed_shard(S_cfg, rank, world)
            x_all = x_all[b * W : e * W].contiguous()
        return x_all, tgt, desc

    def barrier():
        # (drain this rank's launch streams BEFORE the group's barrier: its all-reduce belongs to torch.distributed's communicator,
        # the exchange steps still in flight to the C-ABI one, and kernels of two communicators must not wait for each other in a
        # different order on different ranks)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def closing_barrier():
        # the closing half of the bracket: AFTER the rank's clock has stopped (Runner.timed), so its latency -- tens of microseconds
        # across eight ranks, as long as the whole region at --steps 20 -- is recorded, not timed
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=device)
        if dist is not None:
            if share_gpu:
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    rows_main = S_main * W
    batch, G, G_req, n_streams = launch_plan(rows_main, args.steps, args.batch, args.gather_every, args.streams, shape == _hip.SHAPE_QUAD, _hip.MAX_BATCH)
    rows_launch = rows_main * batch

    x0, target, inputs_desc = inputs_for(S_main, scaling, args.inputs)
    use_graphs = args.graphs == "on" or (args.graphs == "auto" and rows_launch <= 65536)
    run = Runner(robot, x0, target, K, collide, n_streams, G, transport, world, shape, device, solver, graphs=use_graphs, batch=batch)
    n = run.n
    # the timed region, `--repeats` times back to back; every repetition is the maximum over ranks, the headline the median one
    reps_local, closing_local = run.timed(args.steps, args.warmup, args.prewarm_ms, barrier, args.repeats, closing_barrier if dist is not None else None)
    reps_s = [max_over_ranks(t) for t in reps_local]
    closing_us = [1e6 * max_over_ranks(t) for t in closing_local]
    # A region of a few hundred microseconds (the driver's --steps 20 is ~0.12 ms for a 32 768-row shard; 0.74 ms at N = 1, which
    # repeats to within 0.5 % and is left alone) is at the mercy of single host hiccups -- five repetitions of the shard region came out as 6.05, 6.07, 7.53, 7.75 and 9.89 us per step on one
    # box -- so such a run repeats the region `--short-region-repeats` more times (every rank takes the same decision: it is made on
    # the maximum over ranks) and reports the median of all of them; every repetition is still exactly --steps steps.
    if args.short_region_repeats > 0 and float(np.median(reps_s)) < 0.4e-3:
        # (behind its own pre-warm and warm-up steps: the decision above cost the GPU an idle gap, and the first launches after one
        # run at lower clocks -- without it the extra repetitions of an N = 1 region came out 15 - 30 % long)
        more_local, more_closing = run.timed(args.steps, args.warmup, args.prewarm_ms, barrier, args.short_region_repeats,
                                             closing_barrier if dist is not None else None)
        reps_s += [max_over_ranks(t) for t in more_local]
        closing_us += [1e6 * max_over_ranks(t) for t in more_closing]
    elapsed = float(np.median(reps_s))
    host_us = run.host_enqueue_us()
    kstats = run.kernel_ms(max(200, args.kernel_reps))
    kernel_ms = kstats["median"]

    # sanity on the result of the last step (not timed): most rows converged
    outputs = run.outputs
    conv_frac = float((outputs["pos_err_m"] < 1e-4).float().mean().item())
    selected = None
    selection_check = None
    if run.selected is not None:
        torch.cuda.synchronize()
        selected = [int(v) for v in run.selected[0][0].cpu()]
        # Self-verification of the exchange step (every step is the same computation on the same inputs, so ONE answer is right):
        # (1) every rank must hold the same selection -- it is computed on each rank from the gathered summaries of all ranks;
        # (2) under strong scaling it must equal what ONE process selects over the same S_cfg seeds (every rank built all of
        # them; this rank runs one unsharded launch and the same selection kernel over its own [S_cfg, 8] summary).
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, selected)
        same = all(e == everyone[0] for e in everyone)
        single = None
        if scaling == "strong" or world == 1:
            x_full, tgt_full = full_inputs[args.inputs]
            summ = torch.empty((S_cfg, 8), dtype=torch.float32, device=device)
            pk = torch.empty(robot.PACKED_BYTES_PER_ROW * S_cfg * W, dtype=torch.uint8, device=device)
            robot.lm_pose_steps(x_full, tgt_full, n_steps=K, packed_out=pk, summary_out=summ, shape=shape, solver=solver, **run.prm)
            single = [int(v) for v in robot.select_valid_seed(summ.view(1, 1, S_cfg, 8), run.constraints).reshape(-1).cpu()]
            del pk
        selection_check = {"selected_by_rank": everyone, "identical_on_every_rank": same, "single_process_selection": single,
                           "equals_single_process": (single == selected) if single is not None else None,
                           "fields": "[first valid seed or -1, number of valid seeds, seed of smallest summed cost, 0] over all ranks' seeds"}
        assert same, f"ranks disagree on the selected seed: {everyone}"
        assert single is None or single == selected, f"sharded selection {selected} != single-process selection {single}"

    # Once per planning call (not per step): every rank gets ALL ranks' per-row costs / masks and candidate paths with one
    # all-gather each and runs dp_search over them (cppflow/search.py:146-173 consumes every candidate's cost row).  Untimed
    # here, reported beside the headline: it precedes the LM iterations in the reference pipeline (planners.py:274 -> 402).
    plan_search = None
    if collide:
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k_all = run.S * (world if dist is not None else 1)
        for rep in range(2):
            ev0.record()
            if dist is not None and world > 1 and not share_gpu:
                g = allgather_seed_outputs(run.packeds[0], run.S, W)
                q_all = torch.empty((world,) + tuple(run.x_outs[0].shape), dtype=torch.float32, device=device)
                dist.all_gather_into_tensor(q_all, run.x_outs[0])
                cost_all, q_all = g.ext_cost.contiguous(), q_all.view(k_all, W, d)
            else:
                k_all = run.S
                cost_all = run.outputs["ext_cost"].view(run.S, W)
                q_all = run.x_outs[0].view(run.S, W, d)
            robot.dp_search(q_all, cost_all)
            ev1.record()
            torch.cuda.synchronize()
        plan_search = {"candidates": k_all, "waypoints": W, "ms": ev0.elapsed_time(ev1),
                       "allgather_bytes_per_rank": int(run.packeds[0].numel() + run.x_outs[0].numel() * 4) if (dist is not None and world > 1) else 0,
                       "what": "all-gather of the packed per-row outputs + candidate paths, then cppf_dp_search over every rank's "
                       "candidates (once per planning call; untimed, outside `value`)"}  # fmt: skip

    def measure_sibling(S_local, mode, kind, streams, steps):
        """a second workload / pipeline depth measured like the headline (same barriers, same max over ranks)"""
        xs, tg, _ = inputs_for(S_local, mode, kind)
        b2 = max(1, min(_hip.MAX_BATCH, FULL_WIDTH_ROWS // max(xs.shape[0], 1))) if args.batch <= 0 else args.batch
        b2 = 1 if shape == _hip.SHAPE_QUAD else b2
        G2 = max(b2, (min(G_req, max(steps // 2, 1)) // b2) * b2)
        r2 = Runner(robot, xs, tg, K, collide, streams, G2, transport, world, shape, device, solver, batch=b2)
        el = float(np.median([max_over_ranks(t) for t in r2.timed(steps, min(args.warmup, 100), min(args.prewarm_ms, 30.0), barrier, 3,
                                                                    closing_barrier if dist is not None else None)[0]]))
        km = r2.kernel_ms(200, prewarm=200)["median"]
        rows = float(r2.n) * world
        del r2
        torch.cuda.empty_cache()
        return {"value": rows * K * steps / el, "ms_per_step": 1e3 * el / steps, "kernel_ms": km, "streams": streams,
                "rows_per_gpu": int(rows // world), "steps_per_launch": b2}  # fmt: skip

    def measure_neighbour_stages(robot, run, W, d, device):
        """The two stages on either side of the hot path, on this workload's own result, untimed and
        outside `value`: the coupled differencing step (cppf_lm_full_step, ALT_LOSS_V2_1_DIFF) for all seeds and for one, and
        dp_search over the reference's k = 175 candidates.  HIP events, median of 5 rounds of 5 calls."""
        from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF

        def timed_us(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 5 * 1e3)
            return float(np.median(ts))

        x = run.x_outs[0]
        out = {"coupled_step_us": timed_us(lambda: robot.lm_full_step(x, target, ALT_LOSS_V2_1_DIFF)), "coupled_step_trajectories": run.S,
               "coupled_step_one_trajectory_us": timed_us(lambda: robot.lm_full_step(x[:W], target, ALT_LOSS_V2_1_DIFF))}
        kk = min(175, run.S)
        q3 = x.view(run.S, W, d)[:kk].contiguous()
        cost = run.outputs["ext_cost"].view(run.S, W)[:kk].contiguous()
        out.update(dp_search_us=timed_us(lambda: robot.dp_search(q3, cost)), dp_search_candidates=kk, waypoints=W)
        return out

    siblings = {}
    if not args.no_siblings:
        sib_steps = min(args.steps, 1000)
        if world > 1:
            other = "weak" if scaling == "strong" else "strong"
            if other == "weak" or S_cfg % world == 0:
                S_o = S_cfg if other == "weak" else S_cfg // world
                r = measure_sibling(S_o, other, args.inputs, 2 if other == "weak" else 4, sib_steps)
                r["scaling"] = other
                r["seeds_per_gpu"] = S_o
                siblings[other + "_scaling"] = r
        else:
            siblings["one_stream"] = measure_sibling(S_main, scaling, args.inputs, 1, sib_steps)
            if args.inputs == "problem":
                siblings["random_inputs"] = measure_sibling(S_main, scaling, "random", n_streams, sib_steps)
            if collide and d <= 12 and W >= 2:
                siblings["neighbour_stages"] = measure_neighbour_stages(robot, run, W, d, device)

    census = device_census(dist, rank, dev_index) if not share_gpu else None
    if rank == 0:
        iters = float(n) * K * args.steps * world
        alg_flops = n * (K * algorithmic_flops_per_row_iter(d)
                         + (algorithmic_flops_collision(robot.n_capsules, robot.n_collision_pairs, len(obstacles)) if collide else 0.0))  # fmt: skip
        bytes_launch = n * algorithmic_bytes_per_row(d, collide)
        t_k = kernel_ms * 1e-3
        alg_tflops = alg_flops / t_k / 1e12
        ach_gbps = bytes_launch / t_k / 1e9
        build_id = _hip.lib().cppf_build_id().decode()
        # the committed counters describe ONE launch of `run.B` steps of this workload (scripts/pmc_probe.py records the launch shapes
        # bench.py issues); B > 1 launches of a shard are keyed with their batch
        rec, rec_why = (issue_record_from_profiles(args.robot, run.S, W, K, collide, args.inputs + ("" if run.B == 1 else f"_b{run.B}"), build_id)
                        if args.solver == "auto" else (None, "no counter record for this solver mode"))
        kname = "lm_fused_kernel" if shape != _hip.SHAPE_QUAD else "lm_quad_kernel"
        kprof, kprof_why = kernel_profile_from_profiles(f"{kname}<cppf::StaRobot<cppf::gen::{''.join(p.capitalize() for p in args.robot.split('_'))}>, {1 if collide else 0}", build_id)
        if kprof and run.B != 1:
            kprof, kprof_why = None, "the committed kernel-trace summary is of the N = 1 command (one step per launch)"
        step_s = elapsed / args.steps
        alg_launch = alg_flops * run.B  # one launch = B steps
        roof = {
            # the binding resource is the fp32 VALU issue rate (157.3 TFLOP/s of FMAs = one wave-instruction per SIMD per 2
            # cycles); the contract's vocabulary has no word for it, so `bound` says what it is and `mfma_used` that no
            # matrix instruction is issued on this path
            "bound": "valu",
            "mfma_used": False,
            "peak": F32_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "kernel": kname,
            "kernel_ms": kernel_ms,
            "kernel_launch_steps": run.B,
            "kernel_ms_how": f"median of {kstats['n']} isolated launches of {run.B} step(s) (HIP events on the launch stream, one in flight) after a pre-warm",
            "kernel_ms_stats": kstats,
            # the same kernel's average duration in the committed `rocprofv3 --kernel-trace --stats` summary of this command
            # (profiles/r4_fused_kernel_stats.csv), when that summary was taken with this build of the library
            "kernel_ms_profile": kprof["ms"] if kprof else None,
            "kernel_profile": kprof if kprof else {"unavailable": kprof_why},
            "drift": (kernel_ms / kprof["ms"]) if kprof else None,
            "library_build_id": build_id,
            # the SURVEY 8(d) flop model, for comparison ONLY (never `frac`): it prices the reference's formulation -- primal J^T J +
            # d^3/3 Cholesky, every collision test -- not what this kernel executes (dual 6x6 solve, broad-phase culls)
            "algorithmic": {
                "tflops": alg_launch / t_k / 1e12,
                "note": "SURVEY 8d flop model / isolated kernel time: what the reference's formulation would need, NOT a utilisation figure",
            },
            "traffic": traffic_from_profiles(args.robot, run.S, W, K, collide, build_id) if run.B == 1 else None,
            "hbm": {"achieved": ach_gbps * run.B, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gbps * run.B / HBM_PEAK_GBPS},
        }
        if rec is not None:
            valu = float(rec["valu_insts_per_launch"])
            flops_launch = valu * 64.0 * float(rec["flops_per_valu_lane_op"])
            exe_tflops = flops_launch / t_k / 1e12
            issue_s = valu * 2.0 / N_SIMD / (CLOCK_GHZ * 1e9)
            roof.update({
                "achieved": exe_tflops,
                "frac": exe_tflops / F32_PEAK_TFLOPS,
                "basis": f"executed flops: SQ_INSTS_VALU_* of the matching launch (profiles/{PROFILE_ROUND}_issue.json, same library build) x 64 lanes x "
                "flops per VALU lane-op (FMA = 2, mul / add = 1, moves / selects / compares / transcendentals = 0), / live isolated kernel time",
                "valu_issue_frac": issue_s / t_k,
                "at_step_rate": {"ms_per_step": 1e3 * step_s, "launches_in_flight": run.n_streams, "executed_tflops": flops_launch / run.B / step_s / 1e12,
                                 "frac": flops_launch / run.B / step_s / 1e12 / F32_PEAK_TFLOPS, "valu_issue_frac": issue_s / run.B / step_s},
            })  # fmt: skip
        else:
            roof.update({"achieved": None, "frac": None, "basis": f"unavailable: {rec_why} (the executed-flop basis needs the committed counters of this "
                         "launch shape and library build; the algorithmic model is kept apart under `algorithmic`)",
                         "at_step_rate": {"ms_per_step": 1e3 * step_s, "launches_in_flight": run.n_streams}})  # fmt: skip
        if kprof and abs(kernel_ms / kprof["ms"] - 1.0) > 0.15:
            print(f"bench: WARNING: live kernel time {1e3 * kernel_ms:.1f} us differs from the committed rocprofv3 average "
                  f"{1e3 * kprof['ms']:.1f} us by more than 15 % (drift {kernel_ms / kprof['ms']:.2f})", file=sys.stderr)
        line = {
            "metric": "LM-IK iterations/sec (seeds x waypoints)",
            "value": iters / elapsed,
            "unit": "LM-IK iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.robot}{'__2cubes geometry' if obstacles else ''}, "
                + (f"{S_cfg} seeds sharded over {world} GPU(s) = {run.S} seeds/GPU" if scaling == "strong" else f"{run.S} seeds/GPU")
                + f" x {W} waypoints x {d}-DoF, K={K} fused LM iterations per launch"
                + (" + self/env collision masks + jlim mask + search cost" if collide else " (FK+Jacobian+LM only)"),
                "inputs": inputs_desc,
                "streams": run.n_streams,
                "timed_region": {"repeats": len(reps_s), "reported": "median", "ms_per_step_min": 1e3 * min(reps_s) / args.steps,
                                 "ms_per_step_max": 1e3 * max(reps_s) / args.steps,
                                 "ms_per_step_all": [1e3 * t / args.steps for t in reps_s],
                                 "clock": "opening barrier + synchronize | K steps (+ the exchange of a partly filled bucket) | this rank's "
                                          "synchronize; maximum over ranks.  The group's closing barrier follows the clock.",
                                 "closing_barrier_us": closing_us if closing_us else None,
                                 "region_start_bucket": run.start_bucket,
                                 "region_start_bucket_calibration_us_per_step": run.start_bucket_calibration_us_per_step},
                "steps_per_launch": run.B,
                "launches_per_region": f"{args.steps // run.B} x {run.B} steps" + (f" + 1 x {args.steps % run.B} steps" if args.steps % run.B else ""),
                "rows_per_launch": run.n * run.B,
                "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                "hip_graphs": (f"on: each stream's {run.G} consecutive steps replayed as one captured graph" if run.graphs is not None else "off (one host call per step)"),
                "kernel_shape": args.shape,
                "solver": {"auto": "auto: fp32 (the reference's dtype) + conditioning-gated double-precision redo of the rows whose estimated "
                                   "task-space error exceeds 1e-5 (cppf_lm_params.solver = CPPF_SOLVER_AUTO, the default)",
                           "f32": "f32, no gate", "f64": "f64: every row in double precision"}[args.solver],
                "early_out": "off (every row runs all K iterations: the metric counts K iterations per row)",
                "prewarm_ms": args.prewarm_ms,
                "host_enqueue_us_per_step": host_us,
                "robot": args.robot,
                "seeds_total": S_cfg * (world if scaling == "weak" else 1),
                "seeds_per_gpu": run.S,
                "waypoints": W,
                "ndof": d,
                "lm_iterations_per_step": K,
                "collision_fused": collide,
                "obstacles": len(obstacles),
                "world_size": dist.get_world_size() if dist is not None else 1,
                "collective_backend": (None if dist is None else ("gloo, host-staged (one-GPU rehearsal: NOT a multi-GPU result)" if share_gpu else
                                                                  {"RcclGather": "nccl (RCCL) through torch.distributed", "CAbiGather": "RCCL through the C ABI "
                                                                   "(cppf_allgather_bytes on the auxiliary stream)", "NoGather": "none (diagnostic)", "NoneType": "none (diagnostic)"}[type(transport).__name__])),
                "per_step": ("one fused launch" if run.B == 1 else f"1/{run.B} of a fused launch of {run.B} independent steps (cppf_lm_batch_launch)")
                + " incl. the per-seed summary reduction"
                + (f" + async all-gather of the [S,8] summaries, {run.G} steps per collective, + x_is_valid seed selection over all "
                   f"{run.S * world} seeds of every step on each rank" if run.gathered is not None else ""),
                "allgather_bytes_per_rank_per_step": int(run.summ_all[0].numel() * 4) if run.gathered is not None else 0,
                "steps_per_allgather": run.G if run.gathered is not None else 0,
                "steps_per_allgather_requested": G_req if run.gathered is not None else 0,
                "selected_seed_last_bucket": selected,
                "converged_frac_pos_err_lt_1e-4": conv_frac,
            },
            "roofline": roof,
        }
        if plan_search is not None:
            line["plan_search"] = plan_search
        if rccl_rec is not None:
            line["rccl"] = dict(rccl_rec, ranks=census)
        elif census is not None:
            line["devices"] = census
        if selection_check is not None:
            line["selection_check"] = selection_check
        line.update(siblings)
        if "weak_scaling" in siblings and world > 1:
            # what ONE GPU of this run does with the whole N = 1 workload (S_cfg seeds): comparable with the N = 1 BENCH record
            w = siblings["weak_scaling"]
            line["n1_equivalent"] = {"value": w["value"] / world, "ms_per_step": w["ms_per_step"],
                                     "how": f"the weak-scaling sibling ({S_cfg} seeds on every GPU, the N = 1 workload) divided by {world}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_torch(args.robot, obstacles, d, W, K)
            line["cpu_baseline_c"] = cpu_baseline_c(args.robot, obstacles, d, W, K)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            line["gpu_over_cpu_c"] = line["value"] / line["cpu_baseline_c"]["value"]
        sys.stdout.flush()
        if saved_stdout_fd is not None:
            os.dup2(saved_stdout_fd, 1)
        print(json.dumps(line), flush=True)
        if saved_stdout_fd is not None:
            os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
