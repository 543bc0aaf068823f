#!/usr/bin/env python3
"""bench.py -- LM-IK iterations/s of the fused MI355X hot path (BASELINE.json metric), one process per GPU.

A "step" is ONE pass of the hot path over one batch of synthetic seeds: a single launch of the fused kernel
(cppf_lm_pose_steps) doing `--lm-steps` K iterations of { pose-only LM step ; clamp to joint limits } on every
(seed, waypoint) row, then the pose-error metrics, self / environment collision masks, joint-limit mask and search cost of
the result, followed by the per-seed summary reduction (8 floats per seed: the x_is_valid maxima, collision counts, summed
cost) -- and, for N > 1, RCCL all-gathers of those summaries (32 KB per rank and step, `--gather-every` = 8 steps per
collective, issued on an auxiliary stream while the next bucket's kernels run).  value = rows * K * steps / wall-time, summed over ranks (every rank owns its own S
seeds: weak scaling, seeds sharded, no data-path collective other than that all-gather).

Workload at N = 1: BASELINE.json configs[3] geometry on one GPU -- Panda (7-DoF), 1024 seeds x 256 waypoints, the two
cuboids of panda__2cubes -- the configuration the metric is quoted on ("1024 seeds x 256 waypoints x 7-DoF at 1 MI355X").
Inputs are already resident in HBM when the timed region starts: the target path is the named reference
problem's (panda__2cubes resampled to 256 waypoints; committed fixture), the seeds are synthetic -- per seed an IK branch
tracking the path, x0 = clamp(q*_s + 0.1 randn) (the construction of the reference's tests/optimization_test.py:82).
`--inputs random` switches to the 8d fall-back (independent q* ~ U(limits) per waypoint, tests/optimization_test.py:136-137),
which is the worst case for the wave-uniform collision broad phase.

Timing: `--prewarm-ms` (60) of untimed launches bring the GPU to its sustained clocks, then W untimed warm-up steps, then
exactly K steps between barrier + synchronize pairs; the maximum over ranks is reported.  Consecutive steps are independent
batches (a ring of four output-buffer sets) alternating between `--streams` (2) HIP streams.

Prints ONE JSON line (rank 0).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak == fp32-input MFMA peak


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


def traffic_from_profiles(robot, S, W, K, collide):
    """HBM bytes per launch of the fused kernel from the rocprofv3 PMC passes committed under profiles/ (separate
    --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  bench.py cannot profile itself, so the figure is the recorded one for the matching workload, else None."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    key = f"{robot}_S{S}_W{W}_K{K}_coll{int(collide)}"
    return rec.get(key, {}).get("hbm_bytes_per_launch")


def make_inputs(robot, S, W, device, seed):
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
    # waypoint 0: damped LM from random starts, re-drawing the seeds that did not reach the pose (up to 12 rounds)
    x = torch.empty((S, d), dtype=torch.float32, device=device)
    todo = torch.ones(S, dtype=torch.bool, device=device)
    for _ in range(12):
        start = (lo + (hi - lo) * (0.1 + 0.8 * torch.rand((S, d), generator=g))).to(device).contiguous()
        r = robot.lm_pose_steps(start, target[0:1], 1e-2, 3.5, 0.35, n_steps=60)
        r = robot.lm_pose_steps(r["x"], target[0:1], 1e-6, 3.5, 0.35, n_steps=10, want_errors=True)
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
        r = robot.lm_pose_steps(x, target[w : w + 1].contiguous(), 1e-6, 3.5, 0.35, n_steps=8, want_errors=True)
        x = r["x"]
        # a branch that loses the path (runs into a joint limit) continues on a branch that did not
        ok = (r["pos_err_m"] < 1e-4) & (r["rot_err_rad"] < 1.75e-3)
        donors = torch.nonzero(ok).reshape(-1)
        if 0 < donors.numel() < S:
            pick = donors[torch.randint(donors.numel(), (S,), generator=gd, device=device)]
            x = torch.where(ok[:, None], x, x[pick]).contiguous()
        branch[:, w] = x
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--robot", default="panda")
    ap.add_argument("--seeds", type=int, default=1024, help="seeds per GPU")
    ap.add_argument("--waypoints", type=int, default=256)
    ap.add_argument("--lm-steps", type=int, default=10, help="K fused LM iterations per launch")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed launches before the W warm-up steps, to reach sustained clocks (0 disables)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps per all-gather of the per-seed summaries (the summaries of G steps travel in one collective)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the independent steps alternate between")
    ap.add_argument("--inputs", choices=["problem", "random"], default="problem",
                    help="problem: the named reference problem's target path + per-seed IK branches (SURVEY 8d); "
                    "random: independent random configurations per waypoint (the 8d fall-back, worst case for the broad phase)")  # fmt: skip
    ap.add_argument("--no-collide", action="store_true", help="FK+Jacobian+LM only (BASELINE configs[1] style)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default=None,
                    help="BASELINE.json configs[1..4] geometry per GPU (default = C4, the configuration the metric is quoted on)")  # fmt: skip
    args = ap.parse_args()
    if args.config is not None:
        preset = {  # robot, seeds per GPU, waypoints, collision fused
            "C2": ("panda", 128, 64, False),  # FK+Jacobian+LM only
            "C3": ("fetch", 512, 256, True),  # + collision fused (fetch__hello has no obstacles: self-collision only)
            "C4": ("panda", 1024, 256, True),
            "C5": ("chain12", 4096, 512, True),
        }[args.config]
        args.robot, args.seeds, args.waypoints = preset[0], preset[1], preset[2]
        args.no_collide = not preset[3]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    saved_stdout_fd = None
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
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
    from cppflow_amd.robots import get_robot
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    robot = get_robot(args.robot)
    d, S, W, K = robot.ndof, args.seeds, args.waypoints, args.lm_steps
    n = S * W
    collide = not args.no_collide
    obstacles = obstacle_arrays(PANDA_2CUBES_OBSTACLES) if (collide and args.config != "C3") else []
    robot.set_obstacles([c for c, _ in obstacles], [T for _, T in obstacles])
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)

    if args.inputs == "problem":
        x0, target, inputs_desc = make_inputs_problem(robot, S, W, device, seed=rank)
    else:
        x0, target = make_inputs(robot, S, W, device, seed=rank)
        inputs_desc = "per waypoint q* ~ U(limits), target = FK(q*), seeds = clamp(q* + 0.1 randn) (SURVEY 8d fall-back inputs)"
    G = max(1, args.gather_every)  # steps per collective (N > 1): the [S,8] summaries of G consecutive steps travel together
    NBUF = max(4, 2 * G)  # ring of output buffer sets: two buckets of G steps, one being filled while the other is on the wire
    x_outs = [torch.empty_like(x0) for _ in range(NBUF)]
    packeds = [torch.empty(robot.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=device) if collide else None
               for _ in range(NBUF)]  # fmt: skip
    x_out, packed = x_outs[0], packeds[0]
    prm = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE
    # per-seed summaries (8 floats per seed) are what every rank needs from every other rank; the fused launch of ring slot b
    # writes summ_all[b] itself (in-kernel epilogue when W is 64 / 128 / 256, else a second reduction kernel issued by the same
    # C call); a bucket of G slots is all-gathered in one collective on an auxiliary stream, overlapping the next bucket's kernels
    summ_all = torch.empty((NBUF, S, 8), dtype=torch.float32, device=device) if collide else None
    use_dist = dist is not None and collide
    gathered = [torch.empty((world, G, S, 8), dtype=torch.float32, device=device) for _ in range(NBUF // G)] if use_dist else None

    if collide:
        plans = [robot.lm_launch_plan(x0, target, n_steps=K, x_out=xo, packed_out=pk, summary_out=summ_all[b], **prm)
                 for b, (xo, pk) in enumerate(zip(x_outs, packeds))]  # fmt: skip
        launch, outputs = plans[0].launch, plans[0].outputs
    else:
        pos_err = torch.empty(n, dtype=torch.float32, device=device)

        def launch():
            robot.lm_pose_steps(x0, target, n_steps=K, clamp=True, x_out=x_out, want_errors=True, **prm)

        outputs = None

    step_no = [0]

    # --streams 2: consecutive steps are independent batches (own output buffers per ring slot), so they alternate between
    # two HIP streams and one launch's tail overlaps the next one's ramp-up
    n_streams = max(1, min(args.streams, NBUF))
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(device))
    aux = torch.cuda.Stream(device=device) if use_dist else None  # the collectives are issued (and waited for) here
    launched = [torch.cuda.Event() for _ in streams]  # "this stream's launches of the bucket are enqueued"
    bucket_done = [None] * (NBUF // G)  # recorded on aux when the bucket's collective has completed

    def step():
        if not collide:
            return launch()
        b = step_no[0] % NBUF
        step_no[0] += 1
        st = streams[b % n_streams]
        if not use_dist:
            plans[b].launch_on(st)
            return
        bucket = b // G
        if bucket_done[bucket] is not None:
            st.wait_event(bucket_done[bucket])  # this bucket's slots were on the wire 2 G steps ago
        plans[b].launch_on(st)
        if b % G == G - 1:  # the bucket is complete: gather its G summaries from every rank
            gather_bucket(bucket)

    def gather_bucket(bucket):
        for k, s_k in enumerate(streams):
            launched[k].record(s_k)
            aux.wait_event(launched[k])
        with torch.cuda.stream(aux):
            work = dist.all_gather_into_tensor(gathered[bucket], summ_all[bucket * G : (bucket + 1) * G], async_op=True)
            work.wait()  # stream-side: aux waits for the communicator's stream
            done = bucket_done[bucket] if bucket_done[bucket] is not None else torch.cuda.Event()
            done.record(aux)
            bucket_done[bucket] = done

    def drain():
        if aux is not None:
            if step_no[0] % G != 0:  # a partly filled bucket: its summaries are exchanged too before the clock stops
                gather_bucket((step_no[0] % NBUF) // G)
                step_no[0] += G - step_no[0] % G  # the next step starts a fresh bucket
            aux.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed pre-warm: the GPU needs tens of milliseconds of continuous work to settle at its sustained clocks and to fill
    # the two-stream pipeline (a 200-step run after 10 warm-up steps measures 54 us per step, the same run after 25 ms of
    # work 46 us); the W warm-up steps of the contract follow it, then exactly K timed steps between barriers.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(50):
            step()
        drain()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    # diagnostic, outside the timed region: host cost of issuing one step (64 steps into an empty queue, no waiting on the GPU)
    th = time.perf_counter()
    for i in range(64):
        step()
    t_enqueued = (time.perf_counter() - th) / 64 * args.steps
    drain()
    torch.cuda.synchronize()
    # kernel duration: a second, untimed pass with HIP events bracketing each launch on the launch stream (torch's
    # current stream is the stream the kernel is launched on); no collective inside the bracket
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for i in range(args.steps):
        kev[i][0].record()
        launch()
        kev[i][1].record()
    torch.cuda.synchronize()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # sanity on the result of the last step (not timed): most rows converged
    if outputs is None:
        outputs = robot.lm_pose_steps(x0, target, n_steps=K, clamp=True, x_out=x_out, want_errors=True, **prm)
        del pos_err
    conv_frac = float((outputs["pos_err_m"] < 1e-4).float().mean().item())

    if rank == 0:
        iters = float(n) * K * args.steps * world
        flops_launch = n * (K * algorithmic_flops_per_row_iter(d)
                            + (algorithmic_flops_collision(robot.n_capsules, robot.n_collision_pairs, len(obstacles)) if collide else 0.0))
        bytes_launch = n * algorithmic_bytes_per_row(d, collide)
        ach_tflops = flops_launch / (kernel_ms * 1e-3) / 1e12
        ach_gbps = bytes_launch / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "LM-IK iterations/sec (seeds x waypoints)",
            "value": iters / elapsed,
            "unit": "LM-IK iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.robot}{'__2cubes geometry' if obstacles else ''}, {S} seeds/GPU x {W} waypoints x {d}-DoF, K={K} fused LM "
                f"iterations per launch" + (" + self/env collision masks + jlim mask + search cost" if collide else " (FK+Jacobian+LM only)"),
                "inputs": inputs_desc,
                "streams": n_streams,
                "prewarm_ms": args.prewarm_ms,
                "host_enqueue_us_per_step": 1e6 * t_enqueued / args.steps,
                "robot": args.robot,
                "seeds_per_gpu": S,
                "waypoints": W,
                "ndof": d,
                "lm_iterations_per_step": K,
                "collision_fused": collide,
                "obstacles": len(obstacles),
                "per_step": "one fused launch incl. the per-seed summary reduction"
                + (f" + async all-gather of the [S,8] summaries, {G} steps per collective" if gathered is not None else ""),
                "allgather_bytes_per_rank_per_step": int(summ_all[0].numel() * 4) if gathered is not None else 0,
                "steps_per_allgather": G if gathered is not None else 0,
                "converged_frac_pos_err_lt_1e-4": conv_frac,
            },
            "roofline": {
                "bound": "mfma",
                "achieved": ach_tflops,
                "peak": F32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach_tflops / F32_PEAK_TFLOPS,
                "traffic": traffic_from_profiles(args.robot, S, W, K, collide),
                "kernel": "lm_fused_kernel",
                "kernel_ms": kernel_ms,
                "note": "binding resource is the fp32 FMA rate (157.3 TFLOP/s: vector peak == f32-input MFMA peak); the "
                "kernel issues VALU FMAs, no MFMA instructions.  achieved = algorithmic flops (SURVEY 8d) / kernel time",
                "hbm": {"achieved": ach_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gbps / HBM_PEAK_GBPS},
                # the isolated launch has only 4 rows per SIMD lane in flight (262 144 rows on 65 536 lanes): its loop is
                # partly latency-bound; with two independent batches in flight (the timed configuration) the same kernels
                # deliver this rate per GPU
                "at_step_rate": {
                    "achieved": flops_launch / (elapsed / args.steps) / 1e12,
                    "frac": flops_launch / (elapsed / args.steps) / 1e12 / F32_PEAK_TFLOPS,
                    "batches_in_flight": n_streams,
                },
            },
        }
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
