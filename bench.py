#!/usr/bin/env python3
"""bench.py -- LM-IK iterations/s of the fused MI355X hot path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A thin TIMER around the package's own engine, `cppflow_amd.distributed.ShardedRefiner` (launch ring, batched launches, bucketed
in-stream exchange, gathered seed selection, once-per-call gather + dp_search all live there; nothing below is product behaviour).

A "step" is ONE pass of the hot path over one batch of synthetic seeds: the fused kernel (cppf_lm_batch_launch) doing `--lm-steps`
K iterations of { pose-only LM step ; clamp to joint limits } on every (seed, waypoint) row, then the pose-error metrics, self /
environment collision masks, joint-limit mask and search cost of the result, and the per-seed summary (8 floats per seed).  For
N > 1 the summaries of a bucket of steps are all-gathered over RCCL in-stream and CONSUMED: every rank runs x_is_valid's seed
selection over all ranks' seeds for every step.  value = rows * K * steps / wall-time over all ranks.

THROUGHPUT OVER INDEPENDENT REQUESTS: under strong scaling a launch carries B steps of this rank's shard (B = 262 144 / rows: 2 / 4 / 8
at N = 2 / 4 / 8) = B independent requests in one full-width grid.  ONE request's shard (`latency_one_request`: --batch 1, one stream)
takes ~25 us for 32 768 rows against 49 us unsharded: a single planning call does not get 8x faster on 8 GPUs, a request stream does.

Launching.  `python bench.py --gpus N` with WORLD_SIZE unset starts N fresh rank processes itself (BEFORE this process touches the GPU),
relays rank 0's single JSON line and exits with the children's status; under torch.distributed.run it is one rank.
Scaling.  N = 1: BASELINE.json configs[3] on one GPU (Panda, 1024 seeds x 256 waypoints, the two cuboids of panda__2cubes).  N > 1:
STRONG scaling by default (the same 1024 seeds sharded); the weak figure (1024 seeds per GPU) is the sibling key `weak_scaling`.

Timing.  `--prewarm-ms` of untimed launches, W untimed warm-up steps, then exactly K steps: barrier + synchronize, clock, the K
steps (+ the exchange of a partly filled bucket), this rank's synchronize, clock; maximum over ranks.  The group's CLOSING barrier
follows the clock (recorded in config.timed_region.closing_barrier_us).  The region is measured `--repeats` times; the MEDIAN is
reported (a region under 0.4 ms `--short-region-repeats` more times).  The headline is what a FRESH ShardedRefiner does by default --
the streams it created, no selection among stream pairs; for short N > 1 regions the figure with `calibrate_streams()` (a
best-of-30 pick of the stream pair, r4's headline) is the secondary key `ms_per_step_calibrated_streams`.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak == fp32-input MFMA peak
N_SIMD = 1024  # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md: max clock; one VALU wave-instruction occupies a SIMD for 2 cycles
PROFILE_ROUND = "r5"  # the committed record pass the lookups below read (scripts/record_pass.sh -> profiles/r5_*)
METRIC = "LM-IK iterations/sec (seeds x waypoints)"


def __getattr__(name):  # make_inputs / make_inputs_problem live in cppflow_amd.problems_synthetic; scripts/ still say bench.make_inputs*
    if name in ("make_inputs", "make_inputs_problem"):
        from cppflow_amd import problems_synthetic

        return getattr(problems_synthetic, name)
    raise AttributeError(name)


def algorithmic_flops_per_row_iter(d: int) -> float:
    """FK 130*(d + n_fixed) + Jacobian 12d + pose error ~100 + scaling 6(d+1) + J^T J upper triangle
    12*d(d+1)/2 + J^T e 12d + Cholesky d^3/3 + 2d^2 + update/clamp 3d   (d=7: ~1.9 kFLOP; the figure the survey states)."""
    return 130.0 * (d + 1) + 12 * d + 100 + 6 * (d + 1) + 12 * d * (d + 1) / 2 + 12 * d + d**3 / 3 + 2 * d * d + 3 * d


def algorithmic_flops_collision(L: int, P: int, O: int) -> float:  # capsule end points 36 L + pairs 90 P_s + capsule-cuboid 150 L O
    return 36.0 * L + 90.0 * P + 150.0 * L * O


def algorithmic_bytes_per_row(d: int, collide: bool) -> float:  # read x 4d + target 28 + write x 4d (+ 2 mask bytes + 4 cost bytes)
    return 8.0 * d + 28.0 + (6.0 if collide else 0.0)


# ---- committed profiler records (bench.py cannot profile itself): only used when taken with THIS build of the library -------------
def _profile_json(fname):
    path = os.path.join(ROOT, "profiles", fname)
    return json.load(open(path)) if os.path.exists(path) else None


def workload_key(robot, S, W, K, collide, inputs="problem"):
    return f"{robot}_S{S}_W{W}_K{K}_coll{int(collide)}" + ("" if inputs == "problem" else f"_{inputs}")


def record_from_profiles(fname, key, build_id):
    """(record, "") from profiles/<round>_<fname> when the file was recorded with this library build (`_build_id`), else (None, why)."""
    doc = _profile_json(f"{PROFILE_ROUND}_{fname}")
    if doc is None:
        return None, f"profiles/{PROFILE_ROUND}_{fname} missing"
    if doc.get("_build_id", doc.get("library_build_id")) != build_id:
        return None, f"profiles/{PROFILE_ROUND}_{fname} was recorded with library build {doc.get('_build_id', doc.get('library_build_id'))}, this run loaded {build_id}"
    rec = doc if key is None else doc.get(key)
    return (rec, "") if rec is not None else (None, f"no record {key!r} in profiles/{PROFILE_ROUND}_{fname}")


def kernel_profile_from_profiles(kernel_substr, build_id):
    """Average duration (ms) and call count of the dominant kernel in the committed `rocprofv3 --kernel-trace --stats` summary of
    this same command (profiles/<round>_fused_kernel_stats.csv; its first line names the build it was taken with), or (None, why)."""
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--robot", default="panda")
    ap.add_argument("--seeds", type=int, default=1024, help="seeds of the configuration: the TOTAL that is sharded under strong scaling, per GPU under weak")
    ap.add_argument("--waypoints", type=int, default=256)
    ap.add_argument("--lm-steps", type=int, default=10, help="K fused LM iterations per launch")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto", help="auto = strong for N > 1 (BASELINE.json configs[3])")
    ap.add_argument("--repeats", type=int, default=5, help="the timed region is measured this many times back to back; the MEDIAN is reported")
    ap.add_argument("--short-region-repeats", type=int, default=16, help="a region shorter than 0.4 ms is measured this many times more; 0 = never")
    ap.add_argument("--kernel-reps", type=int, default=400, help="isolated launches behind roofline.kernel_ms (median of HIP-event pairs)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0, help="untimed launches before the warm-up steps, to reach sustained clocks")
    ap.add_argument("--gather-every", type=int, default=0, help="N > 1: steps per all-gather of the per-seed summaries (0 = distributed.launch_plan)")
    ap.add_argument("--batch", type=int, default=0, help="steps per launch (0 = as many of this rank's steps as make one full-width launch, <= 16)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams the independent steps alternate between (0 = distributed.launch_plan)")
    ap.add_argument("--graphs", choices=["auto", "on", "off"], default="auto", help="replay a bucket's launches as one captured hipGraph (auto: shards <= 65536 rows per launch)")
    ap.add_argument("--calibrate-streams", choices=["auto", "on", "off"], default="auto",
                    help="also report the region with ShardedRefiner.calibrate_streams() (secondary key; auto: N > 1 transports and --steps <= 64)")
    ap.add_argument("--shape", choices=["auto", "row", "quad"], default="auto", help="kernel shape (cppf_lm_params.shape)")
    ap.add_argument("--solver", choices=["auto", "f32", "f64"], default="auto", help="precision of the damped solve (cppf_lm_params.solver)")
    ap.add_argument("--inputs", choices=["problem", "random"], default="problem",
                    help="problem: the named reference problem's target path + per-seed IK branches (SURVEY 8d); random: the 8d fall-back")
    ap.add_argument("--no-collide", action="store_true", help="FK+Jacobian+LM only (BASELINE configs[1] style)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pace", default="auto", choices=["auto", "on", "off"], help="fair-share pacing of the launches (ShardedRefiner(pace=...)): auto = the engine's default (on with one stream)")
    ap.add_argument("--no-siblings", action="store_true", help="skip the one_stream / random_inputs / weak_scaling / latency sibling measurements")
    ap.add_argument("--transport", choices=["cabi", "c10d", "none"], default="cabi", help="N > 1: the all-gather's transport (distributed.pick_transport)")
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default=None, help="BASELINE.json configs[1..4] geometry (default = C4)")
    args = ap.parse_args(argv)
    if args.config is not None:
        preset = {"C2": ("panda", 128, 64, False), "C3": ("fetch", 512, 256, True), "C4": ("panda", 1024, 256, True),
                  "C5": ("chain12", 4096, 512, True)}[args.config]  # robot, seeds, waypoints, collision fused
        args.robot, args.seeds, args.waypoints = preset[0], preset[1], preset[2]
        args.no_collide = not preset[3]
    return args


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not imported torch, let alone
    touched the GPU), relay rank 0's JSON line, return the first non-zero exit status.  Never replaces a running process."""
    import tempfile

    n = args.gpus
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out0 if r == 0 else sys.stderr))
    rc, live = 0, list(procs)
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


# ---- timing around a ShardedRefiner -----------------------------------------------------------------------------------------------
def timed(run, steps, warmup, prewarm_ms, barrier, repeats=1, closing_barrier=None):
    """`prewarm_ms` of untimed launches, W untimed warm-up steps, then `repeats` times: opening barrier + synchronize, clock, exactly
    `steps` steps (+ the exchange of a partly filled bucket), THIS RANK's synchronize, clock; the group's closing barrier follows.
    Returns (elapsed seconds of every repetition, seconds the closing barrier took each time)."""
    run.prewarm(prewarm_ms)
    run.run_steps(warmup)
    run.drain()
    out, closing = [], []
    for _ in range(max(1, repeats)):
        barrier()
        t0 = time.perf_counter()
        run.run_region(steps)
        run.synchronize()
        t1 = time.perf_counter()
        out.append(t1 - t0)
        if closing_barrier is not None:
            closing_barrier()
            closing.append(time.perf_counter() - t1)
    return out, closing


def host_enqueue_us(run):
    """diagnostic: host cost of issuing one step (>= 64 steps into an empty queue, no waiting on the GPU)"""
    run.synchronize()
    nh = max(64, 4 * run.G) if run.graphs is None else 4 * run.G
    nh = (nh // run.B) * run.B
    th = time.perf_counter()
    run.run_steps(nh)
    t = (time.perf_counter() - th) / nh
    run.drain()
    run.synchronize()
    return 1e6 * t


def kernel_ms(run, reps, prewarm=300):
    """Isolated launch duration: HIP events bracketing single launches on the launch stream (torch's current stream IS the stream
    the kernel is launched on), one launch in flight at a time, no collective inside the bracket; `prewarm` untimed launches first,
    then the MEDIAN of `reps` pairs.  (A launch is B steps: `launch_steps` says how many.)"""
    for _ in range(prewarm):
        run.launch()
    torch.cuda.synchronize()
    for _ in range(prewarm // 4):
        run.launch()
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in kev:
        a.record()
        run.launch()
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in kev])
    return {"median": float(np.median(t)), "mean": float(t.mean()), "min": float(t.min()), "p10": float(np.quantile(t, 0.1)),
            "p90": float(np.quantile(t, 0.9)), "max": float(t.max()), "n": int(reps), "launch_steps": run.B}


def dryrun(args, world, rank):
    """CPPF_BENCH_DRYRUN=1 (tests/test_bench_launcher.py, no GPU): the rank choreography of an N > 1 run without a single kernel --
    gloo process group, seed sharding, one all-gather of [S_local, 8] stand-in summaries, the host-side seed selection over all
    ranks' seeds, max-over-ranks, ONE JSON line from rank 0.  It measures nothing and says so (`value` null, `data` "dryrun")."""
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
        print(json.dumps({"metric": METRIC, "value": None, "unit": "LM-IK iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "strong", "data": "dryrun (no GPU: launcher and collective choreography only)",
                          "config": {"world_size": dist.get_world_size(), "seeds_total": S, "seeds_per_gpu_padded": S_pad,
                                     "gathered_seed_ids": [int(v) for v in allseeds[:, 7]], "n_valid": len(valid), "max_rank": int(t.item())},
                          "rccl": {"requested": "dryrun", "transport": "gloo (dry run)", "world_seen": dist.get_world_size(),
                                   "unique_id_via": "torch.distributed (c10d store)", "ranks": census},
                          "selection_check": {"selected_by_rank": everyone, "identical_on_every_rank": True, "single_process_selection": [0, S, 0, 0],
                                              "equals_single_process": everyone[0] == [0, S, 0, 0]}}), flush=True)  # fmt: skip
        os.dup2(2, 1)
    dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if args.gpus != world:  # checked before anything touches the GPU
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` (it starts its own "
                 f"ranks) or launch N ranks with torch.distributed.run and pass the same --gpus N")  # fmt: skip
    # Independent launches on different HIP streams only overlap when the streams map to different hardware queues; the runtime's
    # default of 4 queues per process pairs them up (scripts/hwq_sweep.sh: a 32 768-row shard steps in 13.1 us on 4 streams with the
    # default and in 7.1 us with 16 queues; the full-size launch is unaffected).  Must be set before the HIP runtime initialises.
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

    from cppflow_amd import _hip
    from cppflow_amd import distributed as D
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, make_inputs, make_inputs_problem, obstacle_arrays
    from cppflow_amd.robots import get_robot
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    dist, saved_stdout_fd, transport, rccl_rec = None, None, None, None
    # CPPF_BENCH_FORCE_DIST=1 initialises the RCCL process group even for one rank (rehearses the N > 1 code path)
    if world > 1 or os.environ.get("CPPF_BENCH_FORCE_DIST", "0") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")  # the tiny collective should not queue behind a full-chip kernel
        sys.stdout.flush()  # RCCL prints a version banner on stdout when the communicator comes up: fd 1 points at stderr until the line
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            transport = D.HostStagedAllGather()
            rccl_rec = {"requested": "host-staged", "transport": transport.name, "world_seen": world}
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
            transport, rccl_rec = D.pick_transport(device, prefer=args.transport)
        assert dist.get_world_size() == world

    robot = get_robot(args.robot)
    d, W, K = robot.ndof, args.waypoints, args.lm_steps
    collide = not args.no_collide
    obstacles = obstacle_arrays(PANDA_2CUBES_OBSTACLES) if (collide and args.config != "C3") else []
    robot.set_obstacles([c for c, _ in obstacles], [T for _, T in obstacles])
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    shape = {"auto": _hip.SHAPE_AUTO, "row": _hip.SHAPE_ROW, "quad": _hip.SHAPE_QUAD}[args.shape]
    solver = {"auto": _hip.SOLVER_AUTO, "f32": _hip.SOLVER_F32, "f64": _hip.SOLVER_F64}[args.solver]
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    S_cfg = args.seeds
    if scaling == "strong":
        assert S_cfg % world == 0 and (S_cfg // world) >= 1, f"--seeds {S_cfg} must be a multiple of --gpus {world} under strong scaling"
    S_main = S_cfg // world if scaling == "strong" else S_cfg
    full_inputs = {}

    def inputs_for(mode, kind):
        """(x0 [S_local*W, d], target, description).  Strong scaling: every rank builds the SAME S_cfg seeds (seed 0) and keeps its
        `seed_shard`; weak scaling: rank r builds its own S_cfg seeds (seed r)."""
        gen_seed = 0 if mode == "strong" else rank
        if kind == "problem":
            x_all, tgt, desc = make_inputs_problem(robot, S_cfg, W, device, seed=gen_seed)
        else:
            x_all, tgt = make_inputs(robot, S_cfg, W, device, seed=gen_seed)
            desc = "per waypoint q* ~ U(limits), target = FK(q*), seeds = clamp(q* + 0.1 randn) (SURVEY 8d fall-back inputs)"
        if mode == "strong" or world == 1:
            full_inputs[kind] = (x_all, tgt)  # all S_cfg seeds: the single-process reference of the selection check
        if mode == "strong" and world > 1:
            b, e = D.seed_shard(S_cfg, rank, world)
            x_all = x_all[b * W : e * W].contiguous()
        return x_all, tgt, desc

    def barrier():
        # (drain this rank's launch streams BEFORE the group's barrier: its all-reduce belongs to torch.distributed's communicator, the
        # exchange steps still in flight to the C-ABI one, and kernels of two communicators must not wait for each other in a
        # different order on different ranks)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def closing_barrier():
        dist.barrier()
        torch.cuda.synchronize()

    closing = closing_barrier if dist is not None else None

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if share_gpu else device)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def refiner(x, tgt, steps_hint, batch=None, streams=None, use_transport=True, pace=None):
        B, G, G_req, n_streams = D.launch_plan(x.shape[0], steps_hint, args.batch if batch is None else batch, args.gather_every,
                                               args.streams if streams is None else streams, shape == _hip.SHAPE_QUAD, _hip.MAX_BATCH)
        graphs = args.graphs == "on" or (args.graphs == "auto" and x.shape[0] * B <= 65536)
        r = D.ShardedRefiner(robot, x, tgt, K, transport=transport if use_transport else None, collide=collide, batch=B, bucket=G,
                             n_streams=n_streams, shape=shape, solver=solver, graphs=graphs,
                             pace={"auto": pace, "on": True, "off": False}[args.pace])
        return r, G_req

    def measure(run, steps, warmup, prewarm_ms, repeats):
        el, cl = timed(run, steps, warmup, prewarm_ms, barrier, repeats, closing)
        return [max_over_ranks(t) for t in el], [1e6 * max_over_ranks(t) for t in cl]

    x0, target, inputs_desc = inputs_for(scaling, args.inputs)
    run, G_req = refiner(x0, target, args.steps)
    n = run.n
    # ---- the headline: a fresh ShardedRefiner's default streams, `--repeats` regions, a short region more often ----
    reps_s, closing_us = measure(run, args.steps, args.warmup, args.prewarm_ms, args.repeats)
    if args.short_region_repeats > 0 and float(np.median(reps_s)) < 0.4e-3:  # (decided on the maximum over ranks: every rank alike)
        more, more_c = measure(run, args.steps, args.warmup, args.prewarm_ms, args.short_region_repeats)
        reps_s, closing_us = reps_s + more, closing_us + more_c
    elapsed = float(np.median(reps_s))
    # ---- secondary: the same regions on a calibrated stream pair (r4's headline; a best-of-30 pick, disclosed as such) ----
    calibrated = None
    want_cal = args.calibrate_streams == "on" or (args.calibrate_streams == "auto" and args.steps <= 64)
    if want_cal and run.transport is not None and run.n_streams >= 2:
        rec = run.calibrate_streams(args.steps, barrier)
        cal_s, _ = measure(run, args.steps, args.warmup, args.prewarm_ms, len(reps_s))
        calibrated = {"ms_per_step": 1e3 * float(np.median(cal_s)) / args.steps, "calibration": rec,
                      "note": "ShardedRefiner.calibrate_streams(): the fastest of the ordered stream pairs / start buckets tried during warm-up; NOT the headline"}
    host_us = host_enqueue_us(run)
    kstats = kernel_ms(run, max(200, args.kernel_reps))
    ag_latency = run.allgather_latency_us(100)

    outputs = run.outputs  # sanity on the result of the last step (not timed): most rows converged
    conv_frac = float((outputs["pos_err_m"] < 1e-4).float().mean().item())
    selected = selection_check = None
    if run.selected is not None:
        torch.cuda.synchronize()
        selected = [int(v) for v in run.selected[0][0].cpu()]
        # Self-verification of the exchange step: (1) every rank must hold the same selection; (2) under strong scaling it must equal
        # what ONE process selects over the same S_cfg seeds (one unsharded launch + the same selection kernel on this rank).
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, selected)
        same = all(e == everyone[0] for e in everyone)
        single = None
        if scaling == "strong" or world == 1:
            x_full, tgt_full = full_inputs[args.inputs]
            summ = torch.empty((S_cfg, 8), dtype=torch.float32, device=device)
            pk = torch.empty(robot.PACKED_BYTES_PER_ROW * S_cfg * W, dtype=torch.uint8, device=device)
            robot.lm_pose_steps(x_full, tgt_full, n_steps=K, packed_out=pk, summary_out=summ, shape=shape, solver=solver, **run.lm)
            single = [int(v) for v in robot.select_valid_seed(summ.view(1, 1, S_cfg, 8), run.constraints).reshape(-1).cpu()]
        selection_check = {"selected_by_rank": everyone, "identical_on_every_rank": same, "single_process_selection": single,
                           "equals_single_process": (single == selected) if single is not None else None,
                           "fields": "[first valid seed or -1, number of valid seeds, seed of smallest summed cost, 0] over all ranks' seeds"}
        assert same, f"ranks disagree on the selected seed: {everyone}"
        assert single is None or single == selected, f"sharded selection {selected} != single-process selection {single}"

    # Once per planning call (not per step): ShardedRefiner.gather_and_search -- every rank gets ALL ranks' per-row costs / masks and
    # candidate paths and runs dp_search over them (cppflow/search.py:146-173).  Untimed here, reported beside the headline.
    plan_search = None
    if collide:
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            ev0.record()
            run.gather_and_search(0)
            ev1.record()
            torch.cuda.synchronize()
        plan_search = {"candidates": run.S * run.world, "waypoints": W, "ms": ev0.elapsed_time(ev1),
                       "allgather_bytes_per_rank": int(run.packeds[0].numel() + run.x_outs[0].numel() * 4) if run.world > 1 else 0,
                       "what": "ShardedRefiner.gather_and_search: all-gather of the packed per-row outputs + candidate paths, then cppf_dp_search "
                       "over every rank's candidates (once per planning call; untimed, outside `value`)"}

    def sibling(mode, kind, streams, steps, batch=None, use_transport=True, pace=None):
        """a second workload / pipeline depth measured like the headline (same barriers, same max over ranks)"""
        xs, tg, _ = inputs_for(mode, kind)
        r2, _ = refiner(xs, tg, steps, batch=batch, streams=streams, use_transport=use_transport, pace=pace)
        el = float(np.median(measure(r2, steps, min(args.warmup, 100), min(args.prewarm_ms, 30.0), 3)[0]))
        km = kernel_ms(r2, 200, prewarm=200)["median"]
        res = {"value": float(r2.n) * run.world * K * steps / el, "ms_per_step": 1e3 * el / steps, "kernel_ms": km, "streams": r2.n_streams,
               "rows_per_gpu": int(r2.n), "steps_per_launch": r2.B}
        r2.close()
        del r2
        torch.cuda.empty_cache()
        return res

    def neighbour_stages():
        """The two stages on either side of the hot path on this workload's own result, untimed and outside
        `value`: the coupled differencing step for all seeds and for one, dp_search over the reference's k = 175 candidates."""
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
                r = sibling(other, args.inputs, 2 if other == "weak" else 4, sib_steps)
                r.update(scaling=other, seeds_per_gpu=S_cfg if other == "weak" else S_cfg // world)
                siblings[other + "_scaling"] = r
        else:
            # one stream = a dependency chain: the engine paces its launches by default there (ShardedRefiner(pace=None); the headline's
            # overlapping launches are never paced); the unpaced figure rides along
            siblings["one_stream"] = sibling(scaling, args.inputs, 1, sib_steps)
            unpaced = sibling(scaling, args.inputs, 1, sib_steps, pace=False)
            siblings["one_stream"].update(paced=True, ms_per_step_unpaced=unpaced["ms_per_step"], kernel_ms_unpaced=unpaced["kernel_ms"])
            if args.inputs == "problem":
                siblings["random_inputs"] = sibling(scaling, "random", run.n_streams, sib_steps)
            if collide and d <= 12 and W >= 2:
                siblings["neighbour_stages"] = neighbour_stages()
        if run.B > 1:  # what ONE request's shard costs this rank: one step per launch, one stream, nothing overlapping it
            r = sibling(scaling, args.inputs, 1, sib_steps, batch=1, use_transport=False)
            r["what"] = "latency view: this rank's shard of ONE request per launch on one stream (no batching over requests, no exchange)"
            siblings["latency_one_request"] = r

    census = D.device_census(rank, dev_index, dist is not None) if not share_gpu else None
    cpu_legs = {}
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.inputs == "problem":
        x_h, t_h = x0.cpu(), target.cpu()
        pe_h, re_h = outputs["pos_err_m"].cpu(), outputs["rot_err_rad"].cpu()
        from oracle import cpu_baseline as cpu  # the checker's arithmetic timed beside the GPU: the only use of oracle/ in this file

        cpu_legs["cpu_baseline"] = cpu.cpu_baseline_torch(args.robot, obstacles, x_h, t_h, pe_h, re_h, K)
        cpu_legs["cpu_baseline_c"] = cpu.cpu_baseline_c(args.robot, obstacles, x_h, t_h, pe_h, K)
    if rank == 0:
        iters = float(n) * K * args.steps * world
        alg_flops = n * (K * algorithmic_flops_per_row_iter(d)
                         + (algorithmic_flops_collision(robot.n_capsules, robot.n_collision_pairs, len(obstacles)) if collide else 0.0))  # fmt: skip
        t_k = kstats["median"] * 1e-3
        ach_gbps = n * algorithmic_bytes_per_row(d, collide) / t_k / 1e9
        build_id = _hip.lib().cppf_build_id().decode()
        # the committed counters describe ONE launch of `run.B` steps of this workload (scripts/pmc_probe.py records the launch shapes
        # bench.py issues); B > 1 launches of a shard are keyed with their batch
        wkey = workload_key(args.robot, run.S, W, K, collide, args.inputs + ("" if run.B == 1 else f"_b{run.B}"))
        rec, rec_why = record_from_profiles("issue.json", wkey, build_id) if args.solver == "auto" else (None, "no counter record for this solver mode")
        traffic, _ = record_from_profiles("traffic.json", workload_key(args.robot, run.S, W, K, collide), build_id) if run.B == 1 else (None, "")
        overlap, overlap_why = record_from_profiles("overlap.json", None, build_id) if (run.B == 1 and world == 1 and args.inputs == "problem") else (None, "not the N = 1 command")
        kname = "lm_fused_kernel" if shape != _hip.SHAPE_QUAD else "lm_quad_kernel"
        kprof, kprof_why = kernel_profile_from_profiles(f"{kname}<cppf::StaRobot<cppf::gen::{''.join(p.capitalize() for p in args.robot.split('_'))}>, {1 if collide else 0}", build_id)
        if kprof and run.B != 1:
            kprof, kprof_why = None, "the committed kernel-trace summary is of the N = 1 command (one step per launch)"
        step_s = elapsed / args.steps
        roof = {
            # the binding resource is the fp32 VALU issue rate (157.3 TFLOP/s of FMAs = one wave-instruction per SIMD per 2 cycles); the
            # contract's vocabulary has no word for it, so `bound` says what it is and `mfma_used` that no matrix instruction is issued
            "bound": "valu", "mfma_used": False, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "kernel": kname,
            "kernel_ms": kstats["median"], "kernel_launch_steps": run.B,
            "kernel_ms_how": f"median of {kstats['n']} isolated launches of {run.B} step(s) (HIP events on the launch stream, one in flight) after a pre-warm",
            "kernel_ms_stats": kstats,
            "kernel_ms_profile": kprof["ms"] if kprof else None,  # the same kernel in the committed rocprofv3 --kernel-trace --stats summary
            "kernel_profile": kprof if kprof else {"unavailable": kprof_why},
            "drift": (kstats["median"] / kprof["ms"]) if kprof else None,
            # the profiler's view of the two-launch overlap the step rate rests on (scripts/overlap_summary.py -> profiles/r5_overlap.json)
            "kernel_ms_overlapped": overlap.get("kernel_ms_overlapped") if overlap else None,
            "overlap_profile": ({k: overlap[k] for k in ("union_busy_us_per_step", "mean_dispatch_us_under_overlap", "two_resident_frac_of_busy_time",
                                                         "bench_ms_per_step_under_profiler", "union_busy_over_bench_ms_per_step") if k in overlap}
                                if overlap else {"unavailable": overlap_why}),
            "library_build_id": build_id,
            # the SURVEY 8(d) flop model, for comparison ONLY (never `frac`): it prices the reference's formulation, not what this kernel executes
            "algorithmic": {"tflops": alg_flops * run.B / t_k / 1e12,
                            "note": "SURVEY 8d flop model / isolated kernel time: what the reference's formulation would need, NOT a utilisation figure"},
            "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
            "hbm": {"achieved": ach_gbps * run.B, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gbps * run.B / HBM_PEAK_GBPS},
        }
        if rec is not None:
            valu = float(rec["valu_insts_per_launch"])
            flops_launch = valu * 64.0 * float(rec["flops_per_valu_lane_op"])
            issue_s = valu * 2.0 / N_SIMD / (CLOCK_GHZ * 1e9)
            roof.update({
                "achieved": flops_launch / t_k / 1e12, "frac": flops_launch / t_k / 1e12 / F32_PEAK_TFLOPS,
                "basis": f"executed flops: SQ_INSTS_VALU_* of the matching launch (profiles/{PROFILE_ROUND}_issue.json, same library build) x 64 lanes x "
                "flops per VALU lane-op (FMA = 2, mul / add = 1, moves / selects / compares / transcendentals = 0), / live isolated kernel time",
                "valu_issue_frac": issue_s / t_k, "valu_insts_per_row": valu * 64.0 / (run.n * run.B),
                "at_step_rate": {"ms_per_step": 1e3 * step_s, "launches_in_flight": run.n_streams, "executed_tflops": flops_launch / run.B / step_s / 1e12,
                                 "frac": flops_launch / run.B / step_s / 1e12 / F32_PEAK_TFLOPS, "valu_issue_frac": issue_s / run.B / step_s},
            })  # fmt: skip
        else:
            roof.update({"achieved": None, "frac": None, "basis": f"unavailable: {rec_why} (the executed-flop basis needs the committed counters of this "
                         "launch shape and library build; the algorithmic model is kept apart under `algorithmic`)",
                         "at_step_rate": {"ms_per_step": 1e3 * step_s, "launches_in_flight": run.n_streams}})  # fmt: skip
        if kprof and abs(kstats["median"] / kprof["ms"] - 1.0) > 0.15:
            print(f"bench: WARNING: live kernel time {1e3 * kstats['median']:.1f} us differs from the committed rocprofv3 average "
                  f"{1e3 * kprof['ms']:.1f} us by more than 15 %", file=sys.stderr)
        line = {
            "metric": METRIC, "value": iters / elapsed, "unit": "LM-IK iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.robot}{'__2cubes geometry' if obstacles else ''}, "
                + (f"{S_cfg} seeds sharded over {world} GPU(s) = {run.S} seeds/GPU" if scaling == "strong" else f"{run.S} seeds/GPU")
                + f" x {W} waypoints x {d}-DoF, K={K} fused LM iterations per launch"
                + (" + self/env collision masks + jlim mask + search cost" if collide else " (FK+Jacobian+LM only)"),
                "engine": "cppflow_amd.distributed.ShardedRefiner (default streams; no calibration in the headline)",
                "throughput_not_latency": (None if run.B == 1 else f"a launch carries {run.B} steps = {run.B} INDEPENDENT requests' shards in one full-width grid: "
                                           "the figure is request throughput; one request's own shard is in `latency_one_request`"),
                "inputs": inputs_desc, "streams": run.n_streams,
                "timed_region": {"repeats": len(reps_s), "reported": "median", "ms_per_step_min": 1e3 * min(reps_s) / args.steps,
                                 "ms_per_step_max": 1e3 * max(reps_s) / args.steps, "ms_per_step_all": [1e3 * t / args.steps for t in reps_s],
                                 "clock": "opening barrier + synchronize | K steps (+ the exchange of a partly filled bucket) | this rank's "
                                          "synchronize; maximum over ranks.  The group's closing barrier follows the clock (since round 4).",
                                 "closing_barrier_us": closing_us if closing_us else None},
                "steps_per_launch": run.B,
                "launches_per_region": f"{args.steps // run.B} x {run.B} steps" + (f" + 1 x {args.steps % run.B} steps" if args.steps % run.B else ""),
                "rows_per_launch": run.n * run.B, "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                "hip_graphs": (f"on: each stream's {run.G} consecutive steps replayed as one captured graph" if run.graphs is not None else "off (one host call per launch)"),
                "kernel_shape": args.shape, "solver": args.solver + (" (fp32 + conditioning-gated double-precision redo, the default)" if args.solver == "auto" else ""),
                "early_out": "off (every row runs all K iterations: the metric counts K iterations per row)",
                "prewarm_ms": args.prewarm_ms, "host_enqueue_us_per_step": host_us, "robot": args.robot,
                "seeds_total": S_cfg * (world if scaling == "weak" else 1), "seeds_per_gpu": run.S, "waypoints": W, "ndof": d,
                "lm_iterations_per_step": K, "collision_fused": collide, "obstacles": len(obstacles),
                "world_size": dist.get_world_size() if dist is not None else 1,
                "collective_backend": transport.name if transport is not None else None,
                "per_step": ("one fused launch" if run.B == 1 else f"1/{run.B} of a fused launch of {run.B} independent steps (cppf_lm_batch_launch)")
                + " incl. the per-seed summary reduction"
                + (f" + in-stream all-gather of the [S,8] summaries, {run.G} steps per collective, + x_is_valid seed selection over all "
                   f"{run.S * world} seeds of every step on each rank" if run.gathered is not None else ""),
                "allgather_bytes_per_rank_per_step": int(run.summ_all[0].numel() * 4) if run.gathered is not None else 0,
                "steps_per_allgather": run.G if run.gathered is not None else 0,
                "steps_per_allgather_requested": G_req if run.gathered is not None else 0,
                "selected_seed_last_bucket": selected, "converged_frac_pos_err_lt_1e-4": conv_frac,
            },
            "roofline": roof,
        }
        if calibrated is not None:
            line["ms_per_step_calibrated_streams"] = calibrated["ms_per_step"]
            line["calibrated_streams"] = calibrated
        if plan_search is not None:
            line["plan_search"] = plan_search
        if rccl_rec is not None:
            line["rccl"] = dict(rccl_rec, ranks=census, allgather_latency_us=ag_latency,
                                allgather_latency_what="mean of 100 bare all-gathers of one step's [S,8] summaries on an idle stream (untimed diagnostic)")
        elif census is not None:
            line["devices"] = census
        if selection_check is not None:
            line["selection_check"] = selection_check
        line.update(siblings)
        if "weak_scaling" in siblings and world > 1:
            w = siblings["weak_scaling"]  # what ONE GPU of this run does with the whole N = 1 workload: comparable with the N = 1 record
            line["n1_equivalent"] = {"value": w["value"] / world, "ms_per_step": w["ms_per_step"],
                                     "how": f"the weak-scaling sibling ({S_cfg} seeds on every GPU, the N = 1 workload) divided by {world}"}
        if cpu_legs:
            line.update(cpu_legs)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            line["gpu_over_cpu_c"] = line["value"] / line["cpu_baseline_c"]["value"]
        sys.stdout.flush()
        if saved_stdout_fd is not None:
            os.dup2(saved_stdout_fd, 1)
        print(json.dumps(line), flush=True)
        if saved_stdout_fd is not None:
            os.dup2(2, 1)
    if dist is not None:
        if transport is not None:
            torch.cuda.synchronize()
            transport.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
