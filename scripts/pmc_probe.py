#!/usr/bin/env python3
"""Launch a fixed list of labelled variants (three rounds; the last one is summarised) so that a rocprofv3 --pmc pass attributes
counters per dispatch.  The labels, in dispatch order, go to gpurun_out/rec/pmc_labels.json for scripts/summarize_profiles.py.

Variants (Panda, 2 cuboids, K = 10 unless stated):
  A  row shape, 262 144 rows, no collision (random inputs)          B  the same, K = 20
  C  row shape + collision, random inputs                             D  collision_masks alone, random inputs
  E  the bench launch: row + collision + summary, problem inputs     F  E with the fp64 solve
  G  C2 (128 x 64 rows), row shape                                    H  C2, quad shape       I  C2, quad shape + MFMA J J^T
  J  16 384 rows (64 x 256), K = 20, row    K  quad    L  quad + MFMA (the measured MFMA question)
  M / N  the bench launches of configs C3 / C5      O / P / Q  what each of 8 / 4 / 2 GPUs issues under strong scaling: 8 / 4 / 2 steps of
  its 128 / 256 / 512-seed shard in ONE launch (cppf_lm_batch_launch)      R  C2 as one plain launch of 8 192 rows
E, G, M, N, O, P, Q go through the batch entry point exactly as bench.py issues them (steps per launch = 262 144 // rows, at most 16).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs, make_inputs_problem  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "panda"
rb = get_robot(name)
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
dev = torch.device("cuda:0")
S, W = 1024, 256
LM = (1e-6, 3.5, 0.35)
x0, target = make_inputs(rb, S, W, dev, 0)
xo = torch.empty_like(x0)
pk = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev)
xp, tp, _ = make_inputs_problem(rb, S, W, dev, 0) if name == "panda" else (x0, target, "")
sm = torch.empty((S, 8), dtype=torch.float32, device=dev)
xc2, tc2, _ = make_inputs_problem(rb, 128, 64, dev, 0) if name == "panda" else (x0[: 128 * 64], target[:64], "")
xo2 = torch.empty_like(xc2)
x16 = xp[: 64 * 256].contiguous()
xo16 = torch.empty_like(x16)
ROW, QUAD = _hip.SHAPE_ROW, _hip.SHAPE_QUAD
mfma = lambda on: rb.debug_set("quad_mfma", on)  # noqa: E731

def batch_launch(robot, x, tg, S_, W_, B, collide=True):
    """what bench.py's Runner issues: B steps of one workload in ONE launch, each step with its own outputs"""
    n_ = S_ * W_
    items = []
    for _ in range(B):
        it = dict(x=x, target=tg, x_out=torch.empty_like(x))
        if collide:
            it.update(packed_out=torch.empty(robot.PACKED_BYTES_PER_ROW * n_, dtype=torch.uint8, device=dev),
                      summary_out=torch.empty((S_, 8), dtype=torch.float32, device=dev))
        else:
            it.update(errors_out=(torch.empty(n_, device=dev), torch.empty(n_, device=dev)))
        items.append(it)
    plan = robot.lm_batch_plan(items, *LM, n_steps=10)
    return plan.launch


bench_E = batch_launch(rb, xp, tp, S, W, 1)
bench_G = batch_launch(rb, xc2, tc2, 128, 64, 16, collide=False)
shard = {B: batch_launch(rb, xp[: (S // B) * W].contiguous(), tp, S // B, W, B) for B in (8, 4, 2)}

variants = [
    ("A: row, 262144 rows, K=10, no collision (random inputs)", lambda: rb.lm_pose_steps(x0, target, *LM, n_steps=10, x_out=xo, want_errors=True, shape=ROW)),
    ("B: row, 262144 rows, K=20, no collision (random inputs)", lambda: rb.lm_pose_steps(x0, target, *LM, n_steps=20, x_out=xo, want_errors=True, shape=ROW)),
    ("C: row, K=10 + collision (random inputs, no summary)", lambda: rb.lm_pose_steps(x0, target, *LM, n_steps=10, x_out=xo, packed_out=pk, shape=ROW)),
    ("D: collision_masks alone (random inputs)", lambda: rb.collision_masks(x0.reshape(S, W, -1))),
    ("E: the bench launch: row, K=10 + collision + per-seed summary, problem inputs (batch entry point, 1 step per launch)", bench_E),
    ("F: E with every row solved in double precision (CPPF_SOLVER_F64)", lambda: rb.lm_pose_steps(xp, tp, *LM, n_steps=10, x_out=xo, packed_out=pk, summary_out=sm, shape=ROW, solver=_hip.SOLVER_F64)),
    ("G: bench --config C2: 16 steps of 128 x 64 rows in one launch, K=10, no collision, row shape", bench_G),
    ("H: C2, quad shape (VALU J J^T)", lambda: (mfma(0), rb.lm_pose_steps(xc2, tc2, *LM, n_steps=10, x_out=xo2, shape=QUAD))),
    ("I: C2, quad shape, J J^T by v_mfma_f32_4x4x1", lambda: (mfma(1), rb.lm_pose_steps(xc2, tc2, *LM, n_steps=10, x_out=xo2, shape=QUAD), mfma(0))),
    ("J: 16384 rows, K=20, no collision, row shape", lambda: rb.lm_pose_steps(x16, tp, *LM, n_steps=20, x_out=xo16, shape=ROW)),
    ("K: 16384 rows, K=20, quad shape (VALU J J^T)", lambda: (mfma(0), rb.lm_pose_steps(x16, tp, *LM, n_steps=20, x_out=xo16, shape=QUAD))),
    ("L: 16384 rows, K=20, quad shape, J J^T by v_mfma_f32_4x4x1", lambda: (mfma(1), rb.lm_pose_steps(x16, tp, *LM, n_steps=20, x_out=xo16, shape=QUAD), mfma(0))),
]
# the bench launches of BASELINE configs C3 and C5 (bench.py --config C3 / C5): keys of profiles/r2_issue.json
if name == "panda":
    fetch = get_robot("fetch")
    fetch.set_obstacles([], [])
    fetch.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    x3, t3, _ = make_inputs_problem(fetch, 512, 256, dev, 0)
    xo3, pk3, sm3 = torch.empty_like(x3), torch.empty(fetch.PACKED_BYTES_PER_ROW * 512 * 256, dtype=torch.uint8, device=dev), torch.empty((512, 8), device=dev)
    variants.append(("M: bench --config C3: fetch 512 x 256, row, K=10 + self-collision + summary, problem inputs",
                     lambda: fetch.lm_pose_steps(x3, t3, *LM, n_steps=10, x_out=xo3, packed_out=pk3, summary_out=sm3, shape=ROW)))
    c12 = get_robot("chain12")
    c12.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    c12.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    x5, t5, _ = make_inputs_problem(c12, 4096, 512, dev, 0)
    xo5, pk5, sm5 = torch.empty_like(x5), torch.empty(c12.PACKED_BYTES_PER_ROW * 4096 * 512, dtype=torch.uint8, device=dev), torch.empty((4096, 8), device=dev)
    variants.append(("N: bench --config C5: chain12 4096 x 512, row, K=10 + collision + summary, problem inputs",
                     lambda: c12.lm_pose_steps(x5, t5, *LM, n_steps=10, x_out=xo5, packed_out=pk5, summary_out=sm5, shape=ROW)))
    bench_M = batch_launch(fetch, x3, t3, 512, 256, 2)
    bench_N = batch_launch(c12, x5, t5, 4096, 512, 1)
    variants[-2] = (variants[-2][0] + " (2 steps per launch)", bench_M)
    variants[-1] = (variants[-1][0] + " (batch entry point, 1 step per launch)", bench_N)
    variants.append(("O: strong-scaling shard of 8 GPUs: 8 steps of 128 x 256 rows in one launch, K=10 + collision + summary", shard[8]))
    variants.append(("P: strong-scaling shard of 4 GPUs: 4 steps of 256 x 256 rows in one launch", shard[4]))
    variants.append(("Q: strong-scaling shard of 2 GPUs: 2 steps of 512 x 256 rows in one launch", shard[2]))
    variants.append(("R: C2 = 128 x 64 rows as ONE plain launch, K=10, no collision, row shape", lambda: rb.lm_pose_steps(xc2, tc2, *LM, n_steps=10, x_out=xo2, shape=ROW)))
os.makedirs(os.path.join(ROOT, "gpurun_out", "rec"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "rec", "pmc_labels.json"), "w") as f:
    json.dump([v[0] for v in variants], f, indent=1)
torch.cuda.synchronize()
for rep in range(3):
    for _, fn in variants:
        fn()
torch.cuda.synchronize()
