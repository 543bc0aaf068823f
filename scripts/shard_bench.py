#!/usr/bin/env python3
"""Isolated-launch latency of the fused kernel at the batch sizes that cannot fill the chip (developer tool): the C2
configuration (Panda 128 x 64, FK+Jacobian+LM only), one eighth / quarter / half of C4 (a strong-scaling shard at 8 / 4 / 2
GPUs: K = 10 + collision + summary) and full C4, for every kernel shape.  HIP events around single launches after a pre-warm,
medians; also the two-streams-in-flight step rate.

    python scripts/shard_bench.py [--shapes row,quad] [--lib path/to/libcppflow_hip.so]
"""
import argparse
import os
import sys
import time

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="row,quad")
ap.add_argument("--lib", default=None)
ap.add_argument("--robot", default="panda")
ap.add_argument("--mfma", type=int, default=0, help="quad shape: J J^T by v_mfma_f32_4x4x1 (CPPF_TUNE_QUAD_MFMA)")
ap.add_argument("--sizes", default="128,256,512,1024", help="seeds (x 256 waypoints) of the collision-fused cases")
args = ap.parse_args()
if args.lib:
    os.environ["CPPFLOW_HIP_LIB"] = os.path.abspath(args.lib)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

SHAPES = {"row": _hip.SHAPE_ROW, "quad": _hip.SHAPE_QUAD, "auto": _hip.SHAPE_AUTO}


def median_us(fn, reps=300):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def step_rate_us(plans, streams, steps=2000):
    for i in range(200):
        plans[i % len(plans)].launch_on(streams[i % len(streams)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        plans[i % len(plans)].launch_on(streams[i % len(streams)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    dev = torch.device("cuda:0")
    rb = get_robot(args.robot)
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    W = 256
    x_all, target, _ = bench.make_inputs_problem(rb, 1024, W, dev, 0)
    x_c2, t_c2, _ = bench.make_inputs_problem(rb, 128, 64, dev, 0) if args.robot == "panda" else (x_all[: 128 * 64], target[:64], "")
    t0 = time.perf_counter()
    warm = rb.lm_launch_plan(x_all, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=torch.empty_like(x_all))
    while time.perf_counter() - t0 < 0.1:
        for _ in range(50):
            warm.launch()
        torch.cuda.synchronize()
    rb.debug_set("quad_mfma", args.mfma)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for shape in args.shapes.split(","):
        sh = SHAPES[shape]
        xo = torch.empty_like(x_c2)
        t = median_us(lambda: rb.lm_pose_steps(x_c2, t_c2, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, shape=sh))
        p = rb.lm_launch_plan(x_c2, t_c2, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, shape=sh)
        tp = median_us(p.launch)
        print(f"{shape:5s} C2  {args.robot} 128 x 64 = 8192 rows, K=10, no collision     isolated {tp:7.1f} us (python API {t:7.1f})")
        for S in [int(v) for v in args.sizes.split(",")]:
            x0 = x_all[: S * W].contiguous()
            plans = []
            for _ in range(4):
                plans.append(rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=torch.empty_like(x0),
                                               packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev),
                                               summary_out=torch.empty((S, 8), device=dev), shape=sh))
            t_iso = median_us(plans[0].launch)
            t_2 = step_rate_us(plans, streams)
            pk1 = rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=1, x_out=torch.empty_like(x0),
                                    packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev), shape=sh)
            t_k1 = median_us(pk1.launch)
            tag = shape + ("+mfma" if args.mfma and shape == "quad" else "")
            print(f"{tag:9s} C4/{1024 // S:<3d} {args.robot} {S:4d} x 256 = {S * W:6d} rows, K=10 + coll + summary  isolated {t_iso:7.1f} us   "
                  f"2 streams {t_2:7.1f} us/step   K=1 + coll {t_k1:6.1f} us")


if __name__ == "__main__":
    main()
