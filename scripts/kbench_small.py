#!/usr/bin/env python3
"""Latency of the planner-cadence calls on one MI355X (developer tool): single-trajectory coupled step, small-k dp_search,
K = 1 fused step -- the sizes `CppFlowPlanner` issues once the search has picked one path.  HIP events, medians."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for name in ("panda", "fetch"):
        rb = get_robot(name)
        obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
        rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
        rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
        cases = []
        for S, W in ((1, 64), (1, 256), (1, 1024), (8, 256), (64, 256), (1024, 256)):
            x0, target = make_inputs(rb, S, W, dev, 0)
            cases.append((f"{name} lm_full_step S={S} W={W}", lambda x0=x0, t=target: rb.lm_full_step(x0, t, ALT_LOSS_V2_1_DIFF)))
            cases.append((f"{name} lm K=1+coll  S={S} W={W}", lambda x0=x0, t=target: rb.lm_pose_steps(x0, t, 1e-6, 3.5, 0.35, n_steps=1, want_errors=True, want_collisions=True)))
        for k, T in ((16, 256), (64, 256), (175, 64), (175, 256), (175, 1024)):
            q = torch.rand((k, T, rb.ndof), device=dev)
            ext = torch.zeros((k, T), device=dev)
            cases.append((f"{name} dp_search k={k} T={T}", lambda q=q, e=ext: rb.dp_search(q, e)))
        for _, fn in cases:
            fn()
        torch.cuda.synchronize()
        times = {c[0]: [] for c in cases}
        for _ in range(5):
            for nm, fn in cases:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    fn()
                b.record()
                torch.cuda.synchronize()
                times[nm].append(a.elapsed_time(b) / 5 * 1e3)
        for nm, _ in cases:
            print(f"{nm:40s} median {np.median(times[nm]):9.1f} us")


if __name__ == "__main__":
    main()
