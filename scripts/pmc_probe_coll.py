#!/usr/bin/env python3
"""Two labelled launches for a rocprofv3 --pmc pass: the mask-only fused launch (Panda, 1024 x 256, K = 10, packed outputs) on the planner
inputs and on independent random configurations; three rounds, the last one counts."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs, make_inputs_problem
from cppflow_amd import _hip
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
rb = get_robot("panda")
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
dev = torch.device("cuda:0")
S, W = 1024, 256
xr, tr = make_inputs(rb, S, W, dev, 0)
xp, tp, _ = make_inputs_problem(rb, S, W, dev, 0)
xo = torch.empty_like(xr)
pk = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev)
for _ in range(3):
    rb.lm_pose_steps(xp, tp, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk, shape=_hip.SHAPE_ROW)
    torch.cuda.synchronize()
    rb.lm_pose_steps(xr, tr, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk, shape=_hip.SHAPE_ROW)
    torch.cuda.synchronize()
