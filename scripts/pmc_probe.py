#!/usr/bin/env python3
"""Launch a few labelled variants once each (after warm-up) so that a rocprofv3 --pmc pass attributes counters per dispatch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs, make_inputs_problem
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
name = sys.argv[1] if len(sys.argv) > 1 else "panda"
rb = get_robot(name)
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
dev = torch.device("cuda:0")
S, W = 1024, 256
x0, target = make_inputs(rb, S, W, dev, 0)
xo = torch.empty_like(x0)
pk = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev)
xp, tp, _ = make_inputs_problem(rb, S, W, dev, 0) if name == "panda" else (x0, target, "")
sm = torch.empty((S, 8), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, want_errors=True)      # dispatch A: K=10 no coll
    rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=20, x_out=xo, want_errors=True)      # dispatch B: K=20 no coll
    rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk)         # dispatch C: K=10 coll
    rb.collision_masks(x0.reshape(S, W, -1))                                                    # dispatch D: collision only
    rb.lm_pose_steps(xp, tp, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk, summary_out=sm)  # dispatch E: the bench launch (problem inputs, + summary)
torch.cuda.synchronize()
