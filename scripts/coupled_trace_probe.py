#!/usr/bin/env python3
"""Developer probe for `rocprofv3 --kernel-trace`: 200 coupled LM steps (cppf_lm_full_step) of ONE trajectory of 256 waypoints, Fetch then Panda,
so that the trace shows the kernels of a step, their durations and the gaps between them."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_problem
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["fetch", "panda"]):
    rb = get_robot(name)
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    x0, target, _ = make_inputs_problem(rb, 1, 256, dev, 0)
    for _ in range(200):
        rb.lm_full_step(x0, target, ALT_LOSS_V2_1_DIFF)
    torch.cuda.synchronize()
