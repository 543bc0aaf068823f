#!/usr/bin/env python3
"""Coupled LM step: parallel cyclic reduction over the waypoints against the waypoint-after-waypoint elimination, by number
of trajectories (developer tool; the default switch-over in cppf_lm_full_step comes from this table)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from cppflow_amd import _hip
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
for name in ("panda", "fetch"):
    rb = get_robot(name)
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    for S in (1, 64, 128, 256, 512, 1024):
        x0, target = make_inputs(rb, S, 256, dev, 0)
        res = {}
        for mode, lim in (("pcr", 1 << 30), ("seq", 0)):
            rb.debug_set("pcr_max_rows", lim)
            for _ in range(3): rb.lm_full_step(x0, target, ALT_LOSS_V2_1_DIFF)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5): rb.lm_full_step(x0, target, ALT_LOSS_V2_1_DIFF)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 5 * 1e3)
            res[mode] = np.median(ts)
        print(f"{name} S={S:5d} W=256  pcr {res['pcr']:8.1f} us   sequential {res['seq']:8.1f} us")
