#!/usr/bin/env python3
"""Same-process A/B of the parallel-in-time elimination's forms (CPPF_TUNE_PCR_LDS: 0 workspace, 1 LDS, 2 LDS + split where the
library uses it by default (up to kPcrSplitMaxD joints), 3 LDS + split forced)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_problem
from cppflow_amd import _hip
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
L = _hip.lib()
for name in ("panda", "fetch"):
    rb = get_robot(name)
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    for S in (1, 64, 256, 512):
        x0, target, _ = make_inputs_problem(rb, S, 256, dev, 0)
        fn = lambda: rb.lm_full_step(x0, target, ALT_LOSS_V2_1_DIFF)
        rb.debug_set("pcr_max_rows", 1 << 30)
        res = {0: [], 1: [], 2: [], 3: []}
        for rnd in range(5):
            for mode in (0, 1, 2, 3):
                rb.debug_set("pcr_lds", mode)
                fn(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): fn()
                b.record(); torch.cuda.synchronize()
                res[mode].append(a.elapsed_time(b) / 10 * 1e3)
        rb.debug_set("pcr_lds", 2); rb.debug_set("pcr_max_rows", -1)
        print(f"{name:6s} S={S:4d} T=256: workspace {np.median(res[0]):7.1f}   LDS {np.median(res[1]):7.1f}   LDS + split (default rule) {np.median(res[2]):7.1f}   LDS + split (forced) {np.median(res[3]):7.1f} us", flush=True)
