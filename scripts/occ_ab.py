#!/usr/bin/env python3
"""A/B of the fused kernel's occupancy-capped instantiation in ONE process (developer tool): the same launch plans, alternating
`cppf_debug_set_occ_min_rows` between rounds, two streams, HIP events.   python scripts/occ_ab.py [robot seeds waypoints]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_problem  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

cases = [("panda", 1024, 256), ("fetch", 512, 256), ("chain12", 4096, 512), ("panda", 512, 256)]
if len(sys.argv) > 3:
    cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
dev = torch.device("cuda:0")
for name, S, W in cases:
    rb = get_robot(name)
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES) if name != "fetch" else []
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    x0, target, _ = make_inputs_problem(rb, S, W, dev, 0)
    n = x0.shape[0]
    plans = []
    for _ in range(4):
        xo = torch.empty_like(x0)
        pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=dev)
        sm = torch.empty((S, 8), device=dev)
        plans.append(rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk, summary_out=sm))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    res = {0: [], 1: []}
    steps = 400 if n <= 262144 else 60
    for rnd in range(6):
        for mode in (0, 1):
            _hip.lib().cppf_debug_set_occ_min_rows(-1 if mode else (1 << 30))
            for i in range(steps // 4):
                plans[i % 4].launch_on(streams[i % 2])
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            for i in range(steps):
                plans[i % 4].launch_on(streams[i % 2])
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            b.record()
            torch.cuda.synchronize()
            res[mode].append(a.elapsed_time(b) / steps * 1e3)
    _hip.lib().cppf_debug_set_occ_min_rows(-1)
    print(f"{name:8s} {S} x {W}: default build {np.median(res[0]):8.2f} us/step   occupancy-capped build {np.median(res[1]):8.2f} us/step", flush=True)
