#!/usr/bin/env python3
"""Developer measurement: CPPF_TUNE_LM_PACE sweep on the product library (C4 planner inputs, Panda): one stream back to back and two
streams alternating, plain launches, for explicit schedules (ticks of 10 ns per LM iteration), the built-in estimate (-1) and off (0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cppflow_amd import _hip
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "panda"
rb = get_robot(name)
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
S, W, K = 1024, 256, 10
x0, target, _ = bench.make_inputs_problem(rb, S, W, DEV, seed=0) if name == "panda" else (*bench.make_inputs(rb, S, W, DEV, 0), "")
n = S * W
bufs = [(torch.empty_like(x0), torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV), torch.empty((S, 8), dtype=torch.float32, device=DEV)) for _ in range(2)]
streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
def launch(i=0):
    xo, pk, sm = bufs[i]
    rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=K, x_out=xo, packed_out=pk, summary_out=sm, shape=_hip.SHAPE_ROW)
ev = lambda: torch.cuda.Event(enable_timing=True)
for _ in range(2000):
    launch()
torch.cuda.synchronize()
for pace in [0, -1, 250, 300, 350, 400, 450, 500, 0]:
    rb.debug_set("lm_pace", pace)
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    one = []
    for _ in range(3):
        a, b = ev(), ev(); a.record()
        for _ in range(400):
            launch()
        b.record(); torch.cuda.synchronize(); one.append(a.elapsed_time(b) * 1e3 / 400)
    two = []
    for _ in range(3):
        torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        for i in range(1000):
            with torch.cuda.stream(streams[i & 1]):
                launch(i & 1)
        for s in streams: torch.cuda.current_stream().wait_stream(s)
        b.record(); torch.cuda.synchronize(); two.append(a.elapsed_time(b) * 1e3 / 1000)
    print(f"{name} lm_pace {pace:5d}: one stream {np.median(one):6.2f} us per launch   two streams {np.median(two):6.2f}", flush=True)
rb.debug_set("lm_pace")
