#!/usr/bin/env python3
"""How the fused launch scales with the number of workgroups on one MI355X (developer tool): duration of isolated launches for
S = 256 ... 2048 seeds x 256 waypoints (S workgroups of 256 rows on 256 CUs), after a pre-warm, with HIP events.

Reads as  T(S) ~ latency of one wavefront (one row set per SIMD)  +  (wavefronts per SIMD - 1) x throughput cost per wavefront:
the marginal wavefront runs at the chip's issue ceiling, the first one cannot (a single wavefront issues at most one VALU
instruction per ~5.5 cycles), which is why two independent batches in flight (bench.py --streams 2) reach the ceiling and an
isolated 1024-workgroup launch does not."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_problem  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402


def main():
    rb = get_robot("panda")
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    dev, W = torch.device("cuda:0"), 256
    x_all, target, _ = make_inputs_problem(rb, 2048, W, dev, 0)
    plans = {}
    for S in (256, 512, 768, 1024, 1280, 1536, 2048):
        x0 = x_all[: S * W].contiguous()
        xo = torch.empty_like(x0)
        pk = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev)
        sm = torch.empty((S, 8), device=dev)
        plans[S] = rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=xo, packed_out=pk, summary_out=sm)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        for _ in range(50):
            plans[1024].launch()
        torch.cuda.synchronize()
    prev = None
    for S, p in plans.items():
        for _ in range(200):
            p.launch()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
        for a, b in ev:
            a.record()
            p.launch()
            b.record()
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
        extra = "" if prev is None else f"   +{(t - prev[1]) / ((S - prev[0]) / 256):5.1f} us per extra workgroup per CU"
        print(f"S = {S:5d} seeds = {S / 256:4.1f} workgroups per CU   isolated launch {t:6.1f} us{extra}")
        prev = (S, t)


if __name__ == "__main__":
    main()
