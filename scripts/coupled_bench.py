#!/usr/bin/env python3
"""Coupled LM step (cppf_lm_full_step) timings on one MI355X (developer tool): S trajectories x T waypoints under the three
elimination kernels -- parallel cyclic reduction (pcr), eight trajectories per wavefront with one block row per lane (rows),
one wavefront per trajectory (wave) -- and the largest difference between their steps.
Usage: python scripts/coupled_bench.py [--robots panda,fetch] [--waypoints 256] [--seeds 1,8,64,256,512,1024,4096]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs, make_inputs_problem  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402


def timed(fn, reps, rounds):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", default="panda,fetch")
    ap.add_argument("--waypoints", type=int, default=256)
    ap.add_argument("--seeds", default="1,8,64,256,512,1024,4096")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inputs", default="problem", choices=["problem", "random"],
                    help="problem: IK branches tracking the reference problem's target path + 0.1 rad noise (bench.py's inputs); "
                         "random: an independent uniform configuration per waypoint (no coherence along the path, many collisions)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _hip.lib()
    for name in args.robots.split(","):
        rb = get_robot(name)
        obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
        rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
        for S in [int(s) for s in args.seeds.split(",")]:
            if args.inputs == "problem":
                x0, target, _ = make_inputs_problem(rb, S, args.waypoints, dev, 0)
            else:
                x0, target = make_inputs(rb, S, args.waypoints, dev, 0)
            m = rb.collision_masks(x0.reshape(S, args.waypoints, -1))
            hit = float((m["self_mask"] | m["env_mask"]).float().mean())
            fn = lambda: rb.lm_full_step(x0, target, ALT_LOSS_V2_1_DIFF)  # noqa: E731
            res, outs = {}, {}
            for mode, (pcr, rows) in {"pcr": (1 << 30, 1), "rows": (0, 1), "wave": (0, 0)}.items():
                rb.debug_set("pcr_max_rows", pcr)
                rb.debug_set("full_rows", rows)
                outs[mode] = fn().clone()
                res[mode] = timed(fn, args.reps, args.rounds)
            rb.debug_set("pcr_max_rows", -1)
            rb.debug_set("full_rows", 1)
            auto = timed(fn, args.reps, args.rounds)
            step = float((outs["wave"] - x0).abs().max())
            d_rows = float((outs["rows"] - outs["wave"]).abs().max())
            d_pcr = float((outs["pcr"] - outs["wave"]).abs().max())
            print(f"{name:6s} S={S:5d} T={args.waypoints}: pcr {res['pcr']:8.1f}  rows {res['rows']:8.1f}  wave {res['wave']:8.1f}  "
                  f"default {auto:8.1f} us   colliding rows {hit:.3f}  |step| {step:.3g}  rows-wave {d_rows:.2g}  pcr-wave {d_pcr:.2g}", flush=True)


if __name__ == "__main__":
    main()
