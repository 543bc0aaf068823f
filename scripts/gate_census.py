#!/usr/bin/env python3
"""Census for the conditioning gate of the damped solve (developer tool): on bench.py's own inputs, step the pose-only LM one
iteration at a time through the public API (J and e of every row come back), evaluate the gate's a-posteriori estimate
    est = 6e-8 * max diag(A) * max |y| * a_max,   A = J J^T + lambda S^-2,  A y = e
in fp64 torch, and print per iteration the fraction of rows / wavefronts (64 consecutive rows) / workgroups (256 rows) that hold
a row with est > tau.  Usage: python scripts/gate_census.py [--inputs problem|random] [--robot panda] [--seeds 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--inputs", default="problem")
ap.add_argument("--robot", default="panda")
ap.add_argument("--seeds", type=int, default=1024)
ap.add_argument("--waypoints", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
robot = get_robot(args.robot)
S, W = args.seeds, args.waypoints
if args.inputs == "problem":
    x, target, _ = bench.make_inputs_problem(robot, S, W, dev, seed=0)
else:
    x, target = bench.make_inputs(robot, S, W, dev, seed=0)
lam, a_pos, a_rot = 1e-6, 3.5, 0.35
sc = torch.tensor([a_rot] * 3 + [a_pos] * 3, dtype=torch.float64, device=dev)
Lam = torch.diag(lam / sc**2)
taus = (1e-5, 3e-5, 1e-4)
print(f"# {args.robot} {S} x {W}, inputs = {args.inputs}; per iteration and tau: flagged rows / wavefronts / workgroups holding one")
for it in range(args.iters):
    r = robot.lm_pose_steps(x, target, lam, a_pos, a_rot, n_steps=1, clamp=True, return_residual=True, want_errors=True)
    Js, es = r["J"].double(), r["e"].double().reshape(-1, 6)  # scaled, as the reference returns them
    J = Js / sc[None, :, None]
    e = es / sc[None]
    A = J @ J.transpose(1, 2) + Lam[None]
    y = torch.linalg.solve(A, e[..., None])[..., 0]
    est = 6e-8 * torch.diagonal(A, dim1=1, dim2=2).amax(1) * y.abs().amax(1) * max(a_pos, a_rot)
    conv = float((r["pos_err_m"] < 1e-4).float().mean())
    line = f"it {it}: converged-after {conv:.3f} |"
    for tau in taus:
        f = est > tau
        fw = f.view(-1, 64).any(1).float().mean().item()
        fb = f.view(-1, 256).any(1).float().mean().item()
        nb = f.view(-1, 256).sum(1).float()
        line += f"  tau {tau:g}: rows {f.float().mean().item():.4f} waves {fw:.3f} blocks {fb:.3f} (max/blk {int(nb.max())})"
    print(line)
    x = r["x"]
