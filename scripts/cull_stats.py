#!/usr/bin/env python3
"""Developer measurement (GPU box): how many exact capsule tests does a ROW need, and how many does its WAVEFRONT execute?
The fused kernel's broad phase is wave-uniform (an exact test runs when ANY of the 64 rows of the wavefront is within reach), so
incoherent rows (the independent random configurations of `--inputs random`) execute tests for their neighbours.  Prints, for the C4
planner inputs after K = 10 and for random configurations: exact tests needed per row (per-lane broad phase) and executed per row
(per-wavefront broad phase), self pairs and (capsule, cuboid) items apart.  Capsule end points from the CPU oracle (checker use)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tests import helpers as H
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot

name = sys.argv[1] if len(sys.argv) > 1 else "panda"
ch, o = H.chain(name), H.oracle32(name)
r = ch.cap_r.astype(np.float64)
half = np.linalg.norm(ch.cap_p1 - ch.cap_p0, axis=-1) / 2


def stats(x, label):
    ep = np.asarray(o.capsule_endpoints(x))
    c = 0.5 * (ep[..., :3] + ep[..., 3:])
    tl = tw = 0.0
    for a, b in ch.pairs:
        near = ((c[:, a] - c[:, b]) ** 2).sum(-1) <= (half[a] + half[b] + r[a] + r[b] + 0.01) ** 2
        tl += near.mean(); tw += near.reshape(-1, 64).any(1).mean()
    el = ew = 0.0
    for (x_, y_, z_, sx, sy, sz) in PANDA_2CUBES_OBSTACLES:
        lo, hi = np.array([x_, y_, z_]) - np.array([sx, sy, sz]) / 2, np.array([x_, y_, z_]) + np.array([sx, sy, sz]) / 2
        for k in range(c.shape[1]):
            e = c[:, k] - np.clip(c[:, k], lo, hi)
            near = (e ** 2).sum(-1) <= (half[k] + r[k] + 0.01) ** 2
            el += near.mean(); ew += near.reshape(-1, 64).any(1).mean()
    print(f"{name} {label}: self pairs {len(ch.pairs)}: needed per row {tl:.2f}, executed per row {tw:.2f};  (capsule, cuboid) items {c.shape[1] * 2}: needed {el:.2f}, executed {ew:.2f}", flush=True)


dev = torch.device("cuda:0")
rb = get_robot(name)
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
S, W = 128, 256
for label, (x0, target) in (("planner inputs", bench.make_inputs_problem(rb, S, W, dev, seed=0)[:2]), ("random configurations", bench.make_inputs(rb, S, W, dev, 1))):
    out = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=10)
    stats(out["x"].cpu().numpy(), label + ", x after K = 10")
