"""Obstacle sets of the reference's Panda problems, restated as data (`cppflow/problems/panda__2cubes.yaml:10-28`,
`panda__1cube.yaml:10-22`): (x, y, z, size_x, size_y, size_z), axis-aligned, and the conversion to the
(cuboid[6], Tcuboid[4,4]) pair of `cppflow/data_type_utils.py:109-124`; and the synthetic (seeds x waypoints) workloads of SURVEY.md
8(d) that `bench.py`, the GPU tests and the measurement scripts share (`make_inputs_problem`: the named reference problem's target
path + per-seed IK branches; `make_inputs`: the fall-back set, independent random configurations per waypoint)."""

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

# the target paths of the problems BASELINE.json's configs name, as arrays (tests/golden/make_golden.py wrote them from the
# reference's own csv / yaml data files)
REFERENCE_PATHS_NPZ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_paths.npz")

PANDA_2CUBES_OBSTACLES = [(0.2, 0.3, 0.4, 0.15, 0.15, 0.15), (-0.25, 0.3, 0.75, 0.15, 0.15, 0.15)]
PANDA_1CUBE_OBSTACLES = [(0.0, 0.2, 0.7, 0.25, 0.25, 0.25)]


def obstacle_arrays(obstacles: Sequence[Tuple[float, ...]]) -> List[Tuple[np.ndarray, np.ndarray]]:
    out = []
    for x, y, z, sx, sy, sz in obstacles:
        cuboid = np.array([-sx / 2, -sy / 2, -sz / 2, sx / 2, sy / 2, sz / 2], dtype=np.float32)
        T = np.zeros((4, 4), dtype=np.float32)  # element [3,3] stays 0, as in the reference loader
        T[:3, :3] = np.eye(3, dtype=np.float32)
        T[:3, 3] = (x, y, z)
        out.append((cuboid, T))
    return out


def make_inputs(robot, S, W, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo = torch.tensor([l for l, _ in robot.actuated_joints_limits], dtype=torch.float32)
    hi = torch.tensor([u for _, u in robot.actuated_joints_limits], dtype=torch.float32)
    q_star = lo + (hi - lo) * torch.rand((W, robot.ndof), generator=g)
    target = robot.forward_kinematics(q_star.to(device))  # [W,7]
    g2 = torch.Generator(device="cpu").manual_seed(1000 + seed)
    x0 = q_star[None] + 0.1 * torch.randn((S, W, robot.ndof), generator=g2)
    x0 = torch.minimum(torch.maximum(x0, lo), hi).reshape(S * W, robot.ndof).contiguous()
    return x0.to(device), target.contiguous()


PROBLEM_PATHS = {  # tests/golden/reference_paths.npz: the target paths of the problems BASELINE.json's configs name
    ("panda", 64): "panda__1cube_first64",
    ("fetch", 256): "fetch__hello_first256",
    ("panda", 256): "panda__2cubes_resampled256",
}


def make_inputs_problem(robot, S, W, device, seed):
    """SURVEY.md 8(d) inputs: the target path of the reference problem the configuration names (committed fixture; the 12-DoF
    chain has no reference problem: target = FK of a smooth random walk q*_{t+1} = clamp(q*_t + 0.02 randn)) and, per seed, a
    distinct IK branch q*_s that tracks the path (waypoint 0 solved by damped LM from a U(limits) start, every later waypoint
    warm-started from its predecessor, a branch that loses the path continuing on one that did not -- what IKFlow + dp_search
    hand to the optimiser), then x0 = clamp(q*_s + 0.1 randn)
    (the construction of the reference's tests/optimization_test.py:82).  Returns (x0 [S*W,d], target [W,7], description)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo = torch.tensor([l for l, _ in robot.actuated_joints_limits], dtype=torch.float32)
    hi = torch.tensor([u for _, u in robot.actuated_joints_limits], dtype=torch.float32)
    d = robot.ndof
    key = PROBLEM_PATHS.get((robot.name, W))
    if key is not None:
        z = np.load(REFERENCE_PATHS_NPZ)
        target = torch.tensor(z[key], dtype=torch.float32, device=device).contiguous()
        what = f"target path = {key} (reference problem, tests/golden/reference_paths.npz)"
    else:
        q = torch.empty((W, d), dtype=torch.float32)
        q[0] = lo + (hi - lo) * torch.rand(d, generator=g)
        steps = 0.02 * torch.randn((W, d), generator=g)
        for t in range(1, W):
            q[t] = torch.minimum(torch.maximum(q[t - 1] + steps[t], lo), hi)
        target = robot.forward_kinematics(q.to(device)).contiguous()
        what = "target path = FK of a smooth random walk (q*_{t+1} = clamp(q*_t + 0.02 randn))"
    lo_d, hi_d = lo.to(device), hi.to(device)
    branch = torch.empty((S, W, d), dtype=torch.float32, device=device)
    # Fetch: the lift joint is a pure z translation at the root of the chain (torso_lift_link is unrotated w.r.t. the world,
    # cppflow/data_type_utils.py:65-73), so a seed is a lift height -- drawn once per seed from the middle 80 % of its range, as a
    # sampler of whole-body configurations would -- and an ARM branch tracking the path lowered by that height (the 7-joint chain of
    # fetch_arm).  Tracking the path with the pose-only LM step on all 8 joints instead lets the lift joint, whose Jacobian column
    # is a whole metre per unit, take every vertical motion: the branches drift onto its limits, the clamp of
    # cppflow/optimization.py:259 pins them there, and half the rows of a batch built that way can no longer converge (round 3's
    # C3 inputs: 55 %) -- a property of those inputs, not of any kernel.
    ik_robot, lift = robot, None
    if robot.name == "fetch":
        from cppflow_amd.robots import get_robot as _get_robot

        ik_robot = _get_robot("fetch_arm")
        lift = (lo[0] + (hi[0] - lo[0]) * (0.1 + 0.8 * torch.rand(S, generator=g))).to(device)
    d_ik = ik_robot.ndof
    lo_ik = torch.tensor([l for l, _ in ik_robot.actuated_joints_limits], dtype=torch.float32)
    hi_ik = torch.tensor([u for _, u in ik_robot.actuated_joints_limits], dtype=torch.float32)

    def seeds_target(w):
        """[S, 7]: waypoint w as every seed's IK problem sees it (row r of a launch with W = n uses target row r)"""
        t = target[w : w + 1].repeat(S, 1)
        if lift is not None:
            t[:, 2] -= lift
        return t.contiguous()

    # waypoint 0: damped LM from random starts, re-drawing the seeds that did not reach the pose (up to 12 rounds)
    x = torch.empty((S, d_ik), dtype=torch.float32, device=device)
    todo = torch.ones(S, dtype=torch.bool, device=device)
    t0 = seeds_target(0)
    for _ in range(12):
        start = (lo_ik + (hi_ik - lo_ik) * (0.1 + 0.8 * torch.rand((S, d_ik), generator=g))).to(device).contiguous()
        r = ik_robot.lm_pose_steps(start, t0, 1e-2, 3.5, 0.35, n_steps=60)
        r = ik_robot.lm_pose_steps(r["x"], t0, 1e-6, 3.5, 0.35, n_steps=10, want_errors=True)
        ok = (r["pos_err_m"] < 1e-4) & (r["rot_err_rad"] < 1.75e-3)
        take = todo & ok
        x[take] = r["x"][take]
        todo &= ~ok
        if not bool(todo.any()):
            break
    x[todo] = r["x"][todo]
    x = x.contiguous()
    gd = torch.Generator(device=device).manual_seed(seed + 17)
    for w in range(W):
        r = ik_robot.lm_pose_steps(x, seeds_target(w), 1e-6, 3.5, 0.35, n_steps=8, want_errors=True)
        x = r["x"]
        # a branch that loses the path (runs into a joint limit) continues on a branch that did not (with the donor's lift height)
        ok = (r["pos_err_m"] < 1e-4) & (r["rot_err_rad"] < 1.75e-3)
        donors = torch.nonzero(ok).reshape(-1)
        if 0 < donors.numel() < S:
            pick = donors[torch.randint(donors.numel(), (S,), generator=gd, device=device)]
            x = torch.where(ok[:, None], x, x[pick]).contiguous()
            if lift is not None:
                lift = torch.where(ok, lift, lift[pick]).contiguous()
        branch[:, w] = x if lift is None else torch.cat([lift[:, None], x], dim=1)
    noise = 0.1 * torch.randn((S, W, d), generator=g)
    x0 = torch.minimum(torch.maximum(branch + noise.to(device), lo_d), hi_d).reshape(S * W, d).contiguous()
    return x0, target, what + "; seeds = per-seed IK branch tracking the path + 0.1 randn"
