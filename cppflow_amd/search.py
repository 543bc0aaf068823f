"""Search-side consumers of the hot path (`cppflow/search.py`): the joint-limit margin mask (`:25-52`), the external cost
`100*jlim + 1000*env + 1000*self` (`:14-15, 146-150`) and `dp_search` (`:128-191`).

The mask and the cost come out of the collision kernel (one launch for all three masks and the cost).  `dp_search`
itself is SURVEY.md 8(f) item 2 ("next"): it is kept here as the host-side torch recurrence the reference runs, working
on the device the candidates live on instead of forcing them to the CPU (`search.py:140-141`).
"""

import math
from typing import Optional, Tuple

import numpy as np
import torch

K_JLIM_COST = 100
K_COLLISION_COST = 1000
DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE = float(np.deg2rad(1.5))
DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC = 0.03


def joint_limit_almost_violations_3d(
    robot,
    qs: torch.Tensor,
    eps_revolute: float = DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE,
    eps_prismatic: float = DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC,
) -> torch.Tensor:
    """[k, T] float32, 1 where any joint is within eps of a limit (cppflow/search.py:25-52)."""
    assert len(qs.shape) == 3
    prev = robot.padded_joint_limits()
    robot.set_joint_limit_padding(eps_revolute, eps_prismatic)
    try:
        out = robot.collision_masks(qs, only=("jlim",))["jlim_mask"]
    finally:
        if prev is None:
            robot.set_joint_limit_padding(None, None)
        else:
            robot._jl_padding = prev
            for h in robot._handles.values():
                robot._apply_jl_padding(h)
    return out.type(torch.float32)


def q_costs_external(robot, q: torch.Tensor, problem=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(cost [k,T] float32, jlim, env, self masks) in ONE launch, with the default paddings of search.py:20-21."""
    if problem is not None:
        problem.bind_obstacles()
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    r = robot.collision_masks(q)
    return r["ext_cost"], r["jlim_mask"], r["env_mask"], r["self_mask"]


def _get_mjacs(q: torch.Tensor, robot, prismatic_joint_scaling: float = 5.0) -> torch.Tensor:
    """[k, k, T-1]: max wrapped joint change from path a at t to path b at t+1 (cppflow/search.py:100-125)."""
    dqs = q[:, 1:, :].unsqueeze(1) - q[:, :-1, :].unsqueeze(0)
    if robot.has_prismatic_joints:
        dqs = dqs.clone()
        dqs[:, :, :, robot.prismatic_joint_idxs] *= prismatic_joint_scaling
    return torch.abs(torch.remainder(dqs + math.pi, 2 * math.pi) - math.pi).amax(dim=3)


def dp_search(
    robot,
    q: torch.Tensor,
    self_collision_violations: torch.Tensor,
    env_collision_violations: torch.Tensor,
    use_cuda: bool = True,
    verbosity: int = 0,
    q_costs: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """min-max dynamic programme over k candidate paths (cppflow/search.py:128-191) -> best path [T, d]."""
    k, T, d = q.shape
    if q_costs is None:
        jlim = joint_limit_almost_violations_3d(robot, q)
        q_costs = (
            K_JLIM_COST * jlim
            + K_COLLISION_COST * env_collision_violations.to(q.device, torch.float32)
            + K_COLLISION_COST * self_collision_violations.to(q.device, torch.float32)
        )
    costs = torch.zeros((k, T), device=q.device, dtype=q.dtype)
    costs[:, 0] = q_costs[:, 0]
    mjacs = _get_mjacs(q, robot)
    memo = torch.zeros((k, T), dtype=torch.long, device=q.device)
    for t in range(1, T):
        # entry [b, a]: arrive at candidate b from candidate a
        nxt = torch.maximum(mjacs[:, :, t - 1], costs[:, t - 1].unsqueeze(0)) + q_costs[:, t].unsqueeze(1)
        costs[:, t], memo[:, t] = torch.min(nxt, dim=1)
    best = torch.zeros((T, d), dtype=q.dtype, device=q.device)
    i = torch.argmin(costs[:, -1])
    for t in range(T - 1, -1, -1):
        best[t] = q[i, t]
        i = memo[i, t]
    return best
