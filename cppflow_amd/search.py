"""Search-side consumers of the hot path (`cppflow/search.py`): the joint-limit margin mask (`:25-52`), the external cost
`100*jlim + 1000*env + 1000*self` (`:14-15, 146-150`) and `dp_search` (`:128-191`).

The mask and the cost come out of the collision kernel (one launch for all three masks and the cost).  `dp_search` runs on
the device the candidates live on (`cppf_dp_search`; the reference forces them to the CPU, `search.py:140-141`) and never
materialises the `[k,k,T-1]` mjac tensor; `_get_mjacs` still produces that tensor for callers that want it.
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
        robot.set_padded_joint_limits(prev)
    return out.type(torch.float32)


def q_costs_external(robot, q: torch.Tensor, problem=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(cost [k,T] float32, jlim, env, self masks) in ONE launch, with the default paddings of search.py:20-21."""
    if problem is not None:
        problem.bind_obstacles()
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    r = robot.collision_masks(q)
    return r["ext_cost"], r["jlim_mask"], r["env_mask"], r["self_mask"]


def dp_search(
    robot,
    q: torch.Tensor,
    self_collision_violations: Optional[torch.Tensor],
    env_collision_violations: Optional[torch.Tensor],
    use_cuda: bool = True,
    verbosity: int = 0,
    q_costs: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """min-max dynamic programme over k candidate paths (cppflow/search.py:128-191) -> best path [T, d], on the device
    (`cppf_dp_search`: one resident launch up to 1024 candidates; the reference moves q to the CPU, `search.py:140-141`).

    The external cost is `100*jlim + 1000*env + 1000*self` (`:146-150`): pass the two violation masks like the reference
    does, or `q_costs` directly (the `ext_cost` output of the collision / fused launch)."""
    k, T, d = q.shape
    if q_costs is None:
        jlim = joint_limit_almost_violations_3d(robot, q)
        q_costs = (
            K_JLIM_COST * jlim
            + K_COLLISION_COST * env_collision_violations.to(q.device, torch.float32)
            + K_COLLISION_COST * self_collision_violations.to(q.device, torch.float32)
        )
    q_costs = q_costs.contiguous()
    best_path, best_idx, _, ran = robot.dp_search(q, q_costs, return_method=True)
    # Only the single resident launch can fail this way (its bounded waits expire on a CU-masked / partitioned device: it then
    # reports best_idx = -1, include/cppflow_hip.h: cppf_dp_search); the table and per-waypoint forms cannot, so they are not asked
    # (no host synchronisation on their account).  The fall-back is an argument of the repeated CALL -- one launch per waypoint --
    # not a switch on the robot handle: another thread's searches on the same robot are not affected, and no setting is lost.
    if ran == "resident" and int(best_idx[0].item()) < 0:
        best_path, best_idx, _ = robot.dp_search(q, q_costs, method="launches")
    return best_path


def _get_mjacs(q: torch.Tensor, robot, prismatic_joint_scaling: float = 5.0) -> torch.Tensor:
    """[k, k, T-1] maximum joint changes between every pair of candidates at consecutive timesteps
    (cppflow/search.py:100-125)."""
    return robot.mjacs(q, prismatic_joint_scaling)


def dp_search_slow(problem, qpaths, use_cuda: bool = True, verbosity: int = 1) -> torch.Tensor:
    """The reference keeps a doubly-nested Python-loop version of the dynamic programme beside the vectorised one
    (cppflow/search.py:55-97: same costs, same first-minimal-index rule).  Both are the same recurrence, so here both
    names run the device kernel; this entry takes the reference's arguments (a list of [T, d] paths and the problem) and
    evaluates the masks itself."""
    q = torch.stack(list(qpaths)).detach().contiguous()
    cost, _, _, _ = q_costs_external(problem.robot, q, problem)
    return dp_search(problem.robot, q, None, None, use_cuda=use_cuda, verbosity=verbosity, q_costs=cost)
