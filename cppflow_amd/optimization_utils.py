"""Hot-path helpers with the reference's names and argument meaning (`cppflow/optimization_utils.py`):
`get_6d_pose_errors` (`:802-820`), `clamp_to_joint_limits` (`:823-833`), `x_is_valid` (`:836-923`), and the two row-mask
helpers (`:31-60`).  Each compute call is one kernel launch through the C ABI.
"""

from typing import Optional, Tuple

import torch

from cppflow_amd.config import ENV_COLLISIONS_IGNORED, SELF_COLLISIONS_IGNORED
from cppflow_amd.evaluation_utils import seed_metrics_are_below_threshold
from cppflow_amd.lm_hyper_parameters import OptimizationParameters  # noqa: F401  (re-exported like the reference)
from cppflow_amd.utils import make_text_green_or_red


def _get_prismatic_and_revolute_row_mask(robot, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Boolean masks over the n = n_qs * ndof rows of a stacked residual: (revolute rows, prismatic rows)."""
    assert n % robot.ndof == 0, f"error - n {n} is not divisible by ndof {robot.ndof}"
    rev = torch.zeros(robot.ndof, dtype=torch.bool)
    rev[robot.revolute_joint_idxs] = True
    rev = rev.tile(n // robot.ndof)
    return rev, torch.logical_not(rev)


def _get_rotation_and_position_row_mask(n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Masks over the 6n rows of a stacked pose residual laid out [rot3, pos3] per config: (rotation, position)."""
    rot = torch.tensor([True, True, True, False, False, False]).tile(n)
    return rot, torch.logical_not(rot)


def get_6d_pose_errors(robot, x: torch.Tensor, target_poses: torch.Tensor):
    """[n, 6, 1] pose errors [roll, pitch, yaw, x, y, z] (rad, m) and the current poses [n, 7].

    `target_poses` is the stacked [n, 7] tensor the reference passes, or the unstacked [W, 7] path (n % W == 0)."""
    n = x.shape[0]
    assert target_poses.shape[0] > 0 and n % target_poses.shape[0] == 0, (tuple(target_poses.shape), n)
    return robot.pose_errors(x, target_poses, want_current_poses=True)


def clamp_to_joint_limits(robot, x: torch.Tensor, verbosity: int = 0) -> torch.Tensor:
    """Clamp every joint column to its limits, in place; returns the same tensor."""
    if verbosity > 0:
        for i, (l, u) in enumerate(robot.actuated_joints_limits):
            if x[:, i].min() < l:
                print(f"clamp_to_joint_limits() | joint {i} is below lower limit {l}")
            if x[:, i].max() > u:
                print(f"clamp_to_joint_limits() | joint {i} is above upper limit {u}")
    return robot.clamp_to_joint_limits(x)


def evaluate_seeds(problem, target_path: torch.Tensor, x: torch.Tensor, parallel_count: int) -> torch.Tensor:
    """Everything the loop control needs to know about `parallel_count` trajectories, as one host tensor [S,16]
    (columns `Robot.PLAN_METRIC_FIELDS`): the x_is_valid maxima, the collision counts and the revolute path length
    (= the loop's TL measure, cppflow/optimization.py:221-227).  Two launches (capsule masks of every seed, per-seed
    reduction) and ONE device-to-host copy per LM iteration instead of the reference's per-quantity `.item()` syncs."""
    W = problem.n_timesteps
    assert x.shape[0] == W * parallel_count, f"x has {x.shape[0]} rows, expected {W} * {parallel_count}"
    target = target_path[:W] if target_path.shape[0] != W else target_path
    robot = problem.robot
    self_m = env_m = None
    if not (SELF_COLLISIONS_IGNORED and ENV_COLLISIONS_IGNORED):
        problem.bind_obstacles()
        masks = robot.collision_masks(x.view(parallel_count, W, -1), only=("self", "env"))
        self_m = None if SELF_COLLISIONS_IGNORED else masks["self_mask"].view(-1)
        env_m = None if ENV_COLLISIONS_IGNORED else masks["env_mask"].view(-1)
    return robot.plan_metrics(x, target, self_m, env_m).cpu()


def x_is_valid(
    problem,
    constraints,
    target_path_stacked: torch.Tensor,
    x: torch.Tensor,
    parallel_count: int,
    results_df=None,
    verbosity: int = 0,
    seed_metrics: Optional[torch.Tensor] = None,
):
    """First seed (in order) whose trajectory satisfies every constraint: returns `(x_i, i, flags)` or
    `(None, None, flags)` with flags = (pose_pos_valid, pose_rot_valid, mjac_rev_valid, mjac_pris_valid,
    is_a_self_collision, is_a_env_collision) of the last seed examined -- the contract of
    cppflow/optimization_utils.py:836-923 (collision flags stay None for seeds that fail a threshold, as there).

    The decision is taken on the host from `evaluate_seeds` (pass `seed_metrics` to reuse one already computed for this x).
    The reference does the collision part with klampt's exact meshes (`:889-900`), which is outside this build; the capsule
    masks of cppflow/collision_detection.py:72-86 are used instead (conservative: capsules bound the links)."""
    assert results_df is None, "results_df logging is dead code in the reference (data_types.py:420-421) and unsupported"
    W = problem.n_timesteps
    assert x.shape[0] == W * parallel_count, f"x has {x.shape[0]} rows, expected {W} * {parallel_count}"
    m = seed_metrics if seed_metrics is not None else evaluate_seeds(problem, target_path_stacked, x, parallel_count)
    assert m.shape[0] == parallel_count and m.shape[1] >= 12
    is_a_self_collision: Optional[bool] = None
    is_a_env_collision: Optional[bool] = None
    flags = (False, False, False, False)
    for i in range(parallel_count):
        all_valid, flags = seed_metrics_are_below_threshold(constraints, (m[i, 0], m[i, 2], m[i, 4], m[i, 5]))
        if not all_valid:
            continue
        if not SELF_COLLISIONS_IGNORED:
            is_a_self_collision = float(m[i, 9]) > 0
            if is_a_self_collision:
                continue
        if not ENV_COLLISIONS_IGNORED:
            is_a_env_collision = float(m[i, 10]) > 0
            if is_a_env_collision:
                continue
        if verbosity > 1:
            print("x_is_valid() |", make_text_green_or_red("x is valid", True))
        return x[i * W : (i + 1) * W, :], i, (*flags, is_a_self_collision, is_a_env_collision)
    if verbosity > 1:
        print("x_is_valid() |", make_text_green_or_red("x is invalid", False))
    return None, None, (*flags, is_a_self_collision, is_a_env_collision)
