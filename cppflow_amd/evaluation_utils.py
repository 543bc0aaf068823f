"""Trajectory metrics with the reference's names (`cppflow/evaluation_utils.py`).

Per-row pose errors come from the HIP kernels (`Robot.pose_error_metrics`); the small joint-space helpers
(`angular_changes`, `prismatic_changes`, ...) are host-side torch expressions on whatever device the path lives on --
they belong to the loop control of `run_lm_alternating_loss`, which stays Python.
"""

import math
from typing import List, Tuple

import numpy as np
import torch


def angular_changes(qpath):
    """Joint-angle change between consecutive configs, wrapped to [-pi, pi) (cppflow/evaluation_utils.py:144-154)."""
    dqs = qpath[1:] - qpath[:-1]
    if isinstance(qpath, torch.Tensor):
        return torch.remainder(dqs + math.pi, 2 * math.pi) - math.pi
    return np.remainder(dqs + np.pi, 2 * np.pi) - np.pi


def prismatic_changes(x: torch.Tensor) -> torch.Tensor:
    return x[1:] - x[:-1]


def calculate_mjac_deg(x: torch.Tensor) -> float:
    return torch.rad2deg(angular_changes(x).abs().max()).item()


def calculate_per_timestep_mjac_deg(x: torch.Tensor) -> torch.Tensor:
    return torch.rad2deg(angular_changes(x).abs()).max(dim=1).values


def calculate_per_timestep_mjac_cm(x: torch.Tensor) -> torch.Tensor:
    return 100 * prismatic_changes(x).abs().max(dim=1).values


def get_mjacs(robot, qpath: torch.Tensor) -> Tuple[float, float]:
    rev, pris = robot.split_configs_to_revolute_and_prismatic(qpath)
    if pris.numel() > 0:
        return calculate_mjac_deg(rev), calculate_per_timestep_mjac_cm(pris).abs().max().item()
    return calculate_mjac_deg(rev), 0.0


def joint_limits_exceeded(robot_joint_limits: List[Tuple[float, float]], qs: np.ndarray) -> Tuple[bool, List[float]]:
    """Percent of configurations violating each joint's limits (cppflow/evaluation_utils.py:16-26)."""
    assert len(robot_joint_limits) == qs.shape[1]
    pcts = []
    for i, (l, u) in enumerate(robot_joint_limits):
        assert l < u
        col = qs[:, i]
        pcts.append(100 * ((col < l).sum() + (u < col).sum()) / qs.shape[0])
    return any(p > 0 for p in pcts), pcts


def calculate_pose_error_cm_deg(robot, x: torch.Tensor, target_path: torch.Tensor):
    """(position error [n] in cm, rotation error [n] in deg) of a config path -- one kernel launch
    (cppflow/evaluation_utils.py:113-116).  `target_path` is [W,7] (n % W == 0) or the stacked [n,7]."""
    pos_m, rot_rad = robot.pose_error_metrics(x, target_path)
    return 100 * pos_m, torch.rad2deg(rot_rad)


def positional_errors(path_1: torch.Tensor, path_2: torch.Tensor) -> torch.Tensor:
    """||t_1 - t_2||_2 per row of two pose paths [n,7] (cppflow/evaluation_utils.py:134-136)."""
    return torch.linalg.vector_norm(path_1[:, 0:3] - path_2[:, 0:3], dim=1)


def rotational_errors(path_1: torch.Tensor, path_2: torch.Tensor) -> torch.Tensor:
    """Geodesic distance per row between the quaternions of two pose paths [n,7] (cppflow/evaluation_utils.py:139-141); the
    formula is the one the reference quotes in-tree (cppflow/data_types.py:408-411), folded to [0, pi].  Pose paths that
    come from joint configurations should use `Robot.pose_error_metrics` instead (one launch, no quaternion round trip)."""
    eps = 1e-7
    dot = torch.clip((path_1[:, 3:7] * path_2[:, 3:7]).sum(dim=1), -1.0, 1.0)
    dist = 2.0 * torch.acos(torch.clamp(dot, -1.0 + eps, 1.0 - eps))
    return torch.abs(torch.remainder(dist + math.pi, 2.0 * math.pi) - math.pi)


def calculate_pose_error_mm_deg_and_mjac_cm_deg(robot, x: torch.Tensor, target_path: torch.Tensor):
    """(position error [n] mm, rotation error [n] deg, mjac deg, mjac cm) of one config path
    (cppflow/evaluation_utils.py:119-131).  As in the reference, the last value is 100 * max |x_prismatic| -- the largest
    prismatic joint VALUE, not its change (`Plan.mjac_cm` is the change)."""
    pos_m, rot_rad = robot.pose_error_metrics(x, target_path)
    rev, pris = robot.split_configs_to_revolute_and_prismatic(x)
    mjac_cm = float(100 * pris.abs().max()) if pris.numel() > 0 else 0.0
    return 1000 * pos_m, torch.rad2deg(rot_rad), calculate_mjac_deg(rev), mjac_cm


def errors_are_below_threshold(
    max_allowed_position_error_cm: float,
    max_allowed_rotation_error_deg: float,
    max_allowed_mjac_deg: float,
    max_allowed_mjac_cm: float,
    error_t_cm: torch.Tensor,
    error_R_deg: torch.Tensor,
    qdeltas_revolute_deg: torch.Tensor,
    qdeltas_prismatic_cm: torch.Tensor,
    verbosity: int = 0,
):
    """cppflow/evaluation_utils.py:29-75: strict '<' on the maxima; no prismatic joints -> mjac_pris is valid."""
    pose_pos_valid = bool((error_t_cm.max() < max_allowed_position_error_cm).item())
    pose_rot_valid = bool((error_R_deg.max() < max_allowed_rotation_error_deg).item())
    mjac_rev_valid = bool((qdeltas_revolute_deg.abs().max() < max_allowed_mjac_deg).item())
    mjac_pris_valid = (
        bool((qdeltas_prismatic_cm.abs().max() < max_allowed_mjac_cm).item()) if qdeltas_prismatic_cm.numel() > 0 else True
    )
    if verbosity > 0:
        for ok, what in ((pose_pos_valid, "pose-position"), (pose_rot_valid, "pose-rotation"),
                         (mjac_rev_valid, "mjac_rev"), (mjac_pris_valid, "mjac_pris")):  # fmt: skip
            if not ok:
                print(f"errors_are_below_threshold() | {what} is invalid")
    flags = (pose_pos_valid, pose_rot_valid, mjac_rev_valid, mjac_pris_valid)
    return all(flags), flags


def seed_metrics_are_below_threshold(constraints, seed_metrics_row) -> Tuple[bool, Tuple[bool, bool, bool, bool]]:
    """Same decision from one row of `Robot.seed_validity` ([max pos cm, max rot deg, mjac deg, mjac cm])."""
    # fp32 against fp32, as `tensor.max() < python_float` compares in torch (the scalar takes the tensor's dtype) and as the device's
    # seed selection does: a maximum equal to the fp32 value of the threshold is NOT below it
    row = torch.as_tensor([float(v) for v in seed_metrics_row], dtype=torch.float32)
    thr = torch.tensor([constraints.max_allowed_position_error_cm, constraints.max_allowed_rotation_error_deg,
                        constraints.max_allowed_mjac_deg, constraints.max_allowed_mjac_cm], dtype=torch.float32)  # fmt: skip
    flags = tuple(bool(v) for v in (row < thr))
    return all(flags), flags
