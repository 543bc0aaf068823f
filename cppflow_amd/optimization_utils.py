"""Hot-path helpers with the reference's names and argument meaning (`cppflow/optimization_utils.py`):
`get_6d_pose_errors` (`:802-820`), `clamp_to_joint_limits` (`:823-833`), `x_is_valid` (`:836-923`), the two row-mask
helpers (`:224-250`) and the dense residual / Jacobian builder `LmResidualFns.get_r_and_J` (`:252-731`) with its
`LmResidual` / `LmJacobian` containers.  Each compute call is one kernel launch through the C ABI.
"""

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from cppflow_amd.config import ENV_COLLISIONS_IGNORED, SELF_COLLISIONS_IGNORED
from cppflow_amd.evaluation_utils import angular_changes, seed_metrics_are_below_threshold
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


_PARTS = ("pose", "differencing", "virtual_configs", "self_collisions", "env_collisions")


@dataclass
class LmResidual:
    """The stacked residual of one trajectory, by part, in the reference's stacking order (cppflow/optimization_utils.py:
    31-125): each part is a column [m_i, 1] or None.  `times` stands in for the reference's ten time_* fields."""

    pose: Optional[torch.Tensor] = None
    differencing: Optional[torch.Tensor] = None
    virtual_configs: Optional[torch.Tensor] = None
    self_collisions: Optional[torch.Tensor] = None
    env_collisions: Optional[torch.Tensor] = None
    pose_invalid_row_idxs: Optional[torch.Tensor] = None
    differencing_invalid_row_idxs: Optional[torch.Tensor] = None
    times: Dict[str, float] = field(default_factory=dict)

    def total_time(self) -> float:
        return float(sum(self.times.values()))

    def get_r(self) -> torch.Tensor:
        return torch.cat([getattr(self, k) for k in _PARTS if getattr(self, k) is not None], dim=0)

    def verify_J(self, J: "LmJacobian") -> None:
        for k in _PARTS:
            r_k, J_k = getattr(self, k), getattr(J, k)
            if r_k is not None and r_k.numel() > 0:
                assert J_k is not None and J_k.shape[0] == r_k.shape[0], f"{k}: residual / Jacobian row mismatch"


@dataclass
class LmJacobian:
    """The matching Jacobian parts, each [m_i, n * ndof] or None (cppflow/optimization_utils.py:127-221)."""

    pose: Optional[torch.Tensor] = None
    differencing: Optional[torch.Tensor] = None
    virtual_configs: Optional[torch.Tensor] = None
    self_collisions: Optional[torch.Tensor] = None
    env_collisions: Optional[torch.Tensor] = None
    pose_invalid_row_idxs: Optional[torch.Tensor] = None
    differencing_invalid_row_idxs: Optional[torch.Tensor] = None
    times: Dict[str, float] = field(default_factory=dict)

    def total_time(self) -> float:
        return float(sum(self.times.values()))

    def get_J(self) -> torch.Tensor:
        return torch.cat([getattr(self, k) for k in _PARTS if getattr(self, k) is not None], dim=0)

    def verify_r(self, r: LmResidual) -> None:
        r.verify_J(self)


class LmResidualFns:
    """Dense residual and Jacobian of ONE trajectory, as `levenberg_marquardt_full` of the reference stacks them
    (cppflow/optimization_utils.py:252-731).  This is the inspection / test path: the production step
    (`Robot.lm_full_step`, cppf_lm_full_step) never forms these matrices -- it accumulates the waypoint-local blocks of
    J^T J and J^T r on the device and eliminates the block-tridiagonal system.  Here every per-row quantity still comes
    from the HIP kernels (pose errors, geometric Jacobian, capsule distances and their Jacobians); only the placement
    into the dense [m, n*ndof] layout is torch indexing.  Sign conventions are the reference's (differencing rows: +1 at
    (t, j), -1 at (t+1, j); virtual-config blocks -I)."""

    @staticmethod
    def _get_residual_pose(robot, x: torch.Tensor, target_path: torch.Tensor):
        e, current = get_6d_pose_errors(robot, x, target_path)
        return e.reshape(6 * x.shape[0], 1), current

    @staticmethod
    def _get_jacobian_pose(robot, x: torch.Tensor) -> torch.Tensor:
        n, d = x.shape
        J = torch.zeros((6 * n, d * n), dtype=x.dtype, device=x.device)
        idx = torch.arange(n, device=x.device)
        J.view(n, 6, n, d)[idx, :, idx, :] = robot.jacobian(x)
        return J

    @staticmethod
    def _scale_down_rows_from_r_J_pose_below_error(r: torch.Tensor, J: torch.Tensor, error_threshold_m: float,
                                                   error_threshold_rad: float, scale: float,
                                                   shift_invalid_to_threshold: bool = False):  # fmt: skip
        """Down-weight the pose rows that already satisfy their threshold (cppflow/optimization_utils.py:287-330; pinned by
        the known answer of tests/optimization_utils_test.py:405-456).  r [6n,1] and J are modified in place; returns
        (r, J, rows that were NOT down-weighted).  Off in both presets; the device step applies it row by row when switched on
        (cppf_full_params.pose_do_scale_down_satisfied)."""
        assert r.shape[0] == J.shape[0] and r.shape[0] % 6 == 0 and r.numel() == r.shape[0] and 0.0 <= scale < 1.0
        rot_rows, _ = _get_rotation_and_position_row_mask(r.shape[0] // 6)
        thr = torch.where(rot_rows.to(r.device), error_threshold_rad, error_threshold_m).to(r.dtype)
        return _down_weight_rows(r, J, thr, scale, shift_invalid_to_threshold)

    @staticmethod
    def _scale_down_rows_from_r_J_differencing_below_error(robot, r: torch.Tensor, J: torch.Tensor, mjac_threshold_m: float,
                                                           mjac_threshold_rad: float, scale: float,
                                                           shift_invalid_to_threshold: bool = False):  # fmt: skip
        """The same for the differencing rows, thresholds by joint type (cppflow/optimization_utils.py:351-396; known answers
        tests/optimization_utils_test.py:122-342).  Returns (J, r, rows not down-weighted) -- J first, as the reference."""
        d = robot.ndof
        n = r.shape[0] // d + 1
        assert r.shape == ((n - 1) * d, 1) and J.shape == ((n - 1) * d, n * d) and 0.0 <= scale < 1.0
        _, pris_rows = _get_prismatic_and_revolute_row_mask(robot, r.shape[0])
        thr = torch.where(pris_rows.to(r.device), mjac_threshold_m, mjac_threshold_rad).to(r.dtype)
        r, J, kept = _down_weight_rows(r, J, thr, scale, shift_invalid_to_threshold)
        return J, r, kept

    @staticmethod
    def _get_residual_differencing(robot, x: torch.Tensor) -> torch.Tensor:
        return angular_changes(x).reshape((x.shape[0] - 1) * robot.ndof, 1)

    @staticmethod
    def _get_jacobian_differencing(robot, x: torch.Tensor) -> torch.Tensor:
        n, d = x.shape
        m = d * (n - 1)
        J = torch.zeros((m, d * n), dtype=x.dtype, device=x.device)
        k = torch.arange(m, device=x.device)
        J[k, k] = 1.0
        J[k, k + d] = -1.0
        return J

    @staticmethod
    def _virtual_rows(pms, n: int, d: int, device) -> torch.Tensor:
        """Indices of the configs that have a virtual twin: the first and the last n_virtual_configs waypoints."""
        nv = pms.n_virtual_configs
        return torch.cat([torch.arange(nv, device=device), torch.arange(n - nv, n, device=device)])

    @staticmethod
    def _get_residual_virtual_joints(pms, robot, x: torch.Tensor) -> torch.Tensor:
        assert x.shape == pms.virtual_configs.shape
        n, d = x.shape
        assert 2 * pms.n_virtual_configs < n, f"{2 * pms.n_virtual_configs} virtual configs for {n} configs"
        rows = LmResidualFns._virtual_rows(pms, n, d, x.device)
        diff = x[rows] - pms.virtual_configs.to(x.device)[rows]
        return (torch.remainder(diff + math.pi, 2 * math.pi) - math.pi).reshape(-1, 1)  # angular_subtraction

    @staticmethod
    def _get_jacobian_virtual_configs(pms, robot, x: torch.Tensor) -> torch.Tensor:
        n, d = x.shape
        rows = LmResidualFns._virtual_rows(pms, n, d, x.device)
        J = torch.zeros((rows.numel() * d, n * d), dtype=x.dtype, device=x.device)
        r = torch.arange(rows.numel() * d, device=x.device)
        J[r, (rows[:, None] * d + torch.arange(d, device=x.device)[None, :]).reshape(-1)] = -1.0
        return J

    @staticmethod
    def _collision_rows(dist: torch.Tensor, jac: torch.Tensor, alpha: float):
        """Active rows of one distance family: r = -alpha * dist where that is > 0 (row-major over (config, pair)),
        J = alpha * d(dist)/dq placed in the config's column block."""
        n, P = dist.shape
        d = jac.shape[2]
        r_all = (-alpha * dist).reshape(-1)
        active = torch.nonzero(r_all > 0).reshape(-1)
        r = r_all[active].reshape(-1, 1)
        if active.numel() == 0:
            return r, None
        cfg = torch.div(active, P, rounding_mode="floor")
        J = torch.zeros((active.numel(), n * d), dtype=dist.dtype, device=dist.device)
        cols = cfg[:, None] * d + torch.arange(d, device=dist.device)[None, :]
        J[torch.arange(active.numel(), device=dist.device)[:, None], cols] = alpha * jac.reshape(n * P, d)[active]
        return r, J

    @staticmethod
    def get_r_and_J(pms, robot, x: torch.Tensor, target_path: torch.Tensor, Tcuboids: Optional[List] = None,
                    cuboids: Optional[List] = None, constraints=None) -> Tuple[LmJacobian, LmResidual]:  # fmt: skip
        """The dense residual / Jacobian of cppflow/optimization_utils.py:486-731, the three "satisfied" options included
        (:514-533 pose scale-down, :562-598 differencing filter / scale-down).  In the reference those options read
        `pms.constraints`, a field OptimizationParameters does not have, so that path cannot run there as committed; here the
        Constraints come from the argument, else from `pms.constraints` if present, else DEFAULT_CONSTRAINTS."""
        assert not (pms.differencing_do_scale_satisfied and pms.differencing_do_ignore_satisfied), "use one or the other, not both"
        thr = satisfied_thresholds(pms, constraints)
        n, d = x.shape
        residual, jacobian = LmResidual(), LmJacobian()
        if pms.use_pose:
            r_pose, _ = LmResidualFns._get_residual_pose(robot, x, target_path)
            J_pose = LmResidualFns._get_jacobian_pose(robot, x)
            if pms.pose_do_scale_down_satisfied:  # before the alphas (optimization_utils.py:514-533)
                r_pose, J_pose, invalid = LmResidualFns._scale_down_rows_from_r_J_pose_below_error(
                    r_pose, J_pose, error_threshold_m=thr["pose_threshold_m"], error_threshold_rad=thr["pose_threshold_rad"],
                    scale=pms.pose_ignore_satisfied_scale_down)  # fmt: skip
                residual.pose_invalid_row_idxs = jacobian.pose_invalid_row_idxs = invalid
            scale = torch.tensor([pms.alpha_rotation] * 3 + [pms.alpha_position] * 3, dtype=x.dtype, device=x.device).repeat(n)
            residual.pose, jacobian.pose = r_pose * scale[:, None], J_pose * scale[:, None]
        if pms.use_differencing:
            r_diff = LmResidualFns._get_residual_differencing(robot, x)
            J_diff = LmResidualFns._get_jacobian_differencing(robot, x)
            if pms.differencing_do_ignore_satisfied:  # option 1 (:572-580): drop the satisfied rows, shift the others
                r_diff, J_diff = filter_rows_from_r_J_differencing(
                    robot, r_diff, J_diff, threshold_rad=thr["differencing_threshold_rad"],
                    threshold_m=thr["differencing_threshold_m"], shift_to_threshold=True)  # fmt: skip
            if pms.differencing_do_scale_satisfied:  # option 2 (:583-597): scale the satisfied rows down
                J_diff, r_diff, invalid = LmResidualFns._scale_down_rows_from_r_J_differencing_below_error(
                    robot, r_diff, J_diff, mjac_threshold_m=thr["differencing_threshold_m"],
                    mjac_threshold_rad=thr["differencing_threshold_rad"], scale=pms.differencing_scale_down_satisfied_scale,
                    shift_invalid_to_threshold=bool(pms.differencing_scale_down_satisfied_shift_invalid_to_threshold))  # fmt: skip
                residual.differencing_invalid_row_idxs = jacobian.differencing_invalid_row_idxs = invalid
            if robot.has_prismatic_joints and not pms.differencing_do_ignore_satisfied:  # (:601: not in filter mode)
                _, pris_rows = _get_prismatic_and_revolute_row_mask(robot, r_diff.shape[0])
                r_diff[pris_rows] *= pms.alpha_differencing_prismatic_scaling
                J_diff[pris_rows] *= pms.alpha_differencing_prismatic_scaling
            residual.differencing = pms.alpha_differencing * r_diff
            jacobian.differencing = pms.alpha_differencing * J_diff
        if pms.use_virtual_configs:
            beta = pms.alpha_virtual_configs * pms.alpha_differencing
            residual.virtual_configs = beta * LmResidualFns._get_residual_virtual_joints(pms, robot, x)
            jacobian.virtual_configs = beta * LmResidualFns._get_jacobian_virtual_configs(pms, robot, x)
        if pms.use_self_collisions:
            jac, dist = robot.self_collision_distances_jacobian(x, return_distances=True)
            residual.self_collisions, jacobian.self_collisions = LmResidualFns._collision_rows(
                dist, jac, pms.alpha_self_collision
            )
        if pms.use_env_collisions and Tcuboids is not None and len(Tcuboids) > 0:
            rs, Js = [], []
            for Tcuboid, cuboid in zip(Tcuboids, cuboids):
                jac, dist = robot.env_collision_distances_jacobian(x, cuboid, Tcuboid, return_distances=True)
                r_o, J_o = LmResidualFns._collision_rows(dist, jac, pms.alpha_env_collision)
                if J_o is not None:
                    rs.append(r_o)
                    Js.append(J_o)
            if rs:
                residual.env_collisions, jacobian.env_collisions = torch.cat(rs, dim=0), torch.cat(Js, dim=0)
        jacobian.verify_r(residual)
        return jacobian, residual


def satisfied_thresholds(pms, constraints=None) -> dict:
    """The thresholds of the "satisfied" row options exactly as LmResidualFns.get_r_and_J forms them
    (cppflow/optimization_utils.py:515-520, 562-567) -- including that the ROTATION threshold of the pose option is
    `scale * max_allowed_rotation_error_deg` used as radians (:518-520 pass degrees where radians are compared) -- for the dense
    mirror below and for the device step (Robot.lm_full_step -> cppf_full_params)."""
    import numpy as np

    from cppflow_amd.data_types import DEFAULT_CONSTRAINTS
    from cppflow_amd.utils import cm_to_m

    c = constraints if constraints is not None else (getattr(pms, "constraints", None) or DEFAULT_CONSTRAINTS)
    out = {"pose_threshold_m": 0.0, "pose_threshold_rad": 0.0, "differencing_threshold_rad": 0.0, "differencing_threshold_m": 0.0}
    if pms.pose_do_scale_down_satisfied:
        out["pose_threshold_m"] = float(pms.pose_ignore_satisfied_threshold_scale * c.max_allowed_position_error_m)
        out["pose_threshold_rad"] = float(pms.pose_ignore_satisfied_threshold_scale * c.max_allowed_rotation_error_deg)
    if pms.differencing_do_ignore_satisfied or pms.differencing_do_scale_satisfied:
        out["differencing_threshold_rad"] = float(np.deg2rad(c.max_allowed_mjac_deg - pms.differencing_ignore_satisfied_margin_deg))
        out["differencing_threshold_m"] = float(cm_to_m(c.max_allowed_mjac_cm - pms.differencing_ignore_satisfied_margin_cm))
    return out


def _down_weight_rows(r: torch.Tensor, J: torch.Tensor, thr: torch.Tensor, scale: float, shift: bool):
    """Rows with |r| < thr are multiplied by `scale` (r and J, in place); optionally the others are moved towards zero by
    thr.  Returns (r, J, mask of the rows left at full weight)."""
    below = r[:, 0].abs() < thr
    r[below] *= scale
    J[below] *= scale
    if shift:
        beyond = r[:, 0].abs() > thr  # the scaled rows cannot qualify: |scale * r| < thr
        r[beyond, 0] -= torch.sign(r[beyond, 0]) * thr[beyond]
    return r, J, torch.logical_not(below)


def filter_rows_from_r_J_differencing(robot, r: torch.Tensor, J: torch.Tensor, threshold_rad: float, threshold_m: float,
                                      shift_to_threshold: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:  # fmt: skip
    """Keep only the differencing rows whose joint change exceeds its threshold, optionally measured from the threshold
    (cppflow/optimization_utils.py:736-768; known answers tests/optimization_utils_test.py:458-588)."""
    assert r.shape[0] == J.shape[0] and r.shape[0] % robot.ndof == 0
    _, pris_rows = _get_prismatic_and_revolute_row_mask(robot, r.shape[0])
    thr = torch.where(pris_rows.to(r.device), threshold_m, threshold_rad).to(r.dtype)
    keep = r[:, 0].abs() > thr
    if shift_to_threshold:
        r[keep, 0] -= torch.sign(r[keep, 0]) * thr[keep]
    return r[keep, :], J[keep, :]


def get_jacobian_finite_differencing(robot, opt_params, x: torch.Tensor, target_path: torch.Tensor, eps: float = 0.01,
                                     constraints=None) -> torch.Tensor:
    """The Jacobian of the stacked residual by forward differences (cppflow/optimization_utils.py:771-799: the reference's debugging
    aid for `LmResidualFns.get_r_and_J`, eps = 0.01 there too).  The reference perturbs one (row, column) entry per evaluation --
    r.numel() x n d evaluations; a perturbed residual vector gives a whole column, so this takes n d + 1 evaluations and returns the
    same matrix.  What comes out is dr/dx, which is MINUS the J of `get_r_and_J`: every residual there is  desired - current  and J
    the Jacobian of `current` (hence x + solve(J^T J + lambda I, J^T r), cppflow/optimization.py:95-113).  Only meaningful while the perturbation adds or drops no residual row (no collision pair changing sign, no
    differencing row filtered): asserted."""
    n, d = x.shape
    _, r0 = LmResidualFns.get_r_and_J(opt_params, robot, x, target_path, constraints=constraints)
    r0 = r0.get_r()
    J = torch.zeros((r0.shape[0], n * d), device=x.device, dtype=x.dtype)
    for j in range(n * d):
        x_diff = x.clone()
        t, k = divmod(j, d)
        x_diff[t, k] += eps
        _, r_new = LmResidualFns.get_r_and_J(opt_params, robot, x_diff, target_path, constraints=constraints)
        r_new = r_new.get_r()
        assert r_new.shape == r0.shape, "the perturbation changed the set of residual rows"
        J[:, j] = (r_new[:, 0] - r0[:, 0]) / eps
    return J


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
