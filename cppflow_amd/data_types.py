"""Host containers of the reference's `cppflow/data_types.py` that the hot path reads: `Constraints` (`:53-62`),
`PlannerSettings` (`:65-83`), `TimingData` (`:27-50`), `Problem` (`:377-484`), and `Plan` / `PlanNp` (`:86-366`) whose scalar
metrics come from one device reduction for all candidate paths (`cppf_plan_metrics`)."""

import math
from dataclasses import dataclass, field
from time import time
from typing import List, Optional, Tuple

import numpy as np
import torch

from cppflow_amd.config import (
    DEFAULT_RERUN_MJAC_THRESHOLD_CM,
    DEFAULT_RERUN_MJAC_THRESHOLD_DEG,
    SUCCESS_THRESHOLD_initial_q_norm_dist,
)


@dataclass
class TimingData:
    total: float
    ikflow: float
    coll_checking: float
    batch_opt: float
    dp_search: float
    optimizer: float


@dataclass
class Constraints:
    max_allowed_position_error_cm: float
    max_allowed_rotation_error_deg: float
    max_allowed_mjac_deg: float
    max_allowed_mjac_cm: float

    @property
    def max_allowed_position_error_m(self) -> float:
        return self.max_allowed_position_error_cm / 100


# values the reference's CLI uses (scripts/evaluate.py:51-56)
DEFAULT_CONSTRAINTS = Constraints(
    max_allowed_position_error_cm=0.01, max_allowed_rotation_error_deg=0.1, max_allowed_mjac_deg=7.0, max_allowed_mjac_cm=2.0
)


@dataclass
class PlannerSettings:
    k: int
    tmax_sec: float
    anytime_mode_enabled: bool
    latent_distribution: str = "uniform"
    latent_vector_scale: float = 2.0
    run_dp_search: bool = True
    do_rerun_if_optimization_fails: bool = False
    do_rerun_if_large_dp_search_mjac: bool = False
    rerun_mjac_threshold_deg: float = DEFAULT_RERUN_MJAC_THRESHOLD_DEG
    rerun_mjac_threshold_cm: float = DEFAULT_RERUN_MJAC_THRESHOLD_CM
    do_return_search_path_mjac: bool = False
    return_only_1st_plan: bool = False
    verbosity: int = 1

    def __post_init__(self):
        assert self.latent_distribution in {"uniform", "gaussian"}
        assert self.latent_vector_scale > 0.0


_PLAN_METRIC_INDEX = {name: i for i, name in enumerate((
    "max_pos_err_cm", "mean_pos_err_cm", "max_rot_err_deg", "mean_rot_err_deg", "mjac_deg", "mjac_cm", "path_length_rad",
    "path_length_m", "n_joint_limit_violations", "n_self_colliding", "n_env_colliding", "initial_q_norm_dist"))}  # fmt: skip


@dataclass
class Plan:
    """A joint-space path evaluated against its problem: the fields and read-only metrics of the reference's `Plan`
    (cppflow/data_types.py:86-350), same names and units.

    The scalar metrics are not recomputed on the host: `metrics` is one row of `Robot.plan_metrics` (cppf_plan_metrics --
    every candidate path of a planning call is evaluated in one launch, `data_type_utils.plans_from_qpaths`), copied to
    the host once.  A Plan built by hand without `metrics` falls back to the torch expressions of
    `cppflow_amd.evaluation_utils`, which is what the reference does for every property access."""

    q_path: torch.Tensor  # [T, d]
    q_path_revolute: torch.Tensor  # [T, n_revolute]
    q_path_prismatic: torch.Tensor  # [T, n_prismatic]
    pose_path: torch.Tensor  # [T, 7]
    target_path: torch.Tensor  # [T, 7]
    robot_joint_limits: List[Tuple[float, float]]
    self_colliding_per_ts: torch.Tensor  # bool [T]
    env_colliding_per_ts: torch.Tensor  # bool [T]
    positional_errors: torch.Tensor  # [T] metres
    rotational_errors: torch.Tensor  # [T] radians
    provided_initial_configuration: Optional[torch.Tensor]
    constraints: Constraints
    metrics: Optional[torch.Tensor] = None  # host fp32 [>= 12], columns Robot.PLAN_METRIC_FIELDS

    def __post_init__(self):
        T, d = self.target_path.shape[0], len(self.robot_joint_limits)
        assert isinstance(self.q_path, torch.Tensor) and self.q_path.shape == (T, d), (
            f"q_path is {tuple(self.q_path.shape)}, the problem needs {(T, d)}"
        )
        for name in ("q_path_revolute", "q_path_prismatic", "pose_path", "target_path"):
            assert isinstance(getattr(self, name), torch.Tensor), name
        assert self.positional_errors.numel() == T and self.rotational_errors.numel() == T
        if self.metrics is not None:
            self.metrics = self.metrics.detach().to("cpu", torch.float32).reshape(-1)
            assert self.metrics.numel() >= len(_PLAN_METRIC_INDEX)

    def _metric(self, name: str, fallback) -> float:
        if self.metrics is not None:
            return float(self.metrics[_PLAN_METRIC_INDEX[name]])
        return float(fallback())

    # ---- path length (data_types.py:141-153) ----
    @property
    def path_length_rad(self) -> float:
        from cppflow_amd.evaluation_utils import angular_changes

        return self._metric("path_length_rad", lambda: angular_changes(self.q_path_revolute).abs().sum())

    @property
    def path_length_m(self) -> float:
        from cppflow_amd.evaluation_utils import prismatic_changes

        if not self.is_a_prismatic_joint:
            return 0.0
        return self._metric("path_length_m", lambda: prismatic_changes(self.q_path_prismatic).abs().sum())

    @property
    def is_a_prismatic_joint(self) -> bool:
        return self.q_path_prismatic.numel() > 0

    # ---- pose errors (data_types.py:155-186) ----
    @property
    def rotational_errors_deg(self) -> torch.Tensor:
        return torch.rad2deg(self.rotational_errors)

    @property
    def positional_errors_cm(self) -> torch.Tensor:
        return 100 * self.positional_errors

    @property
    def positional_errors_mm(self) -> torch.Tensor:
        return 1000 * self.positional_errors

    @property
    def max_rotational_error_deg(self) -> float:
        return self._metric("max_rot_err_deg", lambda: self.rotational_errors_deg.max())

    @property
    def mean_rotational_error_deg(self) -> float:
        return self._metric("mean_rot_err_deg", lambda: self.rotational_errors_deg.mean())

    @property
    def max_positional_error_cm(self) -> float:
        return self._metric("max_pos_err_cm", lambda: self.positional_errors_cm.max())

    @property
    def mean_positional_error_cm(self) -> float:
        return self._metric("mean_pos_err_cm", lambda: self.positional_errors_cm.mean())

    @property
    def max_positional_error_mm(self) -> float:
        return 10.0 * self.max_positional_error_cm

    @property
    def mean_positional_error_mm(self) -> float:
        return 10.0 * self.mean_positional_error_cm

    # ---- maximum joint changes (data_types.py:188-210) ----
    @property
    def mjac_per_timestep_deg(self) -> torch.Tensor:
        from cppflow_amd.evaluation_utils import calculate_per_timestep_mjac_deg

        return calculate_per_timestep_mjac_deg(self.q_path_revolute)

    @property
    def mjac_per_timestep_cm(self) -> torch.Tensor:
        from cppflow_amd.evaluation_utils import calculate_per_timestep_mjac_cm

        if not self.is_a_prismatic_joint:
            return torch.zeros(self.target_path.shape[0] - 1, device=self.q_path.device, dtype=self.q_path.dtype)
        return calculate_per_timestep_mjac_cm(self.q_path_prismatic)

    @property
    def mjac_deg(self) -> float:
        return self._metric("mjac_deg", lambda: self.mjac_per_timestep_deg.max())

    @property
    def mjac_cm(self) -> float:
        if not self.is_a_prismatic_joint:
            return 0.0
        return self._metric("mjac_cm", lambda: self.mjac_per_timestep_cm.max())

    # ---- validity (data_types.py:212-264) ----
    @property
    def joint_limits_violated(self) -> bool:
        from cppflow_amd.evaluation_utils import joint_limits_exceeded

        if self.metrics is not None:
            return float(self.metrics[_PLAN_METRIC_INDEX["n_joint_limit_violations"]]) > 0
        return bool(joint_limits_exceeded(self.robot_joint_limits, self.q_path)[0])

    @property
    def initial_q_norm_dist(self) -> float:
        if self.provided_initial_configuration is None:
            return 0.0
        q0 = self.provided_initial_configuration.reshape(-1).to(self.q_path.device)
        return self._metric("initial_q_norm_dist", lambda: torch.linalg.vector_norm(q0 - self.q_path[0]))

    def validity_flags(self) -> dict:
        """Every term of the verdict by name (what `is_valid_(verbose=True)` and `__str__` of the reference print)."""
        c = self.constraints

        def below(value, threshold) -> bool:
            # fp32 against fp32, as `tensor.max() < python_float` compares in torch (cppflow/evaluation_utils.py:41-42, 56-59): a
            # maximum equal to the fp32 value of the threshold is not below it
            return bool(np.float32(value) < np.float32(threshold))

        return {
            "mjac_deg": below(self.mjac_deg, c.max_allowed_mjac_deg),
            "mjac_cm": below(self.mjac_cm, c.max_allowed_mjac_cm),
            "max_positional_error": below(self.max_positional_error_cm, c.max_allowed_position_error_cm),
            "max_rotational_error": below(self.max_rotational_error_deg, c.max_allowed_rotation_error_deg),
            "joint_limits": not self.joint_limits_violated,
            "self_collisions": int(self.self_colliding_per_ts.sum()) == 0,
            "env_collisions": int(self.env_colliding_per_ts.sum()) == 0,
            "initial_configuration": self.initial_q_norm_dist < SUCCESS_THRESHOLD_initial_q_norm_dist,
        }

    def is_valid_(self, verbose: bool = False):
        flags = self.validity_flags()
        verdict = all(flags.values())
        if not verbose:
            return verdict
        return verdict, "".join(f"{k}: {v}\n" for k, v in [("is_valid_", verdict), *flags.items()])

    @property
    def is_valid(self) -> bool:
        return self.is_valid_(verbose=False)

    def append_to_results_df(self, df_wrapped: dict) -> None:
        """One row of the reference's results table (data_types.py:120-139; column order of its header comment); the time
        spent in here is excluded from the table's clock as the reference does."""
        t_enter = time()
        n = max(self.self_colliding_per_ts.numel(), 1)
        row = [0, self.is_valid, self.mean_positional_error_mm, self.max_positional_error_mm, self.mean_rotational_error_deg,
               self.max_rotational_error_deg, self.mjac_deg, self.mjac_cm, float(self.self_colliding_per_ts.sum()) / n,
               float(self.env_colliding_per_ts.sum()) / n, self.path_length_rad, self.path_length_m]  # fmt: skip
        df_wrapped["t0"] += time() - t_enter
        row[0] = time() - df_wrapped["t0"]
        df_wrapped["df"].loc[len(df_wrapped["df"])] = row

    def __str__(self) -> str:
        c, flags = self.constraints, self.validity_flags()
        r = lambda v: round(float(v), 5)  # noqa: E731
        lines = [
            "Plan {",
            f"  is_valid:                        {self.is_valid}",
            f"  mjac < {c.max_allowed_mjac_deg} deg:                  {flags['mjac_deg']}",
            f"  mjac < {c.max_allowed_mjac_cm} cm:                   {flags['mjac_cm']}",
            f"  max positional error < {10 * c.max_allowed_position_error_cm} mm:   {flags['max_positional_error']}",
            f"  max rotational error < {c.max_allowed_rotation_error_deg} deg:  {flags['max_rotational_error']}",
            f"  joint limits in bounds:          {flags['joint_limits']}",
            f"  close-to-initial-configuration:  {flags['initial_configuration']}",
            f"  # self collisions:               {int(self.self_colliding_per_ts.sum())}",
            f"  # env. collisions:               {int(self.env_colliding_per_ts.sum())}",
            "  .",
            f"  mjac:                  {r(self.mjac_deg)} deg",
            f"  mjac:                  {r(self.mjac_cm)} cm",
            f"  ave positional error:  {r(self.mean_positional_error_mm)} mm",
            f"  max positional error:  {r(self.max_positional_error_mm)} mm",
            f"  ave rotational error:  {r(self.mean_rotational_error_deg)} deg",
            f"  max rotational error:  {r(self.max_rotational_error_deg)} deg",
            f"  q_initial norm dist:   {r(self.initial_q_norm_dist)}",
            "  .",
            f"  trajectory length:     {r(self.path_length_rad)} rad",
            f"  trajectory length:     {r(self.path_length_m)} m",
            "}",
        ]
        return "\n".join(lines)


class PlanNp:
    """Read-through view of a Plan whose tensor attributes come back as numpy arrays (cppflow/data_types.py:352-366)."""

    def __init__(self, plan: Plan):
        object.__setattr__(self, "plan", plan)

    def __getattr__(self, attr):
        plan = object.__getattribute__(self, "plan")
        if not hasattr(plan, attr):
            raise AttributeError(f"'{attr}' is not an attribute of Plan")
        value = getattr(plan, attr)
        return value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else value


@dataclass
class PlannerResult:
    plan: Plan
    timing: TimingData
    other_plans: List[Plan]
    other_plans_names: List[str]
    debug_info: dict


@dataclass
class Problem:
    constraints: Constraints
    target_path: torch.Tensor  # [W, 7] = x y z qw qx qy qz
    initial_configuration: Optional[torch.Tensor]
    robot: object  # cppflow_amd.robots.Robot
    name: str
    full_name: str
    obstacles: Optional[List] = field(default_factory=list)
    obstacles_Tcuboids: Optional[List] = field(default_factory=list)
    obstacles_cuboids: Optional[List] = field(default_factory=list)
    obstacles_klampt: Optional[List] = field(default_factory=list)  # kept for signature parity; never populated here

    @property
    def n_timesteps(self) -> int:
        return self.target_path.shape[0]

    @property
    def fancy_name(self) -> str:
        return f"{self.robot.formal_robot_name} - {self.name}"

    @property
    def path_length_cumultive_positional_change_cm(self) -> float:
        p = self.target_path[:, 0:3]
        return float(torch.norm(p[1:] - p[:-1], dim=1).sum()) * 100.0

    @property
    def path_length_cumulative_rotational_change_deg(self) -> float:
        """Summed geodesic angle between consecutive target orientations, in degrees (cppflow/data_types.py:403-420).  The
        distance formula clamps its arc-cosine argument to 1 - 1e-7, which reports 2 acos(1 - 1e-7) ~ 1e-3 rad for
        identical orientations; as in the reference that floor is subtracted once per such pair."""
        from cppflow_amd.evaluation_utils import rotational_errors

        eps = 1e-7
        a, b = self.target_path[:-1], self.target_path[1:]
        dot = (a[:, 3:7] * b[:, 3:7]).sum(dim=1)
        n_identical = int(((dot > 1 - eps) | (dot < -1 + eps)).sum())
        # in the path's own dtype: fp32 rounds 1 - 1e-7 to 1 - 2^-23, which makes the floor 9.8e-4 rather than 8.9e-4 rad
        floor_rad = float(2.0 * torch.acos(torch.tensor([1.0 - eps], dtype=self.target_path.dtype)))
        return math.degrees(float(rotational_errors(a, b).abs().sum()) - floor_rad * n_identical)

    def __post_init__(self):
        # unit-quaternion sanity check of cppflow/data_types.py:430-433
        norms = torch.linalg.norm(self.target_path[:, 3:7], dim=1)
        if norms.max() > 1.01 or norms.min() < 0.99:
            raise ValueError("quaternion(s) are not unit quaternion(s)")
        if self.initial_configuration is not None:
            assert self.initial_configuration.dim() == 2, "'initial_configuration' should be [1, ndof]"

    def bind_obstacles(self) -> None:
        """Hand this problem's cuboids to the robot's kernels (host-side copy of <= 8 boxes)."""
        self.robot.set_obstacles(self.obstacles_cuboids or [], self.obstacles_Tcuboids or [])

    def __str__(self) -> str:
        return (
            f"<Problem {self.full_name}: robot={self.robot.name}, waypoints={self.n_timesteps}, "
            f"obstacles={len(self.obstacles_cuboids or [])}, "
            f"path length={self.path_length_cumultive_positional_change_cm / 100.0:.4f} m>"
        )
