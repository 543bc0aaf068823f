"""Host containers of the reference's `cppflow/data_types.py` that the hot path reads: `Constraints` (`:53-62`),
`PlannerSettings` (`:65-83`), `TimingData` (`:27-50`), `Problem` (`:377-484`).  `Plan` (reporting) is out of scope."""

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from cppflow_amd.config import DEFAULT_RERUN_MJAC_THRESHOLD_CM, DEFAULT_RERUN_MJAC_THRESHOLD_DEG


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


@dataclass
class Plan:
    """What a planner returns (a reduced form of cppflow/data_types.py:86-264: the path, its error metrics and the
    validity verdict; the printing / data-frame plumbing of the reference is not reproduced)."""

    q_path: torch.Tensor  # [T, d]
    pose_path: torch.Tensor  # [T, 7]
    target_path: torch.Tensor  # [T, 7]
    positional_errors_cm: torch.Tensor  # [T]
    rotational_errors_deg: torch.Tensor  # [T]
    mjac_deg: float
    mjac_cm: float
    self_colliding_per_ts: torch.Tensor  # bool [T] (capsules)
    env_colliding_per_ts: torch.Tensor  # bool [T] (capsules)
    is_valid: bool

    @property
    def max_positional_error_cm(self) -> float:
        return float(self.positional_errors_cm.max())

    @property
    def max_rotational_error_deg(self) -> float:
        return float(self.rotational_errors_deg.max())

    def __str__(self) -> str:
        return (
            f"<Plan T={self.q_path.shape[0]} valid={self.is_valid} max_pos_err={self.max_positional_error_cm:.5f} cm "
            f"max_rot_err={self.max_rotational_error_deg:.5f} deg mjac={self.mjac_deg:.3f} deg / {self.mjac_cm:.3f} cm "
            f"self_coll={int(self.self_colliding_per_ts.sum())} env_coll={int(self.env_colliding_per_ts.sum())}>"
        )


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
