"""Optimizer parameters: same field surface and presets as the reference's `cppflow/lm_hyper_parameters.py`.

Only the pose preset (`ALT_LOSS_V2_1_POSE`, `:119-151`: lambda = 1e-6, alpha_position = 3.5, alpha_rotation = 0.35) is
consumed by this build's kernels; the differencing preset is carried so that `run_lm_optimization` keeps its signature.
"""

import warnings
from dataclasses import dataclass
from typing import Optional

import torch

ALTERNATING_LOSS_MAX_N_STEPS = 20
ALTERNATING_LOSS_RETURN_IF_SOL_FOUND_AFTER = 15
ALTERNATING_LOSS_CONVERGENCE_THRESHOLD = 0.3


@dataclass
class OptimizationParameters:
    seed_w_only_pose: Optional[bool]
    lm_lambda: float
    # alphas
    alpha_position: Optional[float]
    alpha_rotation: Optional[float]
    alpha_differencing: Optional[float]
    alpha_differencing_prismatic_scaling: Optional[float]
    alpha_virtual_configs: Optional[float]
    alpha_self_collision: Optional[float]
    alpha_env_collision: Optional[float]
    # pose residual
    use_pose: bool
    pose_do_scale_down_satisfied: bool
    pose_ignore_satisfied_threshold_scale: Optional[float]
    pose_ignore_satisfied_scale_down: Optional[float]
    # differencing residual
    use_differencing: bool
    differencing_do_ignore_satisfied: bool
    differencing_ignore_satisfied_margin_deg: Optional[float]
    differencing_ignore_satisfied_margin_cm: Optional[float]
    differencing_do_scale_satisfied: bool
    differencing_scale_down_satisfied_scale: Optional[float]
    differencing_scale_down_satisfied_shift_invalid_to_threshold: Optional[bool]
    # virtual configs
    use_virtual_configs: bool
    virtual_configs: Optional[torch.Tensor]
    n_virtual_configs: Optional[int]
    # collisions
    use_self_collisions: bool
    use_env_collisions: bool

    def __post_init__(self):
        # same consistency rules as cppflow/lm_hyper_parameters.py:58-81
        if self.differencing_do_scale_satisfied and not self.use_virtual_configs:
            warnings.warn("differencing_do_scale_satisfied without virtual configs unbalances the first/last configs")
        if self.use_differencing:
            assert not (self.differencing_do_ignore_satisfied and self.differencing_do_scale_satisfied)
        if self.differencing_do_ignore_satisfied or self.differencing_do_scale_satisfied:
            assert self.differencing_ignore_satisfied_margin_deg > 0
            assert self.differencing_ignore_satisfied_margin_cm > 0
        if self.use_virtual_configs:
            assert self.virtual_configs is not None
            assert isinstance(self.n_virtual_configs, int) and self.n_virtual_configs > 0
        if self.use_self_collisions:
            assert self.alpha_self_collision > 0
        if self.use_env_collisions:
            assert self.alpha_env_collision > 0
        if self.pose_do_scale_down_satisfied:
            assert isinstance(self.pose_ignore_satisfied_threshold_scale, float)
            assert self.pose_ignore_satisfied_threshold_scale > 0


def _params(**kw) -> OptimizationParameters:
    base = dict(
        seed_w_only_pose=None, lm_lambda=1e-6, alpha_position=None, alpha_rotation=None, alpha_differencing=None,
        alpha_differencing_prismatic_scaling=None, alpha_virtual_configs=None, alpha_self_collision=None,
        alpha_env_collision=None, use_pose=False, pose_do_scale_down_satisfied=False,
        pose_ignore_satisfied_threshold_scale=None, pose_ignore_satisfied_scale_down=None, use_differencing=False,
        differencing_do_ignore_satisfied=False, differencing_ignore_satisfied_margin_deg=None,
        differencing_ignore_satisfied_margin_cm=None, differencing_do_scale_satisfied=False,
        differencing_scale_down_satisfied_scale=None, differencing_scale_down_satisfied_shift_invalid_to_threshold=None,
        use_virtual_configs=False, virtual_configs=None, n_virtual_configs=None, use_self_collisions=False,
        use_env_collisions=False,
    )  # fmt: skip
    base.update(kw)
    return OptimizationParameters(**base)


# values: cppflow/lm_hyper_parameters.py:86-118
ALT_LOSS_V2_1_DIFF = _params(
    alpha_differencing=0.00375, alpha_differencing_prismatic_scaling=1.0, alpha_virtual_configs=1.0,
    alpha_self_collision=0.01, alpha_env_collision=0.01, use_differencing=True, use_virtual_configs=True,
    virtual_configs=torch.tensor([]), n_virtual_configs=4, use_self_collisions=True, use_env_collisions=True,
)  # fmt: skip

# values: cppflow/lm_hyper_parameters.py:119-151
ALT_LOSS_V2_1_POSE = _params(
    alpha_position=3.5, alpha_rotation=0.35, use_pose=True,
    differencing_scale_down_satisfied_shift_invalid_to_threshold=True,
)  # fmt: skip
