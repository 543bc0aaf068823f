"""Optimizer parameter record and the two presets of the reference (`cppflow/lm_hyper_parameters.py:14-151`).

`ALT_LOSS_V2_1_POSE` (lambda = 1e-6, alpha_position = 3.5, alpha_rotation = 0.35) parameterises the fused pose-only launches
(`cppf_lm_pose_steps`), `ALT_LOSS_V2_1_DIFF` the coupled step (`cppf_lm_full_step`).  The record type is built from a field
table: the names are the reference's (they are the API), every field is required, positional order as there.
"""

import warnings
from dataclasses import make_dataclass
from typing import Optional

import torch

ALTERNATING_LOSS_MAX_N_STEPS, ALTERNATING_LOSS_RETURN_IF_SOL_FOUND_AFTER = 20, 15
ALTERNATING_LOSS_CONVERGENCE_THRESHOLD = 0.3

_OF, _OB, _OI = Optional[float], Optional[bool], Optional[int]
_FIELD_TABLE = (
    # general
    ("seed_w_only_pose", _OB), ("lm_lambda", float),
    # weights of the residual blocks ('alpha_virtual_configs' multiplies 'alpha_differencing')
    ("alpha_position", _OF), ("alpha_rotation", _OF), ("alpha_differencing", _OF),
    ("alpha_differencing_prismatic_scaling", _OF), ("alpha_virtual_configs", _OF), ("alpha_self_collision", _OF),
    ("alpha_env_collision", _OF),
    # pose block
    ("use_pose", bool), ("pose_do_scale_down_satisfied", bool), ("pose_ignore_satisfied_threshold_scale", _OF),
    ("pose_ignore_satisfied_scale_down", _OF),
    # differencing block
    ("use_differencing", bool), ("differencing_do_ignore_satisfied", bool),
    ("differencing_ignore_satisfied_margin_deg", _OF), ("differencing_ignore_satisfied_margin_cm", _OF),
    ("differencing_do_scale_satisfied", bool), ("differencing_scale_down_satisfied_scale", _OF),
    ("differencing_scale_down_satisfied_shift_invalid_to_threshold", _OB),
    # virtual configurations, collision blocks
    ("use_virtual_configs", bool), ("virtual_configs", Optional[torch.Tensor]), ("n_virtual_configs", _OI),
    ("use_self_collisions", bool), ("use_env_collisions", bool),
)  # fmt: skip


def _check(p) -> None:
    """Consistency rules of the reference's __post_init__ (`:58-81`)."""
    if p.differencing_do_scale_satisfied and not p.use_virtual_configs:
        warnings.warn("differencing_do_scale_satisfied without virtual configs unbalances the first / last configs")
    assert not (p.use_differencing and p.differencing_do_ignore_satisfied and p.differencing_do_scale_satisfied)
    if p.differencing_do_ignore_satisfied or p.differencing_do_scale_satisfied:
        assert p.differencing_ignore_satisfied_margin_deg > 0 and p.differencing_ignore_satisfied_margin_cm > 0
    if p.use_virtual_configs:
        assert p.virtual_configs is not None and isinstance(p.n_virtual_configs, int) and p.n_virtual_configs > 0
    assert not p.use_self_collisions or p.alpha_self_collision > 0
    assert not p.use_env_collisions or p.alpha_env_collision > 0
    if p.pose_do_scale_down_satisfied:
        assert isinstance(p.pose_ignore_satisfied_threshold_scale, float) and p.pose_ignore_satisfied_threshold_scale > 0


OptimizationParameters = make_dataclass("OptimizationParameters", _FIELD_TABLE, namespace={"__post_init__": _check})
OptimizationParameters.__doc__ = "Parameters for the optimizer (fields as cppflow/lm_hyper_parameters.py:14-56)."


def _preset(**given) -> "OptimizationParameters":
    values = {name: (False if tp is bool else None) for name, tp in _FIELD_TABLE}
    values["lm_lambda"] = 1e-6  # "1e-6 seems to be optimal" (reference :88, :123)
    values.update(given)
    return OptimizationParameters(**values)


# cppflow/lm_hyper_parameters.py:86-118 -- expects the 1.5 deg / 3 cm joint-limit padding of dp_search
ALT_LOSS_V2_1_DIFF = _preset(
    alpha_differencing=0.00375, alpha_differencing_prismatic_scaling=1.0, alpha_virtual_configs=1.0,
    alpha_self_collision=0.01, alpha_env_collision=0.01, use_differencing=True, use_virtual_configs=True,
    virtual_configs=torch.tensor([]), n_virtual_configs=4, use_self_collisions=True, use_env_collisions=True,
)  # fmt: skip

# cppflow/lm_hyper_parameters.py:119-151
ALT_LOSS_V2_1_POSE = _preset(
    alpha_position=3.5, alpha_rotation=0.35, use_pose=True,
    differencing_scale_down_satisfied_shift_invalid_to_threshold=True,
)  # fmt: skip
