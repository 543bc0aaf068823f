"""ctypes binding of libcppflow_hip.so (include/cppflow_hip.h, include/cppflow_hip_debug.h).  There is no CPU fallback: if the shared library has
not been built (`python -m cppflow_amd.build` or `__graft_entry__.build()`), importing a compute entry point raises.
"""

import ctypes
import os
from typing import Optional

import numpy as np

from cppflow_amd.robot_model import MAX_CAPSULES, MAX_DOF, MAX_OBSTACLES, MAX_PAIRS, CanonicalChain

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPPFLOW_HIP_LIB", os.path.join(_HERE, "csrc", "libcppflow_hip.so"))

CPPF_OK = 0
CPPF_ERR_INVALID = -1
CPPF_ERR_HIP = -2
CPPF_ERR_UNSUPPORTED = -3

_f = ctypes.c_float
_i32 = ctypes.c_int32
_vp = ctypes.c_void_p


class RobotDesc(ctypes.Structure):
    """struct cppf_robot_desc"""

    _fields_ = [
        ("ndof", _i32),
        ("F", (_f * 12) * MAX_DOF),
        ("F_ee", _f * 12),
        ("jtype", _i32 * MAX_DOF),
        ("lo", _f * MAX_DOF),
        ("hi", _f * MAX_DOF),
        ("n_capsules", _i32),
        ("cap_link", _i32 * MAX_CAPSULES),
        ("cap_p0", (_f * 3) * MAX_CAPSULES),
        ("cap_p1", (_f * 3) * MAX_CAPSULES),
        ("cap_r", _f * MAX_CAPSULES),
        ("n_pairs", _i32),
        ("pairs", (_i32 * 2) * MAX_PAIRS),
    ]


class LmParams(ctypes.Structure):
    """struct cppf_lm_params"""

    _fields_ = [
        ("lm_lambda", _f),
        ("alpha_position", _f),
        ("alpha_rotation", _f),
        ("n_steps", _i32),
        ("clamp", _i32),
        ("tol_pos_m", _f),
        ("tol_rot_rad", _f),
        ("shape", _i32),
        ("solver", _i32),
        ("solver_gate", _f),
    ]


SHAPE_AUTO, SHAPE_ROW, SHAPE_QUAD = 0, 1, 2
SOLVER_AUTO, SOLVER_F64, SOLVER_F32 = 0, 1, 2  # AUTO = fp32 with the conditioning-gated double-precision redo (the default)

# cppflow_hip_debug.h: per-handle test / tuning switches (cppf_debug_set)
TUNE_DEFAULT = -(2**31)
TUNE_KEYS = {"force_generic": 0, "pcr_max_rows": 1, "quad_max_rows": 2, "dp_persistent": 3, "full_rows": 4, "pcr_lds": 5,
             "rows_pose": 6, "quad_mfma": 7, "spread_kb": 8, "dp_spin_log2": 9, "gate_rel_ppm": 10, "cu_count": 11, "lm_pace": 12}  # fmt: skip
DP_AUTO, DP_RESIDENT, DP_LAUNCHES = 0, 1, 2  # cppf_dp_search's `mode`


class Constraints(ctypes.Structure):
    """struct cppf_constraints"""

    _fields_ = [
        ("max_allowed_position_error_cm", _f),
        ("max_allowed_rotation_error_deg", _f),
        ("max_allowed_mjac_deg", _f),
        ("max_allowed_mjac_cm", _f),
        ("self_collisions_ignored", _i32),
        ("env_collisions_ignored", _i32),
    ]


class FullParams(ctypes.Structure):
    """struct cppf_full_params"""

    _fields_ = [(k, _f) for k in ("lm_lambda", "alpha_position", "alpha_rotation", "alpha_differencing",
                                  "alpha_differencing_prismatic_scaling", "alpha_virtual_configs", "alpha_self_collision",
                                  "alpha_env_collision")] + [(k, _i32) for k in (
        "use_pose", "use_differencing", "use_virtual_configs", "n_virtual_configs", "use_self_collisions",
        "use_env_collisions")] + [
        ("pose_do_scale_down_satisfied", _i32), ("pose_threshold_m", _f), ("pose_threshold_rad", _f), ("pose_scale_down", _f),
        ("differencing_mode", _i32), ("differencing_threshold_rad", _f), ("differencing_threshold_m", _f),
        ("differencing_scale_down", _f), ("differencing_shift_invalid_to_threshold", _i32)]  # fmt: skip


class LmOutputs(ctypes.Structure):
    """struct cppf_lm_outputs (device pointers; 0 = not requested)"""

    _fields_ = [
        ("x_out", _vp),
        ("J_out", _vp),
        ("e_out", _vp),
        ("pos_err_m", _vp),
        ("rot_err_rad", _vp),
        ("self_mask", _vp),
        ("env_mask", _vp),
        ("jlim_mask", _vp),
        ("ext_cost", _vp),
        ("min_self", _vp),
        ("min_env", _vp),
        ("seed_summary", _vp),
        ("n_iters", _vp),
    ]


MAX_BATCH = 16  # CPPF_MAX_BATCH


class LmBatchItem(ctypes.Structure):
    """struct cppf_lm_batch_item: one problem of a batched fused launch (cppf_lm_batch_create)"""

    _fields_ = [("x_in", _vp), ("target", _vp), ("S", _i32), ("W", _i32), ("out", LmOutputs)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "cppf_abi_version": (ctypes.c_int, []),
    "cppf_last_error": (ctypes.c_char_p, []),
    "cppf_build_id": (ctypes.c_char_p, []),
    "cppf_robot_create": (ctypes.c_int, [ctypes.POINTER(RobotDesc), ctypes.c_int, ctypes.POINTER(_vp)]),
    "cppf_robot_destroy": (None, [_vp]),
    "cppf_robot_ndof": (ctypes.c_int, [_vp]),
    "cppf_robot_specialization": (ctypes.c_int, [_vp]),
    "cppf_robot_specialize": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "cppf_debug_rtc_compile": (ctypes.c_int, [ctypes.POINTER(RobotDesc), ctypes.c_char_p]),
    "cppf_debug_set": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "cppf_debug_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "cppf_debug_fused_single_offset": (ctypes.c_int, []),
    "cppf_set_obstacles": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_f), ctypes.POINTER(_f)]),
    "cppf_set_joint_limit_padding": (ctypes.c_int, [_vp, ctypes.POINTER(_f), ctypes.POINTER(_f)]),
    "cppf_forward_kinematics": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cppf_jacobian": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cppf_pose_errors": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "cppf_clamp_to_joint_limits": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "cppf_lm_pose_steps": (
        ctypes.c_int,
        [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LmParams), ctypes.POINTER(LmOutputs), _vp],
    ),
    "cppf_lm_batch_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(LmBatchItem), ctypes.POINTER(LmParams), ctypes.POINTER(_vp)]),
    "cppf_lm_batch_launch": (ctypes.c_int, [_vp, _vp]),
    "cppf_lm_batch_destroy": (None, [_vp]),
    "cppf_collision_masks": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cppf_self_collision_distances": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cppf_env_collision_distances": (
        ctypes.c_int,
        [_vp, _vp, ctypes.c_int, ctypes.POINTER(_f), ctypes.POINTER(_f), _vp, _vp],
    ),
    "cppf_pose_error_metrics": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "cppf_seed_validity": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "cppf_self_collision_distances_jacobian": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "cppf_env_collision_distances_jacobian": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.POINTER(_f), ctypes.POINTER(_f),
                                                             _vp, _vp, _vp]),
    "cppf_mjacs": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    "cppf_plan_metrics": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cppf_select_valid_seed": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.POINTER(Constraints), _vp, _vp]),
    "cppf_select_valid_seed_gathered": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.POINTER(Constraints), _vp, _vp]),
    "cppf_seed_summary": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cppf_lm_full_step": (
        ctypes.c_int,
        [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FullParams), _vp, _vp, _vp, _vp, _vp],
    ),
    "cppf_comm_available": (ctypes.c_int, []),
    "cppf_comm_unique_id": (ctypes.c_int, [_vp]),
    "cppf_comm_init_rank": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "cppf_comm_init_all": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_vp)]),
    "cppf_comm_rank": (ctypes.c_int, [_vp]),
    "cppf_comm_world": (ctypes.c_int, [_vp]),
    "cppf_allgather_bytes": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "cppf_comm_group_begin": (ctypes.c_int, []),
    "cppf_comm_group_end": (ctypes.c_int, []),
    "cppf_comm_destroy": (None, [_vp]),
    "cppf_dp_search": (
        ctypes.c_int,
        [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp],
    ),
    "cppf_dp_table_floats": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "cppf_dp_search_tabled": (
        ctypes.c_int,
        [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    ),
}

_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load libcppflow_hip.so (once).  Raises RuntimeError if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (python -m cppflow_amd.build). "
                "cppflow_amd has no CPU fallback."
            )
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be resident BEFORE this library is
        # loaded so that our DT_NEEDED libamdhip64.so.7 binds to that same in-process runtime: the kernels are launched
        # on torch's streams, and two HIP runtimes in one process do not share devices or streams.
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        # provenance: the binary must be the one the sources next to it produce (it is git-ignored and travels to the GPU
        # box as a built artefact; the hash is compiled in by cppflow_amd/build.py)
        # A deployment without the source tree (a wheel) has nothing to compare with: the check then only warns; it is skipped
        # altogether with CPPF_SKIP_BUILD_ID_CHECK=1 or an explicit CPPFLOW_HIP_LIB.  tests/ and __graft_entry__ keep it strict.
        if "CPPFLOW_HIP_LIB" not in os.environ and os.environ.get("CPPF_SKIP_BUILD_ID_CHECK", "0") != "1":
            from cppflow_amd import build as _build

            try:
                want = _build.source_hash()
            except OSError as e:
                import warnings

                warnings.warn(f"cppflow_amd: cannot hash the library's sources ({e}); loading {LIB_PATH} unchecked")
                want = None
            have = handle.cppf_build_id().decode()
            if want is not None and want != have:
                raise RuntimeError(
                    f"{LIB_PATH} was built from other sources (build id {have}, sources on disk hash to {want}): "
                    "rebuild it (python -m cppflow_amd.build)"
                )
        _lib = handle
    return _lib


def check(rc: int) -> None:
    """0 -> ok; contract violations -> AssertionError (the reference asserts, e.g. cppflow/optimization.py:402-405);
    runtime failures -> RuntimeError."""
    if rc == CPPF_OK:
        return
    msg = lib().cppf_last_error().decode("utf-8", "replace")
    if rc == CPPF_ERR_INVALID:
        raise AssertionError(msg)
    raise RuntimeError(msg)


def chain_to_desc(chain: CanonicalChain) -> RobotDesc:
    d = RobotDesc()
    d.ndof = int(chain.ndof)
    for j in range(chain.ndof):
        for k in range(12):
            d.F[j][k] = float(chain.F[j, k])
        d.jtype[j] = int(chain.jtype[j])
        d.lo[j] = float(chain.lo[j])
        d.hi[j] = float(chain.hi[j])
    for k in range(12):
        d.F_ee[k] = float(chain.F_ee[k])
    d.n_capsules = chain.n_capsules
    for c in range(chain.n_capsules):
        d.cap_link[c] = int(chain.cap_link[c])
        for k in range(3):
            d.cap_p0[c][k] = float(chain.cap_p0[c, k])
            d.cap_p1[c][k] = float(chain.cap_p1[c, k])
        d.cap_r[c] = float(chain.cap_r[c])
    d.n_pairs = chain.n_pairs
    for p in range(chain.n_pairs):
        d.pairs[p][0] = int(chain.pairs[p, 0])
        d.pairs[p][1] = int(chain.pairs[p, 1])
    return d


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(_f))


assert MAX_OBSTACLES == 8 and MAX_PAIRS == 128
