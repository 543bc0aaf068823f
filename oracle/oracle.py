"""ctypes loader for the CPU oracle (oracle/lmik_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
cppflow_amd/ does.  `Oracle(chain, f32=False)` wraps liborc64.so (fp64 ground truth) or liborc32.so (canonical-order
fp32, the bit-level reference for masks and FK).  All arrays cross the boundary as float64 numpy arrays.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> None:
    """Compile liborc64.so / liborc32.so with the Makefile next to this file (gcc)."""
    args = ["make", "-C", _HERE]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


class FullParams(ctypes.Structure):
    """orc_full_params: the fields of OptimizationParameters the coupled step reads (lm_hyper_parameters.py:14-56)."""

    _fields_ = [(k, ctypes.c_double) for k in (
        "lm_lambda", "alpha_position", "alpha_rotation", "alpha_differencing", "alpha_differencing_prismatic_scaling",
        "alpha_virtual_configs", "alpha_self_collision", "alpha_env_collision")] + [(k, ctypes.c_int) for k in (
        "use_pose", "use_differencing", "use_virtual_configs", "n_virtual_configs", "use_self_collisions",
        "use_env_collisions")] + [
        ("pose_do_scale_down_satisfied", ctypes.c_int), ("pose_threshold_m", ctypes.c_double), ("pose_threshold_rad", ctypes.c_double),
        ("pose_scale_down", ctypes.c_double), ("differencing_mode", ctypes.c_int), ("differencing_threshold_rad", ctypes.c_double),
        ("differencing_threshold_m", ctypes.c_double), ("differencing_scale_down", ctypes.c_double),
        ("differencing_shift_invalid_to_threshold", ctypes.c_int)]  # fmt: skip

    @classmethod
    def from_params(cls, p, constraints=None):
        """`constraints`: max_allowed_position_error_cm / rotation_error_deg / mjac_deg / mjac_cm (the reference reads
        `pms.constraints`, which OptimizationParameters does not have; default = the values of its CLI, scripts/evaluate.py:51-56).
        Thresholds restated from cppflow/optimization_utils.py:515-520 (pose: threshold_scale x max_allowed_position_error_m and
        threshold_scale x max_allowed_rotation_error_DEG, the degrees compared with radians as the reference does) and :562-567
        (differencing: deg2rad(max_allowed_mjac_deg - margin_deg), (max_allowed_mjac_cm - margin_cm) / 100)."""
        def f(v):
            return 0.0 if v is None else float(v)

        c = constraints if constraints is not None else getattr(p, "constraints", None)
        pos_cm, rot_deg, mj_deg, mj_cm = ((c.max_allowed_position_error_cm, c.max_allowed_rotation_error_deg, c.max_allowed_mjac_deg,
                                           c.max_allowed_mjac_cm) if c is not None else (0.01, 0.1, 7.0, 2.0))
        pose_on = bool(getattr(p, "pose_do_scale_down_satisfied", False))
        mode = 1 if getattr(p, "differencing_do_ignore_satisfied", False) else (2 if getattr(p, "differencing_do_scale_satisfied", False) else 0)
        assert not (getattr(p, "differencing_do_ignore_satisfied", False) and getattr(p, "differencing_do_scale_satisfied", False))
        ts = f(getattr(p, "pose_ignore_satisfied_threshold_scale", None))
        return cls(f(p.lm_lambda), f(p.alpha_position), f(p.alpha_rotation), f(p.alpha_differencing),
                   f(p.alpha_differencing_prismatic_scaling), f(p.alpha_virtual_configs), f(p.alpha_self_collision),
                   f(p.alpha_env_collision), int(bool(p.use_pose)), int(bool(p.use_differencing)),
                   int(bool(p.use_virtual_configs)), int(p.n_virtual_configs or 0), int(bool(p.use_self_collisions)),
                   int(bool(p.use_env_collisions)),
                   int(pose_on), ts * pos_cm / 100.0 if pose_on else 0.0, ts * rot_deg if pose_on else 0.0,
                   f(getattr(p, "pose_ignore_satisfied_scale_down", None)), mode,
                   float(np.deg2rad(mj_deg - f(getattr(p, "differencing_ignore_satisfied_margin_deg", None)))) if mode else 0.0,
                   (mj_cm - f(getattr(p, "differencing_ignore_satisfied_margin_cm", None))) / 100.0 if mode else 0.0,
                   f(getattr(p, "differencing_scale_down_satisfied_scale", None)),
                   int(bool(getattr(p, "differencing_scale_down_satisfied_shift_invalid_to_threshold", False))))  # fmt: skip


def _lib(f32: bool):
    name = "liborc32.so" if f32 else "liborc64.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.orc_robot_create.restype = ctypes.c_void_p
        lib.orc_robot_create.argtypes = [
            ctypes.c_int, _dp, _dp, _ip, _dp, _dp, ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_int, _ip,
        ]  # fmt: skip
        lib.orc_robot_destroy.argtypes = [ctypes.c_void_p]
        lib.orc_set_threads.argtypes = [ctypes.c_int]
        lib.orc_fk.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        lib.orc_link_frames.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        lib.orc_jacobian.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        lib.orc_pose_errors.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, _dp, _dp]
        lib.orc_lm_step.restype = ctypes.c_int
        lib.orc_lm_step.argtypes = [
            ctypes.c_void_p, _dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
            _dp, _dp, _dp,
        ]  # fmt: skip
        lib.orc_clamp.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int]
        lib.orc_lm_steps.restype = ctypes.c_int
        lib.orc_lm_steps.argtypes = [
            ctypes.c_void_p, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.c_int, _dp,
        ]  # fmt: skip
        lib.orc_pose_metrics.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, _dp, _dp]
        lib.orc_pose_metrics_exact.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, _dp, _dp]
        lib.orc_self_dists.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        lib.orc_env_dists.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp, _dp, _dp]
        lib.orc_capsule_endpoints.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        lib.orc_masks.argtypes = [
            ctypes.c_void_p, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _u8p, _u8p, _u8p, _dp, _dp, _dp,
        ]  # fmt: skip
        lib.orc_angular_changes.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp]
        lib.orc_seed_validity.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp]
        lib.orc_plan_metrics.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, _dp]
        lib.orc_self_dists_grads.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp, _dp]
        lib.orc_env_dists_grads.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp, _dp, _dp, _dp]
        lib.orc_lm_full_step.restype = ctypes.c_int
        lib.orc_lm_full_step.argtypes = [
            ctypes.c_void_p, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FullParams), ctypes.c_int, _dp, _dp,
            _dp, _dp, _ip,
        ]  # fmt: skip
        lib.orc_lm_full_step_banded.restype = ctypes.c_int
        lib.orc_lm_full_step_banded.argtypes = [
            ctypes.c_void_p, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FullParams), ctypes.c_int, _dp, _dp, _dp,
        ]  # fmt: skip
        lib.orc_dp_search.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, _ip, _dp]
        _LIBS[name] = lib
    return _LIBS[name]


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    """One robot bound to one oracle build.  `chain` is a cppflow_amd.robot_model.CanonicalChain (plain arrays)."""

    def __init__(self, chain, f32: bool = False, threads: int = 1):
        self.f32 = f32
        self.lib = _lib(f32)
        self.chain = chain
        self.ndof = int(chain.ndof)
        self.n_caps = int(chain.cap_link.shape[0])
        self.n_pairs = int(chain.pairs.shape[0])
        F, Fee = _d(chain.F), _d(chain.F_ee)
        jt = np.ascontiguousarray(chain.jtype, dtype=np.int32)
        lo, hi = _d(chain.lo), _d(chain.hi)
        cl = np.ascontiguousarray(chain.cap_link, dtype=np.int32)
        p0, p1, cr = _d(chain.cap_p0), _d(chain.cap_p1), _d(chain.cap_r)
        pairs = np.ascontiguousarray(chain.pairs, dtype=np.int32)
        self.h = self.lib.orc_robot_create(
            self.ndof, _p(F), _p(Fee), jt.ctypes.data_as(_ip), _p(lo), _p(hi), self.n_caps, cl.ctypes.data_as(_ip),
            _p(p0), _p(p1), _p(cr), self.n_pairs, pairs.ctypes.data_as(_ip),
        )  # fmt: skip
        assert self.h, "orc_robot_create failed"
        self.lib.orc_set_threads(threads)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.orc_robot_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_threads(self, n: int):
        self.lib.orc_set_threads(int(n))

    def _x(self, x):
        x = _d(x)
        assert x.ndim == 2 and x.shape[1] == self.ndof, x.shape
        return x

    def fk(self, x):
        x = self._x(x)
        out = np.empty((x.shape[0], 7))
        self.lib.orc_fk(self.h, _p(x), x.shape[0], _p(out))
        return out

    def link_frames(self, x):
        x = self._x(x)
        out = np.empty((x.shape[0], self.ndof + 1, 12))
        self.lib.orc_link_frames(self.h, _p(x), x.shape[0], _p(out))
        return out

    def jacobian(self, x):
        x = self._x(x)
        out = np.empty((x.shape[0], 6, self.ndof))
        self.lib.orc_jacobian(self.h, _p(x), x.shape[0], _p(out))
        return out

    def pose_errors(self, x, target):
        x, target = self._x(x), _d(target)
        assert target.shape == (x.shape[0], 7)
        e, cur = np.empty((x.shape[0], 6)), np.empty((x.shape[0], 7))
        self.lib.orc_pose_errors(self.h, _p(x), _p(target), x.shape[0], _p(e), _p(cur))
        return e, cur

    def lm_step(self, x, target, lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35, solver=0):
        """Returns (x_new, J_scaled, e_scaled, n_failed_solves)."""
        x, target = self._x(x), _d(target)
        n = x.shape[0]
        assert target.shape == (n, 7)
        xn, J, e = np.empty_like(x), np.empty((n, 6, self.ndof)), np.empty((n, 6))
        fails = self.lib.orc_lm_step(
            self.h, _p(x), _p(target), n, lm_lambda, alpha_position, alpha_rotation, solver, _p(xn), _p(J), _p(e)
        )
        return xn, J, e, fails

    def clamp(self, x):
        x = self._x(x).copy()
        self.lib.orc_clamp(self.h, _p(x), x.shape[0])
        return x

    def lm_steps(self, x, target, n_steps, lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35, solver=0):
        x, target = self._x(x), _d(target)
        n = x.shape[0]
        assert target.shape == (n, 7)
        out = np.empty_like(x)
        self.lib.orc_lm_steps(
            self.h, _p(x), _p(target), n, int(n_steps), lm_lambda, alpha_position, alpha_rotation, solver, _p(out)
        )
        return out

    def pose_metrics(self, x, target):
        x, target = self._x(x), _d(target)
        n = x.shape[0]
        pe, re = np.empty(n), np.empty(n)
        self.lib.orc_pose_metrics(self.h, _p(x), _p(target), n, _p(pe), _p(re))
        return pe, re

    def pose_metrics_exact(self, x, target):
        """positional error and the norm-insensitive evaluation of the geodesic rotation error (see the C source)."""
        x, target = self._x(x), _d(target)
        n = x.shape[0]
        pe, re = np.empty(n), np.empty(n)
        self.lib.orc_pose_metrics_exact(self.h, _p(x), _p(target), n, _p(pe), _p(re))
        return pe, re

    def self_dists(self, x):
        x = self._x(x)
        out = np.empty((x.shape[0], self.n_pairs))
        self.lib.orc_self_dists(self.h, _p(x), x.shape[0], _p(out))
        return out

    def env_dists(self, x, box_lo, box_hi):
        x, lo, hi = self._x(x), _d(box_lo), _d(box_hi)
        out = np.empty((x.shape[0], self.n_caps))
        self.lib.orc_env_dists(self.h, _p(x), x.shape[0], _p(lo), _p(hi), _p(out))
        return out

    def capsule_endpoints(self, x):
        x = self._x(x)
        out = np.empty((x.shape[0], self.n_caps, 6))
        self.lib.orc_capsule_endpoints(self.h, _p(x), x.shape[0], _p(out))
        return out

    def masks(self, x, boxes_lo=None, boxes_hi=None, jl_lo=None, jl_hi=None):
        """Returns dict(self_mask, env_mask, jlim_mask, ext_cost, min_self, min_env) for rows x[n,d]."""
        x = self._x(x)
        n = x.shape[0]
        lo = _d(boxes_lo).reshape(-1, 3) if boxes_lo is not None and len(boxes_lo) else np.zeros((0, 3))
        hi = _d(boxes_hi).reshape(-1, 3) if boxes_hi is not None and len(boxes_hi) else np.zeros((0, 3))
        nobs = lo.shape[0]
        sm, em, jm = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        cost, ms, me = np.empty(n), np.empty(n), np.empty(n)
        jl = _d(jl_lo) if jl_lo is not None else None
        jh = _d(jl_hi) if jl_hi is not None else None
        null = ctypes.cast(None, _dp)
        self.lib.orc_masks(
            self.h, _p(x), n, nobs, _p(lo) if nobs else null, _p(hi) if nobs else null,
            _p(jl) if jl is not None else null, _p(jh) if jh is not None else null,
            sm.ctypes.data_as(_u8p), em.ctypes.data_as(_u8p), jm.ctypes.data_as(_u8p), _p(cost), _p(ms), _p(me),
        )  # fmt: skip
        return dict(self_mask=sm, env_mask=em, jlim_mask=jm, ext_cost=cost, min_self=ms, min_env=me)

    def angular_changes(self, qpath):
        q = _d(qpath)
        T, c = q.shape
        out = np.empty((T - 1, c))
        self.lib.orc_angular_changes(_p(q), T, c, _p(out))
        return out

    def seed_validity(self, x, target, S, W):
        x, target = self._x(x), _d(target)
        assert x.shape[0] == S * W and target.shape == (S * W, 7)
        out = np.empty((S, 4))
        self.lib.orc_seed_validity(self.h, _p(x), _p(target), S, W, _p(out))
        return out

    def plan_metrics(self, x, target, S, W, self_mask=None, env_mask=None, q_init=None):
        """[S,16] Plan metrics of S paths (cppflow/data_types.py:140-264); field order = cppflow_amd.robots.PLAN_METRIC_FIELDS."""
        x, target = self._x(x), _d(target)
        assert x.shape[0] == S * W and target.shape == (S * W, 7)
        out = np.empty((S, 16))
        sm = None if self_mask is None else np.ascontiguousarray(self_mask, dtype=np.uint8).reshape(-1)
        em = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8).reshape(-1)
        qi = None if q_init is None else _d(np.asarray(q_init).reshape(-1))
        self.lib.orc_plan_metrics(
            self.h, _p(x), _p(target), S, W, None if sm is None else sm.ctypes.data_as(ctypes.c_void_p),
            None if em is None else em.ctypes.data_as(ctypes.c_void_p), None if qi is None else qi.ctypes.data_as(ctypes.c_void_p),
            _p(out),
        )  # fmt: skip
        return out

    def dp_search(self, q, ext_cost, prismatic_scaling=5.0):
        """q [k,T,d], ext_cost [k,T] -> (best_idx [T] int, costs [k,T]) per cppflow/search.py:128-191."""
        q = _d(q)
        k, T, d = q.shape
        assert d == self.ndof
        ext = _d(ext_cost)
        assert ext.shape == (k, T)
        idx = np.zeros(T, dtype=np.int32)
        costs = np.empty((k, T))
        self.lib.orc_dp_search(self.h, _p(q), _p(ext), k, T, float(prismatic_scaling), idx.ctypes.data_as(_ip), _p(costs))
        return idx, costs

    def self_dists_grads(self, x):
        x = self._x(x)
        n = x.shape[0]
        dists, grads = np.empty((n, self.n_pairs)), np.empty((n, self.n_pairs, self.ndof))
        self.lib.orc_self_dists_grads(self.h, _p(x), n, _p(dists), _p(grads))
        return dists, grads

    def env_dists_grads(self, x, box_lo, box_hi):
        x, lo, hi = self._x(x), _d(box_lo), _d(box_hi)
        n = x.shape[0]
        dists, grads = np.empty((n, self.n_caps)), np.empty((n, self.n_caps, self.ndof))
        self.lib.orc_env_dists_grads(self.h, _p(x), n, _p(lo), _p(hi), _p(dists), _p(grads))
        return dists, grads

    def lm_full_step(self, x, target, params, S, T, virtual_configs=None, boxes_lo=None, boxes_hi=None, return_residual=False,
                     banded=False, constraints=None):
        """One coupled LM step (cppflow/optimization.py:95-144) for S trajectories x [S*T,d]; target [T,7] shared.
        `params` is an OptimizationParameters-like object.  Returns x_new (and the first trajectory's stacked residual).
        `banded`: the same rows accumulated into band storage and solved by a banded Cholesky (O(T d^3) per trajectory instead of
        the reference's dense O((dT)^3)) -- identical in exact arithmetic, usable at T = 256 .. 512 and hundreds of seeds."""
        x, target = self._x(x), _d(target)
        assert x.shape[0] == S * T and target.shape == (T, 7)
        xv = _d(virtual_configs) if virtual_configs is not None else None
        lo = _d(boxes_lo).reshape(-1, 3) if boxes_lo is not None and len(boxes_lo) else np.zeros((0, 3))
        hi = _d(boxes_hi).reshape(-1, 3) if boxes_hi is not None and len(boxes_hi) else np.zeros((0, 3))
        fp = FullParams.from_params(params, constraints)
        out = np.empty_like(x)
        if banded:
            assert not return_residual
            null = ctypes.cast(None, _dp)
            fails = self.lib.orc_lm_full_step_banded(
                self.h, _p(x), _p(target), _p(xv) if xv is not None else null, S, T, ctypes.byref(fp), lo.shape[0],
                _p(lo) if lo.shape[0] else null, _p(hi) if lo.shape[0] else null, _p(out),
            )  # fmt: skip
            assert fails == 0, f"{fails} Cholesky failures"
            return out
        max_rows = 6 * T + self.ndof * T + 2 * self.ndof * max(fp.n_virtual_configs, 0) + T * (self.n_pairs + self.n_caps * lo.shape[0]) + 8
        r = np.zeros(max_rows)
        rows = ctypes.c_int(0)
        null = ctypes.cast(None, _dp)
        fails = self.lib.orc_lm_full_step(
            self.h, _p(x), _p(target), _p(xv) if xv is not None else null, S, T, ctypes.byref(fp), lo.shape[0],
            _p(lo) if lo.shape[0] else null, _p(hi) if lo.shape[0] else null, _p(out), _p(r), ctypes.byref(rows),
        )  # fmt: skip
        assert fails == 0, f"{fails} Cholesky failures"
        return (out, r[: rows.value]) if return_residual else out
