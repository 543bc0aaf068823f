"""`Robot`: the duck type the reference's hot path reads off `jrl.robot.Robot`, backed by the HIP library.

Attributes / methods mirrored (reference call sites):
  ndof, name, formal_robot_name, actuated_joints_limits, revolute_joint_idxs, prismatic_joint_idxs,
  has_prismatic_joints, split_configs_to_revolute_and_prismatic        cppflow/optimization_utils.py:228-234, 825-832;
                                                                       cppflow/search.py:46-51, 119-121
  forward_kinematics(x, out_device=, dtype=)                           cppflow/optimization_utils.py:811
  jacobian(x)                                                          cppflow/optimization.py:74
  self_collision_distances(x)                                          cppflow/collision_detection.py:65
  env_collision_distances(x, cuboid, Tcuboid)                          cppflow/collision_detection.py:40
  clamp_to_joint_limits(x), sample_joint_angles(n)                     tests/optimization_test.py:82, 135

Every compute method takes CUDA (ROCm) float32 tensors and launches a hand-written gfx950 kernel through the C ABI on
torch's current stream.  CPU tensors are rejected: this package has no CPU path.
"""

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from cppflow_amd import _hip
from cppflow_amd.robot_model import CanonicalChain, RobotSpec, canonicalize, urdf_forward_kinematics
from cppflow_amd.robot_zoo import ROBOT_SPECS


def _require_device_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on '{t.device}': cppflow_amd runs on MI355X only (no CPU fallback); move the tensor to cuda"
        )
    assert t.dtype == dtype, f"{name} must be {dtype}, is {t.dtype}"
    return t if t.is_contiguous() else t.contiguous()


def _require_output_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """A caller-supplied OUTPUT buffer: the kernel writes through its data_ptr(), so a non-contiguous tensor cannot be
    silently replaced by a contiguous copy (the caller's buffer would never be written)."""
    t = _require_device_tensor(t, name, dtype) if t.is_contiguous() else None
    assert t is not None, f"{name} is an output buffer and must be contiguous"
    return t


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Robot:
    PACKED_BYTES_PER_ROW = 15  # 3 x fp32 (ext_cost, pos_err_m, rot_err_rad) + 3 x u8 (self, env, jlim masks)

    def __init__(self, spec: RobotSpec, specialize: Optional[bool] = None):
        """`specialize` (default: on unless CPPF_SPECIALIZE=0): a robot that is not one of the shipped tables gets its kernels
        compiled for it with hipRTC when its first handle is created (a few seconds once; the code object is cached on disk)."""
        import os

        self.specialize: bool = (os.environ.get("CPPF_SPECIALIZE", "1") != "0") if specialize is None else bool(specialize)
        self.spec: RobotSpec = spec
        self.chain: CanonicalChain = canonicalize(spec)
        self.name: str = spec.name
        self.formal_robot_name: str = spec.formal_name
        self.ndof: int = self.chain.ndof
        self.actuated_joints_limits: List[Tuple[float, float]] = [
            (float(l), float(u)) for l, u in zip(self.chain.lo, self.chain.hi)
        ]
        self.revolute_joint_idxs: List[int] = [j for j in range(self.ndof) if self.chain.jtype[j] == 0]
        self.prismatic_joint_idxs: List[int] = [j for j in range(self.ndof) if self.chain.jtype[j] == 1]
        self.n_capsules: int = self.chain.n_capsules
        self.n_collision_pairs: int = self.chain.n_pairs
        self.collision_capsule_names: List[str] = list(self.chain.cap_names)
        # jrl's ordered map link name -> capsule; column i of env_collision_distances is its i-th key
        # (cppflow/collision_detection.py:137-142).  Values: [p0 (3), p1 (3), radius] in the link's canonical frame.
        self._collision_capsules_by_link: Dict[str, torch.Tensor] = {}
        for c, link_name in enumerate(self.chain.cap_names):
            key = link_name if link_name not in self._collision_capsules_by_link else f"{link_name}#{c}"
            self._collision_capsules_by_link[key] = torch.tensor(
                [*self.chain.cap_p0[c], *self.chain.cap_p1[c], self.chain.cap_r[c]], dtype=torch.float32
            )
        self._desc = _hip.chain_to_desc(self.chain)
        self._handles: Dict[int, ctypes.c_void_p] = {}
        self._tuning: Dict[int, int] = {}  # cppf_debug_set switches, re-applied to every handle this Robot creates
        self._obstacles: Optional[Tuple[np.ndarray, np.ndarray]] = None
        self._jl_padding: Optional[Tuple[np.ndarray, np.ndarray]] = None

    # ---- static properties ------------------------------------------------------------------------------------------
    @property
    def has_prismatic_joints(self) -> bool:
        return len(self.prismatic_joint_idxs) > 0

    def __str__(self) -> str:
        return f"<Robot[{self.name}] ndof={self.ndof} capsules={self.n_capsules} pairs={self.n_collision_pairs}>"

    def split_configs_to_revolute_and_prismatic(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return x[:, self.revolute_joint_idxs], x[:, self.prismatic_joint_idxs]

    def sample_joint_angles(self, n: int) -> np.ndarray:
        lo, hi = self.chain.lo, self.chain.hi
        return np.random.uniform(lo, hi, size=(n, self.ndof))

    def link_frame_at_zero(self, link_name: str) -> np.ndarray:
        """4x4 world transform of a named link at q = 0 (host, fp64).  Used for `path_offset_frame`
        (reference cppflow/data_type_utils.py:65-73, there via klampt)."""
        return urdf_forward_kinematics(self.spec, np.zeros(self.ndof), link=link_name)

    # ---- handle management ---------------------------------------------------------------------------------------------
    def debug_set(self, key: str, value: Optional[int] = None) -> None:
        """Test / tuning switch of THIS robot's handles (include/cppflow_hip_debug.h; `None` restores the default).  Per
        handle: no other Robot in the process is affected."""
        k = _hip.TUNE_KEYS[key]
        self._tuning[k] = _hip.TUNE_DEFAULT if value is None else int(value)
        for h in self._handles.values():
            _hip.check(_hip.lib().cppf_debug_set(h, k, self._tuning[k]))

    def _handle(self, device: torch.device) -> ctypes.c_void_p:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            out = ctypes.c_void_p()
            _hip.check(_hip.lib().cppf_robot_create(ctypes.byref(self._desc), idx, ctypes.byref(out)))
            h = out
            self._handles[idx] = h
            for k, v in self._tuning.items():
                _hip.check(_hip.lib().cppf_debug_set(h, k, v))
            self._apply_obstacles(h)
            self._apply_jl_padding(h)
            if self.specialize and _hip.lib().cppf_robot_specialization(h) < 0:
                # a description that matches none of the generated tables: compile-time tables through hipRTC (cached on disk)
                rc = _hip.lib().cppf_robot_specialize(h, None)
                if rc != _hip.CPPF_OK:
                    import warnings

                    warnings.warn(
                        f"robot '{self.name}': run-time specialisation failed, running the generic kernels: "
                        + _hip.lib().cppf_last_error().decode("utf-8", "replace")[:500]
                    )
        return h

    def specialization(self, device=None) -> int:
        """>= 0: index of the generated table this robot runs; 1000: kernels compiled for it at run time (hipRTC); -1: generic."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        return int(_hip.lib().cppf_robot_specialization(self._handle(dev)))

    def __del__(self):
        try:
            for h in self._handles.values():
                _hip.lib().cppf_robot_destroy(h)
            self._handles = {}
        except Exception:
            pass

    def _apply_obstacles(self, h) -> None:
        if self._obstacles is None:
            _hip.check(_hip.lib().cppf_set_obstacles(h, 0, None, None))
        else:
            cub, rt = self._obstacles
            _hip.check(_hip.lib().cppf_set_obstacles(h, cub.shape[0], _hip.fptr(cub), _hip.fptr(rt)))

    def _apply_jl_padding(self, h) -> None:
        if self._jl_padding is None:
            _hip.check(_hip.lib().cppf_set_joint_limit_padding(h, None, None))
        else:
            lo, hi = self._jl_padding
            _hip.check(_hip.lib().cppf_set_joint_limit_padding(h, _hip.fptr(lo), _hip.fptr(hi)))

    @staticmethod
    def _pack_obstacles(cuboids: Sequence, Tcuboids: Sequence) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """Problem.obstacles_cuboids ([6] each) and obstacles_Tcuboids (4x4 each; only R and t are read -- element
        [3,3] is left 0 by the reference loader, cppflow/data_type_utils.py:120-124)."""
        if cuboids is None or len(cuboids) == 0:
            return None
        assert len(cuboids) == len(Tcuboids)
        cub = np.zeros((len(cuboids), 6), dtype=np.float32)
        rt = np.zeros((len(cuboids), 12), dtype=np.float32)
        for i, (c, T) in enumerate(zip(cuboids, Tcuboids)):
            c = c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)
            T = T.detach().cpu().numpy() if isinstance(T, torch.Tensor) else np.asarray(T)
            cub[i] = c.astype(np.float32).reshape(6)
            rt[i, :9] = T[:3, :3].astype(np.float32).reshape(9)
            rt[i, 9:] = T[:3, 3].astype(np.float32)
        return cub, rt

    def set_obstacles(self, cuboids: Sequence, Tcuboids: Sequence) -> None:
        self._obstacles = self._pack_obstacles(cuboids, Tcuboids)
        for h in self._handles.values():
            self._apply_obstacles(h)

    def set_joint_limit_padding(self, eps_revolute: Optional[float], eps_prismatic: Optional[float]) -> None:
        """Padded limits exactly as cppflow/search.py:46-51 forms them: fp32 tensors, in-place += / -= of the scalar."""
        if eps_revolute is None:
            self._jl_padding = None
        else:
            lo = np.array([l for l, _ in self.actuated_joints_limits], dtype=np.float32)
            hi = np.array([u for _, u in self.actuated_joints_limits], dtype=np.float32)
            er, ep = np.float32(eps_revolute), np.float32(eps_prismatic)
            lo[self.prismatic_joint_idxs] += ep
            lo[self.revolute_joint_idxs] += er
            hi[self.prismatic_joint_idxs] -= ep
            hi[self.revolute_joint_idxs] -= er
            self._jl_padding = (lo, hi)
        for h in self._handles.values():
            self._apply_jl_padding(h)

    def padded_joint_limits(self) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        return self._jl_padding

    def set_padded_joint_limits(self, padded: Optional[Tuple[np.ndarray, np.ndarray]]) -> None:
        """Install already-padded (lo, hi) fp32 arrays (or None to disable the joint-limit mask)."""
        if padded is None:
            self._jl_padding = None
        else:
            lo, hi = (np.ascontiguousarray(a, dtype=np.float32) for a in padded)
            assert lo.shape == (self.ndof,) and hi.shape == (self.ndof,)
            self._jl_padding = (lo, hi)
        for h in self._handles.values():
            self._apply_jl_padding(h)

    # ---- jrl-compatible compute methods --------------------------------------------------------------------------------
    def _x2d(self, x: torch.Tensor, name: str = "x") -> torch.Tensor:
        x = _require_device_tensor(x, name)
        assert x.dim() == 2 and x.shape[1] == self.ndof, f"{name} must be [n, {self.ndof}], is {tuple(x.shape)}"
        return x

    def forward_kinematics(self, x: torch.Tensor, out_device=None, dtype=None) -> torch.Tensor:
        x = self._x2d(x)
        poses = torch.empty((x.shape[0], 7), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_forward_kinematics(
                self._handle(x.device), x.data_ptr(), x.shape[0], poses.data_ptr(), _stream_ptr(x.device)
            )
        )
        if dtype is not None and dtype != torch.float32:
            poses = poses.to(dtype)
        if out_device is not None and torch.device(out_device) != poses.device:
            poses = poses.to(out_device)
        return poses

    def jacobian(self, x: torch.Tensor) -> torch.Tensor:
        x = self._x2d(x)
        J = torch.empty((x.shape[0], 6, self.ndof), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_jacobian(self._handle(x.device), x.data_ptr(), x.shape[0], J.data_ptr(), _stream_ptr(x.device))
        )
        return J

    def self_collision_distances(self, x: torch.Tensor) -> torch.Tensor:
        x = self._x2d(x)
        d = torch.empty((x.shape[0], self.n_collision_pairs), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_self_collision_distances(
                self._handle(x.device), x.data_ptr(), x.shape[0], d.data_ptr(), _stream_ptr(x.device)
            )
        )
        return d

    def env_collision_distances(self, x: torch.Tensor, cuboid, Tcuboid) -> torch.Tensor:
        x = self._x2d(x)
        cub, rt = self._pack_obstacles([cuboid], [Tcuboid])
        d = torch.empty((x.shape[0], self.n_capsules), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_env_collision_distances(
                self._handle(x.device), x.data_ptr(), x.shape[0], _hip.fptr(cub[0]), _hip.fptr(rt[0]), d.data_ptr(),
                _stream_ptr(x.device),
            )  # fmt: skip
        )
        return d

    def self_collision_distances_jacobian(self, x: torch.Tensor, return_distances: bool = False):
        """[n, n_pairs, d] = d(self_collision_distances)/dq (jrl API; call site cppflow/optimization_utils.py:670)."""
        x = self._x2d(x)
        n = x.shape[0]
        J = torch.zeros((n, self.n_collision_pairs, self.ndof), dtype=torch.float32, device=x.device)
        dist = torch.empty((n, self.n_collision_pairs), dtype=torch.float32, device=x.device) if return_distances else None
        _hip.check(
            _hip.lib().cppf_self_collision_distances_jacobian(
                self._handle(x.device), x.data_ptr(), n, J.data_ptr(), dist.data_ptr() if dist is not None else None,
                _stream_ptr(x.device),
            )  # fmt: skip
        )
        return (J, dist) if return_distances else J

    def env_collision_distances_jacobian(self, x: torch.Tensor, cuboid, Tcuboid, return_distances: bool = False):
        """[n, n_capsules, d] = d(env_collision_distances)/dq for ONE cuboid (jrl API; call site optimization_utils.py:710)."""
        x = self._x2d(x)
        n = x.shape[0]
        cub, rt = self._pack_obstacles([cuboid], [Tcuboid])
        J = torch.zeros((n, self.n_capsules, self.ndof), dtype=torch.float32, device=x.device)
        dist = torch.empty((n, self.n_capsules), dtype=torch.float32, device=x.device) if return_distances else None
        _hip.check(
            _hip.lib().cppf_env_collision_distances_jacobian(
                self._handle(x.device), x.data_ptr(), n, _hip.fptr(cub[0]), _hip.fptr(rt[0]), J.data_ptr(),
                dist.data_ptr() if dist is not None else None, _stream_ptr(x.device),
            )  # fmt: skip
        )
        return (J, dist) if return_distances else J

    def clamp_to_joint_limits(self, x: torch.Tensor) -> torch.Tensor:
        """In place (and returned), like cppflow/optimization_utils.py:831-833."""
        x_c = self._x2d(x)
        assert x_c.data_ptr() == x.data_ptr(), "clamp_to_joint_limits mutates its argument: x must be contiguous"
        _hip.check(
            _hip.lib().cppf_clamp_to_joint_limits(self._handle(x.device), x.data_ptr(), x.shape[0], _stream_ptr(x.device))
        )
        return x

    # ---- fused entry points ----------------------------------------------------------------------------------------------
    def pose_errors(self, x: torch.Tensor, target: torch.Tensor, want_current_poses: bool = True):
        """get_6d_pose_errors: returns (e [n,6,1], current_poses [n,7]).  `target` is [W,7] with n % W == 0."""
        x = self._x2d(x)
        target = _require_device_tensor(target, "target_poses")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0, (tuple(target.shape), n)
        e = torch.empty((n, 6, 1), dtype=torch.float32, device=x.device)
        cur = torch.empty((n, 7), dtype=torch.float32, device=x.device) if want_current_poses else None
        _hip.check(
            _hip.lib().cppf_pose_errors(
                self._handle(x.device), x.data_ptr(), target.data_ptr(), n // W, W, e.data_ptr(),
                cur.data_ptr() if cur is not None else None, _stream_ptr(x.device),
            )  # fmt: skip
        )
        return e, cur

    def lm_pose_steps(
        self,
        x: torch.Tensor,
        target: torch.Tensor,
        lm_lambda: float,
        alpha_position: float,
        alpha_rotation: float,
        n_steps: int = 1,
        clamp: bool = True,
        return_residual: bool = False,
        want_errors: bool = False,
        want_collisions: bool = False,
        want_min_dists: bool = False,
        x_out: Optional[torch.Tensor] = None,
        packed_out: Optional[torch.Tensor] = None,
        summary_out: Optional[torch.Tensor] = None,
        tol_pos_m: float = 0.0,
        tol_rot_rad: float = 0.0,
        want_iters: bool = False,
        shape: int = _hip.SHAPE_AUTO,
        solver: int = _hip.SOLVER_AUTO,
    ) -> Dict[str, torch.Tensor]:
        """K fused { levenberg_marquardt_only_pose ; clamp_to_joint_limits } iterations in ONE kernel launch, plus
        (optionally) pose-error metrics and collision masks / search cost of the result.  x is [S*W, d]; target [W, 7].

        `x_out` (may be `x` itself) and `packed_out` let a caller reuse buffers: `packed_out` is a uint8 tensor of
        `PACKED_BYTES_PER_ROW * n` bytes that receives ext_cost | pos_err_m | rot_err_rad (fp32 [n] each) then
        self_mask | env_mask | jlim_mask (u8 [n] each) back to back -- the single buffer one RCCL all-gather ships to
        every rank; it implies want_errors and want_collisions.  `summary_out` (fp32 [S,8], fields
        SEED_SUMMARY_FIELDS) receives the per-seed reduction of `seed_summary`, computed inside the same launch when
        W is 64, 128 or 256 and by a second kernel otherwise (which then needs the per-row outputs, i.e. `packed_out` or
        want_errors + want_collisions).

        `tol_pos_m` / `tol_rot_rad` (both or neither) switch on the in-launch early-out: rows already below tolerance at the
        start of an iteration are left untouched and a wavefront of such rows leaves the loop (cppflow/optimization.py:326-358
        stops the same way once the pose is valid); `want_iters` returns the per-row number of steps applied.  `shape` picks
        the kernel shape (`_hip.SHAPE_ROW`: one row per lane; `_hip.SHAPE_QUAD`: four lanes per row; default: by batch size);
        `solver` the precision of the damped solve: `_hip.SOLVER_AUTO` (default: the reference's dtype with the conditioning gate --
        rows whose fp32 solve is estimated to be off by more than 1e-5 in task space redo it in double precision),
        `_hip.SOLVER_F64` (every row in double precision: a verification mode, ~10x the iteration time) or `_hip.SOLVER_F32` (no gate)."""
        x = self._x2d(x)
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7, tuple(target.shape)
        assert W > 0 and n % W == 0, f"x has {n} rows, not a multiple of the {W} target waypoints"
        dev = x.device
        if x_out is not None:
            x_out = _require_output_tensor(x_out, "x_out")
            assert x_out.shape == x.shape and x_out.device == dev
        res: Dict[str, torch.Tensor] = {"x": x_out if x_out is not None else torch.empty_like(x)}
        out = _hip.LmOutputs()
        out.x_out = res["x"].data_ptr()
        if packed_out is not None:
            assert packed_out.dtype == torch.uint8 and packed_out.is_cuda and packed_out.is_contiguous()
            assert packed_out.numel() == self.PACKED_BYTES_PER_ROW * n and packed_out.data_ptr() % 4 == 0
            f = packed_out[: 12 * n].view(torch.float32)
            res["ext_cost"], res["pos_err_m"], res["rot_err_rad"] = f[:n], f[n : 2 * n], f[2 * n :]
            m = packed_out[12 * n :]
            res["self_mask"], res["env_mask"], res["jlim_mask"] = m[:n], m[n : 2 * n], m[2 * n :]
            for k in ("ext_cost", "pos_err_m", "rot_err_rad", "self_mask", "env_mask", "jlim_mask"):
                setattr(out, k, res[k].data_ptr())
            want_errors = want_collisions = False
        if return_residual:
            res["J"] = torch.empty((n, 6, self.ndof), dtype=torch.float32, device=dev)
            res["e"] = torch.empty((n, 6, 1), dtype=torch.float32, device=dev)
            out.J_out, out.e_out = res["J"].data_ptr(), res["e"].data_ptr()
        if want_errors:
            res["pos_err_m"] = torch.empty(n, dtype=torch.float32, device=dev)
            res["rot_err_rad"] = torch.empty(n, dtype=torch.float32, device=dev)
            out.pos_err_m, out.rot_err_rad = res["pos_err_m"].data_ptr(), res["rot_err_rad"].data_ptr()
        if want_collisions:
            for k in ("self_mask", "env_mask", "jlim_mask"):
                res[k] = torch.empty(n, dtype=torch.uint8, device=dev)
            res["ext_cost"] = torch.empty(n, dtype=torch.float32, device=dev)
            out.self_mask, out.env_mask, out.jlim_mask = (res[k].data_ptr() for k in ("self_mask", "env_mask", "jlim_mask"))
            out.ext_cost = res["ext_cost"].data_ptr()
            if want_min_dists:
                res["min_self"] = torch.empty(n, dtype=torch.float32, device=dev)
                res["min_env"] = torch.empty(n, dtype=torch.float32, device=dev)
                out.min_self, out.min_env = res["min_self"].data_ptr(), res["min_env"].data_ptr()
        if summary_out is not None:
            _check_summary_buffer(summary_out, n // W, dev)
            out.seed_summary = summary_out.data_ptr()
            res["seed_summary"] = summary_out
        if want_iters:
            res["n_iters"] = torch.empty(n, dtype=torch.int32, device=dev)
            out.n_iters = res["n_iters"].data_ptr()
        prm = _hip.LmParams(float(lm_lambda), float(alpha_position), float(alpha_rotation), int(n_steps), int(bool(clamp)),
                            float(tol_pos_m), float(tol_rot_rad), int(shape), int(solver))
        _hip.check(
            _hip.lib().cppf_lm_pose_steps(
                self._handle(dev), x.data_ptr(), target.data_ptr(), n // W, W, ctypes.byref(prm), ctypes.byref(out),
                _stream_ptr(dev),
            )  # fmt: skip
        )
        return res

    def lm_launch_plan(self, x: torch.Tensor, target: torch.Tensor, lm_lambda: float, alpha_position: float,
                       alpha_rotation: float, n_steps: int, x_out: torch.Tensor, packed_out: Optional[torch.Tensor] = None,
                       clamp: bool = True, summary_out: Optional[torch.Tensor] = None,
                       shape: int = _hip.SHAPE_AUTO, solver: int = _hip.SOLVER_AUTO,
                       errors_out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> "LmLaunchPlan":  # fmt: skip
        """Pre-marshalled arguments for repeated fused launches over fixed buffers (what a planner loop or a benchmark
        holds on to): `plan.launch()` is then a single C call on torch's current stream, no Python-side allocation.
        `packed_out` (15 bytes per row: cost, pose errors, masks) makes it a launch with the collision stage; without it
        `errors_out` = (pos_err_m [n], rot_err_rad [n]) asks for the pose errors of the result alone."""
        return LmLaunchPlan(self, x, target, lm_lambda, alpha_position, alpha_rotation, n_steps, x_out, packed_out, clamp,
                            summary_out, shape, solver, errors_out)

    def lm_batch_plan(self, items: Sequence[dict], lm_lambda: float, alpha_position: float, alpha_rotation: float, n_steps: int,
                      clamp: bool = True, solver: int = _hip.SOLVER_AUTO) -> "LmBatchPlan":
        """Several INDEPENDENT problems of this robot in ONE fused launch (cppf_lm_batch_*; at most `_hip.MAX_BATCH`): what a planner
        serving several requests holds on to, and what a GPU that owns only a shard of the candidate seeds needs to fill the chip.
        Each item is a dict with `x` [S*W, d], `target` [W, 7], `x_out` [S*W, d] and optionally `packed_out` (15 bytes per row: cost,
        pose errors, masks -- makes it a launch with the collision stage), `summary_out` [S, 8], `errors_out` = (pos_err_m [n],
        rot_err_rad [n]).  Every problem's results are bit for bit those of `lm_pose_steps` / `lm_launch_plan` on it alone."""
        return LmBatchPlan(self, items, lm_lambda, alpha_position, alpha_rotation, n_steps, clamp, solver)

    def select_valid_seed(self, seed_summary: torch.Tensor, constraints, self_collisions_ignored: bool = False,
                          env_collisions_ignored: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_is_valid's seed selection (cppflow/optimization_utils.py:856-909) over [S,8] per-seed summaries (one GPU's, or
        every rank's after the all-gather): int32 [4] on the device = (first valid seed or -1, number of valid seeds, seed of
        smallest summed external cost, 0).  `constraints` is a `Constraints` (max_allowed_* fields).
        Also accepts what one all-gather of [G, S_local, 8] buffers from `world` ranks leaves behind, as a 4-d tensor
        [world, G, S_local, 8]: the result is then int32 [G, 4], row g for the world * S_local seeds of step g."""
        s = _require_device_tensor(seed_summary, "seed_summary")
        assert s.dim() in (2, 4) and s.shape[-1] == 8, tuple(s.shape)
        n_chunks, n_groups, S_chunk = (1, 1, s.shape[0]) if s.dim() == 2 else tuple(s.shape[:3])
        if out is None:
            out = torch.empty(4 if s.dim() == 2 else (n_groups, 4), dtype=torch.int32, device=s.device)
        else:
            out = _require_output_tensor(out, "out", torch.int32)
            assert out.numel() == 4 * n_groups
        c = _hip.Constraints(float(constraints.max_allowed_position_error_cm), float(constraints.max_allowed_rotation_error_deg),
                             float(constraints.max_allowed_mjac_deg), float(constraints.max_allowed_mjac_cm),
                             int(bool(self_collisions_ignored)), int(bool(env_collisions_ignored)))  # fmt: skip
        _hip.check(
            _hip.lib().cppf_select_valid_seed_gathered(
                self._handle(s.device), s.data_ptr(), n_chunks, n_groups, S_chunk, ctypes.byref(c), out.data_ptr(),
                _stream_ptr(s.device),
            )  # fmt: skip
        )
        return out

    def collision_masks(
        self, q: torch.Tensor, want_min_dists: bool = False, only: Optional[Sequence[str]] = None
    ) -> Dict[str, torch.Tensor]:
        """q [S, W, d] -> self_mask / env_mask / jlim_mask (bool [S,W]) and ext_cost (float [S,W]) in one launch, against
        the obstacles / limit padding last given to set_obstacles / set_joint_limit_padding.  `only` (subset of
        "self", "env", "jlim") restricts the work: the kernel skips what no requested output needs."""
        q = _require_device_tensor(q, "q")
        assert q.dim() == 3 and q.shape[2] == self.ndof, f"q must be [k, ntimesteps, {self.ndof}], is {tuple(q.shape)}"
        S, W, _ = q.shape
        dev = q.device
        parts = ("self", "env", "jlim") if only is None else tuple(only)
        assert set(parts) <= {"self", "env", "jlim"} and len(parts) > 0, parts
        res: Dict[str, torch.Tensor] = {}
        for part in parts:
            res[part + "_mask"] = torch.empty((S, W), dtype=torch.uint8, device=dev)
        if only is None:
            res["ext_cost"] = torch.empty((S, W), dtype=torch.float32, device=dev)
        if want_min_dists:
            if "self" in parts:
                res["min_self"] = torch.empty((S, W), dtype=torch.float32, device=dev)
            if "env" in parts:
                res["min_env"] = torch.empty((S, W), dtype=torch.float32, device=dev)

        def ptr(k):
            return res[k].data_ptr() if k in res else None

        _hip.check(
            _hip.lib().cppf_collision_masks(
                self._handle(dev), q.data_ptr(), S, W, ptr("self_mask"), ptr("env_mask"), ptr("jlim_mask"),
                ptr("ext_cost"), ptr("min_self"), ptr("min_env"), _stream_ptr(dev),
            )  # fmt: skip
        )
        for k in ("self_mask", "env_mask", "jlim_mask"):
            if k in res:
                res[k] = res[k].view(torch.bool)
        return res

    def pose_error_metrics(self, x: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(positional error [n] in metres, rotational error [n] in radians); target [W,7], n % W == 0."""
        x = self._x2d(x)
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0
        pe = torch.empty(n, dtype=torch.float32, device=x.device)
        re = torch.empty(n, dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_pose_error_metrics(
                self._handle(x.device), x.data_ptr(), target.data_ptr(), n // W, W, pe.data_ptr(), re.data_ptr(),
                _stream_ptr(x.device),
            )  # fmt: skip
        )
        return pe, re

    PLAN_METRIC_FIELDS = ("max_pos_err_cm", "mean_pos_err_cm", "max_rot_err_deg", "mean_rot_err_deg", "mjac_deg", "mjac_cm",
                          "path_length_rad", "path_length_m", "n_joint_limit_violations", "n_self_colliding",
                          "n_env_colliding", "initial_q_norm_dist")  # fmt: skip
    SEED_SUMMARY_FIELDS = ("max_pos_err_cm", "max_rot_err_deg", "mjac_rev_deg", "mjac_pris_cm", "n_self_colliding",
                           "n_env_colliding", "n_jlim", "sum_ext_cost")  # fmt: skip

    def seed_summary(self, x: torch.Tensor, packed: torch.Tensor, S: int, W: int, out: Optional[torch.Tensor] = None):
        """[S,8] per-seed reduction (SEED_SUMMARY_FIELDS) of a fused launch's packed per-row outputs and its x_out."""
        x = self._x2d(x)
        n = S * W
        assert x.shape[0] == n and packed.dtype == torch.uint8 and packed.numel() == self.PACKED_BYTES_PER_ROW * n
        assert packed.is_cuda and packed.is_contiguous() and packed.data_ptr() % 4 == 0
        if out is None:
            out = torch.empty((S, 8), dtype=torch.float32, device=x.device)
        base = packed.data_ptr()
        _hip.check(
            _hip.lib().cppf_seed_summary(
                self._handle(x.device), x.data_ptr(), S, W, base, base + 4 * n, base + 8 * n, base + 12 * n, base + 13 * n,
                base + 14 * n, out.data_ptr(), _stream_ptr(x.device),
            )  # fmt: skip
        )
        return out

    def lm_full_step(self, x: torch.Tensor, target: torch.Tensor, opt_params, virtual_configs: Optional[torch.Tensor] = None,
                     x_out: Optional[torch.Tensor] = None, constraints=None) -> torch.Tensor:  # fmt: skip
        """One coupled LM step (levenberg_marquardt_full, cppflow/optimization.py:95-144) for every trajectory in
        x [S*W, d]; target [W,7].  `opt_params` is an OptimizationParameters (e.g. ALT_LOSS_V2_1_DIFF); its "satisfied" row
        options (pose scale-down, differencing filter / scale-down: cppflow/optimization_utils.py:514-533, 562-598) are applied
        on the device, with thresholds from `constraints` (default: opt_params.constraints if present, else DEFAULT_CONSTRAINTS)."""
        x = self._x2d(x)
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0
        p = opt_params
        assert not (p.differencing_do_scale_satisfied and p.differencing_do_ignore_satisfied), "use one or the other, not both"
        from cppflow_amd.optimization_utils import satisfied_thresholds

        thr = satisfied_thresholds(p, constraints)
        xv = None
        if virtual_configs is not None and p.use_virtual_configs and virtual_configs.numel() > 0:
            xv = self._x2d(virtual_configs, "virtual_configs")
            assert xv.shape == x.shape, "virtual_configs must have the shape of x (optimization_utils.py:433)"

        def f(v):
            return 0.0 if v is None else float(v)

        prm = _hip.FullParams(f(p.lm_lambda), f(p.alpha_position), f(p.alpha_rotation), f(p.alpha_differencing),
                              f(p.alpha_differencing_prismatic_scaling), f(p.alpha_virtual_configs),
                              f(p.alpha_self_collision), f(p.alpha_env_collision), int(bool(p.use_pose)),
                              int(bool(p.use_differencing)), int(bool(p.use_virtual_configs)), int(p.n_virtual_configs or 0),
                              int(bool(p.use_self_collisions)), int(bool(p.use_env_collisions)),
                              int(bool(p.pose_do_scale_down_satisfied)), thr["pose_threshold_m"], thr["pose_threshold_rad"],
                              f(p.pose_ignore_satisfied_scale_down),
                              1 if p.differencing_do_ignore_satisfied else (2 if p.differencing_do_scale_satisfied else 0),
                              thr["differencing_threshold_rad"], thr["differencing_threshold_m"],
                              f(p.differencing_scale_down_satisfied_scale),
                              int(bool(p.differencing_scale_down_satisfied_shift_invalid_to_threshold)))  # fmt: skip
        d = self.ndof
        nt = d * (d + 1) // 2
        dev = x.device
        blocks = torch.empty(n * (nt + d), dtype=torch.float32, device=dev)
        G = torch.empty(n * d * d, dtype=torch.float32, device=dev)  # L_t (dense) in the parallel-in-time form, G_t (packed) else
        y = torch.empty(n * d, dtype=torch.float32, device=dev)
        if x_out is None:
            x_out = torch.empty_like(x)
        _hip.check(
            _hip.lib().cppf_lm_full_step(
                self._handle(dev), x.data_ptr(), target.data_ptr(), xv.data_ptr() if xv is not None else None, n // W, W,
                ctypes.byref(prm), blocks.data_ptr(), G.data_ptr(), y.data_ptr(), x_out.data_ptr(), _stream_ptr(dev),
            )  # fmt: skip
        )
        return x_out

    def dp_search(self, q: torch.Tensor, ext_cost: torch.Tensor, prismatic_joint_scaling: float = 5.0, method: str = "auto",
                  return_memo: bool = False, return_method: bool = False):
        """q [k,T,d], ext_cost [k,T] -> (best_path [T,d], best_idx [T] int32, cost table [T,k]); cppflow/search.py:128-191.
        `method`: "table" (k <= 256: transition table + the recurrence on one compute unit, cppf_dp_search_tabled), "resident"
        (cppf_dp_search: ONE launch of resident workgroups handing the cost row on up to k = 1024, one launch per waypoint beyond),
        "launches" (one launch per waypoint, any k -- for THIS call, no handle state involved), "auto" = whichever is fastest at this k
        (table up to 128 candidates, 192 for chains of more than 8 joints; resident up to 1024; launches beyond).  Bit-identical results.
        `return_method` appends the method that ran ("table" / "resident" / "launches") to the result."""
        assert method in ("auto", "table", "resident", "launches"), method
        q = _require_device_tensor(q, "q")
        ext_cost = _require_device_tensor(ext_cost, "q_costs_external")
        assert q.dim() == 3 and q.shape[2] == self.ndof, tuple(q.shape)
        k, T, d = q.shape
        assert ext_cost.shape == (k, T), (tuple(ext_cost.shape), (k, T))
        dev = q.device
        qT = torch.empty((T, k, d), dtype=torch.float32, device=dev)
        costsT = torch.empty((T, k), dtype=torch.float32, device=dev)
        memoT = torch.empty((T, k), dtype=torch.int32, device=dev)
        best_path = torch.empty((T, d), dtype=torch.float32, device=dev)
        best_idx = torch.empty(T, dtype=torch.int32, device=dev)
        n_table = ctypes.c_size_t(0)
        _hip.check(_hip.lib().cppf_dp_table_floats(k, T, ctypes.byref(n_table)))
        # measured (scripts/dp_bench.py, profiles/r3_dp_bench.txt): the table form wins up to k = 128 (k = 64: 168 vs 321 us, 128: 365
        # vs 442); from 129 candidates on its rows take the 192-wide register sets (~540 us at T = 256 whatever k) and the resident
        # hand-off form is ahead (k = 175: 491 vs 537 us, 256: 493 vs 812) -- unless the chain has more than 8 joints, whose
        # per-pair arithmetic in front of every hand-off keeps the table form ahead up to k = 192 (12 joints: 551 vs 618 us)
        k_table = 128 if d <= 8 else 192
        tabled = method == "table" or (method == "auto" and k <= k_table and 2 <= T <= 65536 and n_table.value * 4 <= (1 << 30))
        if tabled:
            ran = "table"
            table = torch.empty(max(n_table.value, 1), dtype=torch.float32, device=dev)
            _hip.check(
                _hip.lib().cppf_dp_search_tabled(
                    self._handle(dev), q.data_ptr(), ext_cost.data_ptr(), k, T, float(prismatic_joint_scaling), qT.data_ptr(),
                    costsT.data_ptr(), memoT.data_ptr(), table.data_ptr(), best_path.data_ptr(), best_idx.data_ptr(),
                    _stream_ptr(dev),
                )  # fmt: skip
            )
        else:
            # ("resident" = this entry point as it decides itself, i.e. resident up to 1024 candidates unless the handle's
            # CPPF_TUNE_DP_PERSISTENT test switch is 0; "launches" forces one launch per waypoint for this call only)
            mode = _hip.DP_LAUNCHES if method == "launches" else _hip.DP_AUTO
            resident_on = self._tuning.get(_hip.TUNE_KEYS["dp_persistent"], 1) not in (0,)
            ran = "resident" if T >= 2 and k <= 1024 and mode == _hip.DP_AUTO and resident_on else "launches"
            _hip.check(
                _hip.lib().cppf_dp_search(
                    self._handle(dev), q.data_ptr(), ext_cost.data_ptr(), k, T, float(prismatic_joint_scaling), qT.data_ptr(),
                    costsT.data_ptr(), memoT.data_ptr(), best_path.data_ptr(), best_idx.data_ptr(), mode, _stream_ptr(dev),
                )  # fmt: skip
            )
        res = (best_path, best_idx, costsT) + ((memoT,) if return_memo else ())
        return res + ((ran,) if return_method else ())

    def plan_metrics(self, x: torch.Tensor, target: torch.Tensor, self_mask: Optional[torch.Tensor] = None,
                     env_mask: Optional[torch.Tensor] = None, q_init: Optional[torch.Tensor] = None) -> torch.Tensor:  # fmt: skip
        """[S,16] `Plan` metrics (cppflow/data_types.py:140-264) of S paths in one launch; columns PLAN_METRIC_FIELDS.
        x [S*W,d] (or [S,W,d]); masks are per-row uint8 / bool tensors of the same rows (optional); q_init [d] or [1,d]."""
        x = self._x2d(x.reshape(-1, self.ndof) if x.dim() == 3 else x)
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0
        ptrs = []
        for m, nm in ((self_mask, "self_mask"), (env_mask, "env_mask")):
            if m is None:
                ptrs.append(None)
                continue
            assert m.is_cuda and m.numel() == n and m.dtype in (torch.uint8, torch.bool), f"{nm} must be uint8 / bool [{n}]"
            m = m.contiguous()
            ptrs.append(m)
        qi = None
        if q_init is not None:
            qi = _require_device_tensor(q_init, "q_init").reshape(-1).contiguous()
            assert qi.numel() == self.ndof, tuple(q_init.shape)
        out = torch.empty((n // W, 16), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_plan_metrics(
                self._handle(x.device), x.data_ptr(), target.data_ptr(), n // W, W,
                ptrs[0].data_ptr() if ptrs[0] is not None else None, ptrs[1].data_ptr() if ptrs[1] is not None else None,
                qi.data_ptr() if qi is not None else None, out.data_ptr(), _stream_ptr(x.device),
            )  # fmt: skip
        )
        return out

    def mjacs(self, q: torch.Tensor, prismatic_joint_scaling: float = 5.0) -> torch.Tensor:
        """q [k,T,d] -> [k,k,T-1] maximum (wrapped, prismatic-scaled) joint change from candidate j at t to candidate i at
        t+1 (cppflow/search.py:100-125).  `dp_search` does not use it -- it is the tensor the reference builds."""
        q = _require_device_tensor(q, "q")
        assert q.dim() == 3 and q.shape[2] == self.ndof, tuple(q.shape)
        k, T, _ = q.shape
        out = torch.empty((k, k, max(T - 1, 0)), dtype=torch.float32, device=q.device)
        _hip.check(
            _hip.lib().cppf_mjacs(
                self._handle(q.device), q.contiguous().data_ptr(), k, T, float(prismatic_joint_scaling), out.data_ptr(),
                _stream_ptr(q.device),
            )
        )
        return out

    def seed_validity(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """[S,4]: per seed max position error (cm), max rotation error (deg), mjac revolute (deg), mjac prismatic (cm)."""
        x = self._x2d(x)
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0
        out = torch.empty((n // W, 4), dtype=torch.float32, device=x.device)
        _hip.check(
            _hip.lib().cppf_seed_validity(
                self._handle(x.device), x.data_ptr(), target.data_ptr(), n // W, W, out.data_ptr(), _stream_ptr(x.device)
            )
        )
        return out


def _check_summary_buffer(t: torch.Tensor, S: int, dev) -> None:
    assert t.shape == (S, 8) and t.dtype == torch.float32 and t.is_cuda and t.is_contiguous() and t.device == dev, (
        f"summary_out must be a contiguous fp32 [{S}, 8] tensor on {dev}"
    )


class LmLaunchPlan:
    def __init__(self, robot: Robot, x, target, lm_lambda, alpha_position, alpha_rotation, n_steps, x_out, packed_out, clamp,
                 summary_out=None, shape=_hip.SHAPE_AUTO, solver=_hip.SOLVER_AUTO, errors_out=None):  # fmt: skip
        x = robot._x2d(x)
        x_out = _require_output_tensor(x_out, "x_out")
        target = _require_device_tensor(target, "target_path")
        n, W = x.shape[0], target.shape[0]
        assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0 and x_out.shape == x.shape
        self._keep = (robot, x, target, x_out, packed_out)  # the plan owns references to every buffer it points at
        self._summary_keep = summary_out
        self.outputs: Dict[str, torch.Tensor] = {"x": x_out}
        out = _hip.LmOutputs()
        out.x_out = x_out.data_ptr()
        if packed_out is not None:
            assert packed_out.dtype == torch.uint8 and packed_out.is_cuda and packed_out.is_contiguous()
            assert packed_out.numel() == robot.PACKED_BYTES_PER_ROW * n and packed_out.data_ptr() % 4 == 0
            f = packed_out[: 12 * n].view(torch.float32)
            m = packed_out[12 * n :]
            views = dict(ext_cost=f[:n], pos_err_m=f[n : 2 * n], rot_err_rad=f[2 * n :], self_mask=m[:n],
                         env_mask=m[n : 2 * n], jlim_mask=m[2 * n :])  # fmt: skip
            for k, v in views.items():
                setattr(out, k, v.data_ptr())
            self.outputs.update(views)
        if errors_out is not None:
            assert packed_out is None, "packed_out already holds the pose errors"
            pe, re = (_require_output_tensor(t, nm) for t, nm in zip(errors_out, ("pos_err_m", "rot_err_rad")))
            assert pe.numel() == n and re.numel() == n
            out.pos_err_m, out.rot_err_rad = pe.data_ptr(), re.data_ptr()
            self.outputs.update(pos_err_m=pe, rot_err_rad=re)
            self._keep = self._keep + (pe, re)
        if summary_out is not None:
            _check_summary_buffer(summary_out, n // W, x.device)
            out.seed_summary = summary_out.data_ptr()
            self.outputs["seed_summary"] = summary_out
        self._out = out
        self._prm = _hip.LmParams(float(lm_lambda), float(alpha_position), float(alpha_rotation), int(n_steps), int(bool(clamp)),
                                  0.0, 0.0, int(shape), int(solver))
        self._fn = _hip.lib().cppf_lm_pose_steps
        self._args = (robot._handle(x.device), x.data_ptr(), target.data_ptr(), n // W, W, ctypes.byref(self._prm),
                      ctypes.byref(self._out))  # fmt: skip
        self._device = x.device

    def launch(self) -> None:
        rc = self._fn(*self._args, torch.cuda.current_stream(self._device).cuda_stream)
        if rc:
            _hip.check(rc)

    def launch_on(self, stream: "torch.cuda.Stream") -> None:
        """`launch()` on an explicit HIP stream: independent batches in flight on two streams overlap one launch's tail with
        the next one's ramp-up (each batch needs its own output buffers)."""
        rc = self._fn(*self._args, stream.cuda_stream)
        if rc:
            _hip.check(rc)

    def set_summary_out(self, summary_out: torch.Tensor) -> None:
        """Point the next launches' per-seed summary at another [S,8] buffer (a ring of buffers in flight on the wire)."""
        _check_summary_buffer(summary_out, self._args[3], self._device)
        self._out.seed_summary = summary_out.data_ptr()
        self.outputs["seed_summary"] = summary_out

    def summary_launcher(self, out: torch.Tensor):
        """A zero-allocation callable that reduces this plan's per-row outputs into `out` [S,8] (`Robot.seed_summary`)."""
        robot, x, target, x_out, packed = self._keep[:5]
        assert packed is not None, "the plan has no packed per-row outputs"
        n, W = x.shape[0], target.shape[0]
        assert out.shape == (n // W, 8) and out.dtype == torch.float32 and out.is_cuda and out.is_contiguous()
        base = packed.data_ptr()
        fn = _hip.lib().cppf_seed_summary
        args = (robot._handle(self._device), x_out.data_ptr(), n // W, W, base, base + 4 * n, base + 8 * n, base + 12 * n,
                base + 13 * n, base + 14 * n, out.data_ptr())  # fmt: skip
        device = self._device

        def launch() -> None:
            rc = fn(*args, torch.cuda.current_stream(device).cuda_stream)
            if rc:
                _hip.check(rc)

        return launch


class LmBatchPlan:
    """`Robot.lm_batch_plan`: owns a cppf_lm_batch (a device table of the items' pointers) and references to every buffer in it."""

    def __init__(self, robot: Robot, items, lm_lambda, alpha_position, alpha_rotation, n_steps, clamp=True, solver=_hip.SOLVER_AUTO):
        assert 1 <= len(items) <= _hip.MAX_BATCH, f"a batch holds 1 .. {_hip.MAX_BATCH} problems"
        arr = (_hip.LmBatchItem * len(items))()
        self._keep, self.outputs, dev = [robot], [], None
        for i, it in enumerate(items):
            x = robot._x2d(it["x"])
            target = _require_device_tensor(it["target"], "target_path")
            x_out = _require_output_tensor(it["x_out"], "x_out")
            n, W = x.shape[0], target.shape[0]
            assert target.dim() == 2 and target.shape[1] == 7 and W > 0 and n % W == 0 and x_out.shape == x.shape
            dev = x.device if dev is None else dev
            assert x.device == dev and target.device == dev and x_out.device == dev, "every buffer of a batch lives on one device"
            arr[i].x_in, arr[i].target, arr[i].S, arr[i].W = x.data_ptr(), target.data_ptr(), n // W, W
            out, views = arr[i].out, {"x": x_out}
            out.x_out = x_out.data_ptr()
            self._keep += [x, target, x_out]
            packed = it.get("packed_out")
            if packed is not None:
                assert packed.dtype == torch.uint8 and packed.is_cuda and packed.is_contiguous() and packed.device == dev
                assert packed.numel() == robot.PACKED_BYTES_PER_ROW * n and packed.data_ptr() % 4 == 0
                f, m = packed[: 12 * n].view(torch.float32), packed[12 * n :]
                v = dict(ext_cost=f[:n], pos_err_m=f[n : 2 * n], rot_err_rad=f[2 * n :], self_mask=m[:n], env_mask=m[n : 2 * n],
                         jlim_mask=m[2 * n :])  # fmt: skip
                for k, t in v.items():
                    setattr(out, k, t.data_ptr())
                views.update(v)
                self._keep.append(packed)
            if it.get("errors_out") is not None:
                assert packed is None, "packed_out already holds the pose errors"
                pe, re = (_require_output_tensor(t, nm) for t, nm in zip(it["errors_out"], ("pos_err_m", "rot_err_rad")))
                assert pe.numel() == n and re.numel() == n
                out.pos_err_m, out.rot_err_rad = pe.data_ptr(), re.data_ptr()
                views.update(pos_err_m=pe, rot_err_rad=re)
                self._keep += [pe, re]
            if it.get("summary_out") is not None:
                _check_summary_buffer(it["summary_out"], n // W, dev)
                out.seed_summary = it["summary_out"].data_ptr()
                views["seed_summary"] = it["summary_out"]
                self._keep.append(it["summary_out"])
            self.outputs.append(views)
        prm = _hip.LmParams(float(lm_lambda), float(alpha_position), float(alpha_rotation), int(n_steps), int(bool(clamp)),
                            0.0, 0.0, _hip.SHAPE_ROW, int(solver))
        self._h = ctypes.c_void_p()
        self._lib = _hip.lib()
        _hip.check(self._lib.cppf_lm_batch_create(robot._handle(dev), len(items), arr, ctypes.byref(prm), ctypes.byref(self._h)))
        self._fn, self._device, self.n_items = self._lib.cppf_lm_batch_launch, dev, len(items)

    def launch(self) -> None:
        rc = self._fn(self._h, torch.cuda.current_stream(self._device).cuda_stream)
        if rc:
            _hip.check(rc)

    def launch_on(self, stream: "torch.cuda.Stream") -> None:
        rc = self._fn(self._h, stream.cuda_stream)
        if rc:
            _hip.check(rc)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                self._lib.cppf_lm_batch_destroy(h)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass


def get_robot(name: str) -> Robot:
    """Stands in for jrl.robots.get_robot (cppflow/data_type_utils.py:197)."""
    if name not in ROBOT_SPECS:
        raise ValueError(f"unknown robot '{name}' (have {sorted(ROBOT_SPECS)})")
    return Robot(ROBOT_SPECS[name]())


def Panda() -> Robot:
    return get_robot("panda")


def Fetch() -> Robot:
    return get_robot("fetch")


def FetchArm() -> Robot:
    return get_robot("fetch_arm")
