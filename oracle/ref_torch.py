"""torch restatement of the reference's CPU path (TEST INFRASTRUCTURE ONLY; see oracle/lmik_oracle.c for the rules).

This is the "reference-equivalent CPU path (restated; jrl unavailable)" of BASELINE.md section 3: it issues the SAME
torch op sequence as the reference --

    levenberg_marquardt_only_pose   cppflow/optimization.py:73-92    (in-place row scaling, bmm x2, eye.repeat, + lambda I,
                                                                      torch.linalg.solve, x + delta)
    get_6d_pose_errors              cppflow/optimization_utils.py:802-820
    clamp_to_joint_limits           cppflow/optimization_utils.py:831-833 (python loop over joints)
    qpaths_batched_*_collisions     cppflow/collision_detection.py:27-69  (distance tensor -> min -> "< 0", OR over obstacles)
    joint_limit_almost_violations_3d / q_costs_external  cppflow/search.py:25-52, 146-150

-- with `jrl`'s FK / Jacobian / capsule distances / quaternion helpers written the way a batched-torch kinematics library
does them: a chain of batched 4x4 matmuls straight from the URDF-style description (xyz, rpy, axis), NOT from the
canonical chain the kernels and the C oracle use.  It therefore doubles as an independent check of the canonical rewrite.
It is what bench.py times as `cpu_baseline` (kind "port").
"""

import math
from typing import List, Optional, Tuple

import torch

from cppflow_amd.robot_model import RobotSpec, rpy_to_matrix


def _rotation_about_axis(axis: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
    """Rodrigues: [n] angles about one unit axis [3] -> [n,3,3]."""
    K = torch.tensor(
        [[0.0, -axis[2], axis[1]], [axis[2], 0.0, -axis[0]], [-axis[1], axis[0], 0.0]], dtype=angle.dtype, device=angle.device
    )
    I = torch.eye(3, dtype=angle.dtype, device=angle.device)
    s, c = torch.sin(angle)[:, None, None], torch.cos(angle)[:, None, None]
    return I[None] + s * K[None] + (1 - c) * (K @ K)[None]


def rotation_matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    """[n,3,3] -> [n,4] w-first; the candidate with the largest denominator is used (largest component positive)."""
    m00, m01, m02 = m[:, 0, 0], m[:, 0, 1], m[:, 0, 2]
    m10, m11, m12 = m[:, 1, 0], m[:, 1, 1], m[:, 1, 2]
    m20, m21, m22 = m[:, 2, 0], m[:, 2, 1], m[:, 2, 2]
    qa = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=1)
    d = torch.sqrt(torch.clamp(qa, min=0.0))
    cand = torch.stack(
        [
            torch.stack([d[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=1),
            torch.stack([m21 - m12, d[:, 1] ** 2, m10 + m01, m02 + m20], dim=1),
            torch.stack([m02 - m20, m10 + m01, d[:, 2] ** 2, m12 + m21], dim=1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, d[:, 3] ** 2], dim=1),
        ],
        dim=1,
    )  # [n, 4 candidates, 4]
    best = torch.argmax(qa, dim=1)
    idx = torch.arange(m.shape[0], device=m.device)
    return cand[idx, best] / (2.0 * d[idx, best])[:, None]


def quaternion_inverse(q: torch.Tensor) -> torch.Tensor:
    return q * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=q.dtype, device=q.device)


def quaternion_product(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(dim=1)
    bw, bx, by, bz = b.unbind(dim=1)
    return torch.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        dim=1,
    )


def quaternion_to_rpy(q: torch.Tensor) -> torch.Tensor:
    q0, q1, q2, q3 = q.unbind(dim=1)
    roll = torch.atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2))
    pitch = torch.asin(torch.clamp(2 * (q0 * q2 - q3 * q1), -1.0, 1.0))
    yaw = torch.atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))
    return torch.stack([roll, pitch, yaw], dim=1)


def geodesic_distance_between_quaternions(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """Formula quoted at cppflow/data_types.py:408-411, folded to [0, pi]."""
    eps = 1e-7
    dot = torch.clip(torch.sum(q1 * q2, dim=1), -1, 1)
    dist = 2 * torch.acos(torch.clamp(dot, -1 + eps, 1 - eps))
    return torch.abs(torch.remainder(dist + math.pi, 2 * math.pi) - math.pi)


class TorchRobot:
    """Batched-torch kinematics straight from the URDF-style spec (the stand-in for jrl.robot.Robot on the CPU)."""

    def __init__(self, spec: RobotSpec, device: str = "cpu", dtype: torch.dtype = torch.float32):
        self.spec, self.device, self.dtype = spec, device, dtype
        self.name = spec.name
        self._joints = []  # (fixed 4x4, axis or None, type)
        limits = []
        link_of_joint = {}
        moving = -1
        self._link_parent_moving = {spec.base_link: (-1, torch.eye(4, dtype=dtype, device=device))}
        pending = torch.eye(4, dtype=torch.float64)
        for j in spec.joints:
            T = torch.eye(4, dtype=torch.float64)
            T[:3, :3] = torch.tensor(rpy_to_matrix(*j.rpy))
            T[:3, 3] = torch.tensor(j.xyz, dtype=torch.float64)
            if j.jtype == "fixed":
                pending = pending @ T
                self._link_parent_moving[j.child_link] = (moving, pending.to(dtype).to(device))
                continue
            axis = torch.tensor(j.axis, dtype=torch.float64)
            axis = axis / axis.norm()
            self._joints.append(((pending @ T).to(dtype).to(device), axis.to(dtype).to(device), j.jtype))
            limits.append(j.limits)
            moving += 1
            pending = torch.eye(4, dtype=torch.float64)
            self._link_parent_moving[j.child_link] = (moving, torch.eye(4, dtype=dtype, device=device))
            link_of_joint[moving] = j.child_link
        self._ee_fixed = pending.to(dtype).to(device)
        self.ndof = len(self._joints)
        self.actuated_joints_limits: List[Tuple[float, float]] = [(float(l), float(u)) for l, u in limits]
        self.revolute_joint_idxs = [i for i, (_, _, t) in enumerate(self._joints) if t == "revolute"]
        self.prismatic_joint_idxs = [i for i, (_, _, t) in enumerate(self._joints) if t == "prismatic"]
        # capsules: (moving link, p0, p1, r) in the moving link's frame
        self._caps = []
        for c in spec.capsules:
            li, Tl = self._link_parent_moving[c.link]
            p0 = Tl[:3, :3] @ torch.tensor(c.p0, dtype=dtype, device=device) + Tl[:3, 3]
            p1 = Tl[:3, :3] @ torch.tensor(c.p1, dtype=dtype, device=device) + Tl[:3, 3]
            self._caps.append((li, p0, p1, float(c.radius)))
        from cppflow_amd.robot_model import canonicalize

        self.pairs = [tuple(int(v) for v in p) for p in canonicalize(spec).pairs]

    # -- kinematics -------------------------------------------------------------------------------------------------
    def _chain(self, x: torch.Tensor):
        n = x.shape[0]
        T = torch.eye(4, dtype=x.dtype, device=x.device)[None].repeat(n, 1, 1)
        axes, origins, links = [], [], []
        for i, (Tf, axis, jt) in enumerate(self._joints):
            T = torch.bmm(T, Tf[None].expand(n, 4, 4))
            axes.append(T[:, :3, :3] @ axis)
            origins.append(T[:, :3, 3])
            M = torch.eye(4, dtype=x.dtype, device=x.device)[None].repeat(n, 1, 1)
            if jt == "revolute":
                M[:, :3, :3] = _rotation_about_axis(axis, x[:, i])
            else:
                M[:, :3, 3] = axis[None] * x[:, i : i + 1]
            T = torch.bmm(T, M)
            links.append(T)
        T_ee = torch.bmm(T, self._ee_fixed[None].expand(n, 4, 4))
        return T_ee, axes, origins, links

    def forward_kinematics(self, x: torch.Tensor, out_device=None, dtype=None) -> torch.Tensor:
        T_ee, _, _, _ = self._chain(x)
        return torch.cat([T_ee[:, :3, 3], rotation_matrix_to_quaternion(T_ee[:, :3, :3])], dim=1)

    def jacobian(self, x: torch.Tensor) -> torch.Tensor:
        T_ee, axes, origins, _ = self._chain(x)
        n = x.shape[0]
        J = torch.zeros((n, 6, self.ndof), dtype=x.dtype, device=x.device)
        p_ee = T_ee[:, :3, 3]
        for i, (_, _, jt) in enumerate(self._joints):
            if jt == "revolute":
                J[:, 0:3, i] = axes[i]
                J[:, 3:6, i] = torch.cross(axes[i], p_ee - origins[i], dim=1)
            else:
                J[:, 3:6, i] = axes[i]
        return J

    def split_configs_to_revolute_and_prismatic(self, x):
        return x[:, self.revolute_joint_idxs], x[:, self.prismatic_joint_idxs]

    # -- capsule distances ------------------------------------------------------------------------------------------------
    def _capsule_endpoints(self, x: torch.Tensor):
        _, _, _, links = self._chain(x)
        n = x.shape[0]
        out = []
        for li, p0, p1, r in self._caps:
            if li < 0:
                out.append((p0[None].expand(n, 3), p1[None].expand(n, 3), r))
            else:
                R, t = links[li][:, :3, :3], links[li][:, :3, 3]
                out.append((R @ p0 + t, R @ p1 + t, r))
        return out

    def self_collision_distances(self, x: torch.Tensor) -> torch.Tensor:
        caps = self._capsule_endpoints(x)
        cols = []
        for a, b in self.pairs:
            d = _segment_segment_distance(caps[a][0], caps[a][1], caps[b][0], caps[b][1])
            cols.append(d - (caps[a][2] + caps[b][2]))
        return torch.stack(cols, dim=1)

    def env_collision_distances(self, x: torch.Tensor, cuboid: torch.Tensor, Tcuboid: torch.Tensor) -> torch.Tensor:
        caps = self._capsule_endpoints(x)
        lo = Tcuboid[:3, 3] + cuboid[:3]
        hi = Tcuboid[:3, 3] + cuboid[3:]
        return torch.stack([_segment_box_distance(p0, p1, lo, hi) - r for p0, p1, r in caps], dim=1)


def _segment_segment_distance(P1, Q1, P2, Q2):
    d1, d2, r = Q1 - P1, Q2 - P2, P1 - P2
    a, e, f = (d1 * d1).sum(1), (d2 * d2).sum(1), (d2 * r).sum(1)
    c, b = (d1 * r).sum(1), (d1 * d2).sum(1)
    denom = a * e - b * b
    s = torch.where(denom > 0, torch.clamp((b * f - c * e) / torch.where(denom > 0, denom, torch.ones_like(denom)), 0, 1), torch.zeros_like(denom))
    t = (b * s + f) / e
    s = torch.where(t < 0, torch.clamp(-c / a, 0, 1), torch.where(t > 1, torch.clamp((b - c) / a, 0, 1), s))
    t = torch.clamp(t, 0, 1)
    diff = (P1 + d1 * s[:, None]) - (P2 + d2 * t[:, None])
    return diff.norm(dim=1)


def _segment_box_distance(P0, P1, lo, hi):
    D = P1 - P0

    def g(t):
        x = P0 + D * t[:, None]
        ex = x - torch.minimum(torch.maximum(x, lo), hi)
        return (D * ex).sum(1)

    n = P0.shape[0]
    zeros, ones = torch.zeros(n, dtype=P0.dtype, device=P0.device), torch.ones(n, dtype=P0.dtype, device=P0.device)
    safe = torch.where(D != 0, D, torch.ones_like(D))
    inv = torch.where(D != 0, 1.0 / safe, torch.zeros_like(D))
    cands = [zeros, ones] + [torch.clamp((lo[i] - P0[:, i]) * inv[:, i], 0, 1) for i in range(3)] + [
        torch.clamp((hi[i] - P0[:, i]) * inv[:, i], 0, 1) for i in range(3)
    ]
    gv = [g(c) for c in cands]
    tl, gl, tr, gr = zeros.clone(), gv[0].clone(), ones.clone(), gv[1].clone()
    for k in range(2, 8):
        m = (gv[k] <= 0) & (cands[k] >= tl)
        tl, gl = torch.where(m, cands[k], tl), torch.where(m, gv[k], gl)
        m = (gv[k] >= 0) & (cands[k] <= tr)
        tr, gr = torch.where(m, cands[k], tr), torch.where(m, gv[k], gr)
    dg = gr - gl
    t_in = torch.where(dg > 0, tl + (tr - tl) * (-gl) / torch.where(dg > 0, dg, torch.ones_like(dg)), tl)
    t = torch.where(gv[0] >= 0, zeros, torch.where(gv[1] <= 0, ones, t_in))
    x = P0 + D * t[:, None]
    return (x - torch.minimum(torch.maximum(x, lo), hi)).norm(dim=1)


# ---- the reference's op sequences ----------------------------------------------------------------------------------------


def get_6d_pose_errors(robot: TorchRobot, x: torch.Tensor, target_poses: torch.Tensor):
    n = x.shape[0]
    current_poses = robot.forward_kinematics(x)
    pose_errors = torch.zeros((n, 6, 1), device=x.device, dtype=x.dtype)
    for i in range(3):
        pose_errors[:, i + 3, 0] = target_poses[:, i] - current_poses[:, i]
    rotation_error_quat = quaternion_product(target_poses[:, 3:], quaternion_inverse(current_poses[:, 3:7]))
    pose_errors[:, 0:3, 0] = quaternion_to_rpy(rotation_error_quat)
    return pose_errors, current_poses


def levenberg_marquardt_only_pose(robot: TorchRobot, x, target_path, lm_lambda, alpha_position, alpha_rotation, return_residual=False):
    n, ndof = x.shape
    error, _ = get_6d_pose_errors(robot, x, target_path)
    J_batch = robot.jacobian(x)
    error[:, 3:, 0] *= alpha_position
    error[:, :3, 0] *= alpha_rotation
    J_batch[:, 3:] *= alpha_position
    J_batch[:, :3] *= alpha_rotation
    J_batch_T = torch.transpose(J_batch, 1, 2)
    eye = torch.eye(ndof, device=x.device, dtype=x.dtype)[None, :, :].repeat(n, 1, 1)
    lhs_A = torch.bmm(J_batch_T, J_batch) + lm_lambda * eye
    rhs_B = torch.bmm(J_batch_T, error)
    delta_x = torch.linalg.solve(lhs_A, rhs_B)
    if return_residual:
        return x + torch.squeeze(delta_x, 2), J_batch, error
    return x + torch.squeeze(delta_x, 2)


def clamp_to_joint_limits(robot: TorchRobot, x: torch.Tensor) -> torch.Tensor:
    for i, (l, u) in enumerate(robot.actuated_joints_limits):
        x[:, i] = torch.clamp(x[:, i], l, u)
    return x


def lm_pose_steps(robot, x, target_stacked, n_steps, lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35):
    for _ in range(n_steps):
        x = clamp_to_joint_limits(robot, levenberg_marquardt_only_pose(robot, x, target_stacked, lm_lambda, alpha_position, alpha_rotation))
    return x


def calculate_pose_error_m_rad(robot, x, target_stacked):
    traced = robot.forward_kinematics(x)
    return torch.norm(target_stacked[:, :3] - traced[:, :3], dim=1), geodesic_distance_between_quaternions(target_stacked[:, 3:], traced[:, 3:])


def qpaths_batched_self_collisions(robot, q: torch.Tensor) -> torch.Tensor:
    k, n, ndof = q.shape
    dists = robot.self_collision_distances(q.reshape((k * n, ndof)))
    min_dists, _ = torch.min(dists, dim=1)
    return (min_dists < 0).reshape((k, n))


def qpaths_batched_env_collisions(robot, q: torch.Tensor, cuboids, Tcuboids) -> torch.Tensor:
    k, n, ndof = q.shape
    colliding = torch.zeros((k, n), dtype=torch.bool, device=q.device)
    q_2d = q.reshape((k * n, ndof))
    for cuboid, Tcuboid in zip(cuboids, Tcuboids):
        dists = robot.env_collision_distances(q_2d, cuboid, Tcuboid)
        min_dists, _ = torch.min(dists, dim=1)
        colliding = torch.logical_or(colliding, (min_dists < 0).reshape((k, n)))
    return colliding


def joint_limit_almost_violations_3d(robot, qs, eps_revolute, eps_prismatic):
    l_lim = torch.tensor([l for l, _ in robot.actuated_joints_limits], dtype=qs.dtype, device=qs.device)
    u_lim = torch.tensor([u for _, u in robot.actuated_joints_limits], dtype=qs.dtype, device=qs.device)
    l_lim[robot.prismatic_joint_idxs] += eps_prismatic
    l_lim[robot.revolute_joint_idxs] += eps_revolute
    u_lim[robot.prismatic_joint_idxs] -= eps_prismatic
    u_lim[robot.revolute_joint_idxs] -= eps_revolute
    return torch.logical_or((qs < l_lim).any(dim=2), (qs > u_lim).any(dim=2)).type(torch.float32)


def q_costs_external(robot, q, cuboids, Tcuboids, eps_revolute, eps_prismatic):
    jl = joint_limit_almost_violations_3d(robot, q, eps_revolute, eps_prismatic)
    env = qpaths_batched_env_collisions(robot, q, cuboids, Tcuboids)
    slf = qpaths_batched_self_collisions(robot, q)
    return 100 * jl + 1000 * env + 1000 * slf, jl, env, slf


def _get_mjacs(q: torch.Tensor, robot, prismatic_joint_scaling: float = 5.0) -> torch.Tensor:
    """[k, k, T-1]: max wrapped joint change from path a at t to path b at t+1 (cppflow/search.py:100-125)."""
    dqs = q[:, 1:, :].unsqueeze(1) - q[:, :-1, :].unsqueeze(0)
    if len(robot.prismatic_joint_idxs) > 0:
        dqs = dqs.clone()
        dqs[:, :, :, robot.prismatic_joint_idxs] *= prismatic_joint_scaling
    return torch.abs(torch.remainder(dqs + math.pi, 2 * math.pi) - math.pi).amax(dim=3)


def dp_search(robot, q: torch.Tensor, q_costs_external: torch.Tensor) -> torch.Tensor:
    """The torch recurrence of cppflow/search.py:145-173 given the external cost matrix -> best path [T, d]."""
    k, T, d = q.shape
    costs = torch.zeros((k, T), dtype=q.dtype)
    costs[:, 0] = q_costs_external[:, 0]
    mjacs = _get_mjacs(q, robot)
    memo = torch.zeros((k, T), dtype=torch.long)
    for t in range(1, T):
        nxt = torch.maximum(mjacs[:, :, t - 1], costs[:, t - 1].unsqueeze(0)) + q_costs_external[:, t].unsqueeze(1)
        costs[:, t], memo[:, t] = torch.min(nxt, dim=1)
    best = torch.zeros((T, d), dtype=q.dtype)
    i = torch.argmin(costs[:, -1])
    for t in range(T - 1, -1, -1):
        best[t] = q[i, t]
        i = memo[i, t]
    return best
