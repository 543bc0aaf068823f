"""Serial-chain robot descriptions and their canonical ("all joints move about local z") form.

The reference takes its robots from the un-vendored `jrl` package (`cppflow/planners.py:10`,
`cppflow/data_type_utils.py:197`: `get_robot(problem_dict["robot"])`).  Nothing of those models is in the
reference tree, so the models in `robots.py` are this build's own definitions, written in the
URDF vocabulary (`parent->joint xyz/rpy`, axis, type, limits) and turned here into the flat arrays the HIP library
and the oracle both consume.

Canonical form
--------------
For joint j with fixed parent->joint transform ``T_j`` and unit axis ``a_j`` we pick a rotation ``Q_j`` with
``Q_j @ e_z = a_j`` and rewrite the chain as

    world_T_link_j = prod_{i<=j}  F_i @ M_z(q_i)          F_i = Q_{i-1}^T @ T_i @ Q_i   (Q_{-1} = I)
    world_T_ee     = world_T_link_{d-1} @ F_ee            F_ee = Q_{d-1}^T @ T_fixed_tail

where ``M_z(q)`` is a rotation about / translation along local z.  The rewrite is exact; for principal axes ``Q``
is a signed permutation so the 0 / +-1 entries of ``F`` stay exact (they are snapped after the fp64 product), which
the robot-specialised kernels exploit.  Fixed joints are folded into the next ``F``; capsules and named frames are
re-expressed in the canonical frame of the moving link they are rigidly attached to (-1 = the base / world frame).
"""

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAX_DOF = 12
MAX_CAPSULES = 24
MAX_PAIRS = 128
MAX_OBSTACLES = 8

JOINT_REVOLUTE = 0
JOINT_PRISMATIC = 1


def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) @ Ry(pitch) @ Rx(roll)."""
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=np.float64)
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=np.float64)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def _snap(m: np.ndarray, tol: float = 1e-12) -> np.ndarray:
    """Snap entries within `tol` of 0 / +-1 so that exact structure survives the fp64 products."""
    m = m.copy()
    for v in (0.0, 1.0, -1.0):
        m[np.abs(m - v) < tol] = v
    return m


def _axis_to_z_rotation(axis: Sequence[float]) -> np.ndarray:
    """A rotation Q with Q @ e_z = axis.  Signed permutation for principal axes."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    principal = {
        (0, 0, 1): np.eye(3),
        (0, 0, -1): np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float64),
        (1, 0, 0): np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.float64),
        (-1, 0, 0): np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]], dtype=np.float64),
        (0, 1, 0): np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=np.float64),
        (0, -1, 0): np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64),
    }
    for k, q in principal.items():
        if np.allclose(a, k, atol=1e-12):
            assert np.isclose(np.linalg.det(q), 1.0) and np.allclose(q @ [0, 0, 1], k)
            return q
    # general axis: complete to a right-handed orthonormal basis
    helper = np.array([1.0, 0, 0]) if abs(a[0]) < 0.9 else np.array([0, 1.0, 0])
    x = np.cross(helper, a)
    x /= np.linalg.norm(x)
    y = np.cross(a, x)
    return np.stack([x, y, a], axis=1)


@dataclass
class JointSpec:
    """One URDF-style joint: fixed transform parent-link -> joint frame, then motion about `axis`."""

    name: str
    child_link: str
    xyz: Tuple[float, float, float]
    rpy: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    axis: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    jtype: str = "revolute"  # revolute | prismatic | fixed
    limits: Tuple[float, float] = (0.0, 0.0)


@dataclass
class CapsuleSpec:
    """Collision capsule rigidly attached to `link` (segment p0-p1 in that link's URDF frame, radius r)."""

    link: str
    p0: Tuple[float, float, float]
    p1: Tuple[float, float, float]
    radius: float


@dataclass
class RobotSpec:
    name: str
    formal_name: str
    base_link: str
    joints: List[JointSpec]
    capsules: List[CapsuleSpec] = field(default_factory=list)
    # capsule index pairs that are *checked*; None -> all pairs whose links are >= `min_link_gap` moving links apart
    collision_pairs: Optional[List[Tuple[int, int]]] = None
    min_link_gap: int = 2
    ignored_pairs: List[Tuple[int, int]] = field(default_factory=list)


@dataclass
class CanonicalChain:
    """Flat arrays (fp64 values, all exactly representable in fp32) shared by the HIP library and the oracle."""

    name: str
    ndof: int
    F: np.ndarray  # [d, 12]  rows of R (9) then t (3)
    F_ee: np.ndarray  # [12]
    jtype: np.ndarray  # [d] int32, 0 revolute / 1 prismatic
    lo: np.ndarray  # [d]
    hi: np.ndarray  # [d]
    cap_link: np.ndarray  # [L] int32, -1 = base
    cap_p0: np.ndarray  # [L, 3]
    cap_p1: np.ndarray  # [L, 3]
    cap_r: np.ndarray  # [L]
    pairs: np.ndarray  # [P, 2] int32 capsule indices
    frames: Dict[str, Tuple[int, np.ndarray]]  # link name -> (moving link idx, 4x4 local transform)
    cap_names: List[str]

    @property
    def n_capsules(self) -> int:
        return int(self.cap_link.shape[0])

    @property
    def n_pairs(self) -> int:
        return int(self.pairs.shape[0])


def _f32(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def _pack(T: np.ndarray) -> np.ndarray:
    return np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])


def canonicalize(spec: RobotSpec) -> CanonicalChain:
    """Rewrite `spec` into canonical form (module docstring) and round every constant to fp32."""
    frames: Dict[str, Tuple[int, np.ndarray]] = {spec.base_link: (-1, np.eye(4))}
    F_list, jt, lo, hi = [], [], [], []
    pending = np.eye(4)  # canonical moving-link frame -> current URDF link frame
    link_idx = -1
    for j in spec.joints:
        T = np.eye(4)
        T[:3, :3] = _snap(rpy_to_matrix(*j.rpy))
        T[:3, 3] = j.xyz
        if j.jtype == "fixed":
            pending = pending @ T
            frames[j.child_link] = (link_idx, pending.copy())
            continue
        assert j.jtype in ("revolute", "prismatic"), j.jtype
        Q = np.eye(4)
        Q[:3, :3] = _axis_to_z_rotation(j.axis)
        Fj = pending @ T @ Q
        Fj[:3, :3] = _snap(Fj[:3, :3])
        F_list.append(_pack(Fj))
        jt.append(JOINT_REVOLUTE if j.jtype == "revolute" else JOINT_PRISMATIC)
        lo.append(j.limits[0])
        hi.append(j.limits[1])
        link_idx += 1
        pending = np.eye(4)
        pending[:3, :3] = Q[:3, :3].T  # canonical frame of link_idx -> URDF frame of child link
        frames[j.child_link] = (link_idx, pending.copy())
    d = len(F_list)
    assert 1 <= d <= MAX_DOF, f"ndof {d} outside [1, {MAX_DOF}]"
    F_ee = pending.copy()
    F_ee[:3, :3] = _snap(F_ee[:3, :3])

    cap_link, cap_p0, cap_p1, cap_r, cap_names = [], [], [], [], []
    for c in spec.capsules:
        li, Tl = frames[c.link]
        cap_link.append(li)
        cap_p0.append(Tl[:3, :3] @ np.asarray(c.p0, dtype=np.float64) + Tl[:3, 3])
        cap_p1.append(Tl[:3, :3] @ np.asarray(c.p1, dtype=np.float64) + Tl[:3, 3])
        cap_r.append(c.radius)
        cap_names.append(c.link)
    L = len(cap_link)
    assert L <= MAX_CAPSULES

    if spec.collision_pairs is not None:
        pairs = [tuple(p) for p in spec.collision_pairs]
    else:
        ignored = {tuple(sorted(p)) for p in spec.ignored_pairs}
        pairs = [
            (a, b)
            for a in range(L)
            for b in range(a + 1, L)
            if abs(cap_link[a] - cap_link[b]) >= spec.min_link_gap and (a, b) not in ignored
        ]
    assert len(pairs) <= MAX_PAIRS, f"{len(pairs)} pairs > {MAX_PAIRS}"

    return CanonicalChain(
        name=spec.name,
        ndof=d,
        F=_f32(np.stack(F_list)),
        F_ee=_f32(_pack(F_ee)),
        jtype=np.asarray(jt, dtype=np.int32),
        lo=_f32(lo),
        hi=_f32(hi),
        cap_link=np.asarray(cap_link, dtype=np.int32).reshape(L),
        cap_p0=_f32(np.asarray(cap_p0, dtype=np.float64).reshape(L, 3)),
        cap_p1=_f32(np.asarray(cap_p1, dtype=np.float64).reshape(L, 3)),
        cap_r=_f32(np.asarray(cap_r, dtype=np.float64).reshape(L)),
        pairs=np.asarray(pairs, dtype=np.int32).reshape(len(pairs), 2),
        frames=frames,
        cap_names=cap_names,
    )


def urdf_forward_kinematics(spec: RobotSpec, q: np.ndarray, link: Optional[str] = None) -> np.ndarray:
    """Plain fp64 4x4-chain FK straight from the URDF-style spec (no canonical rewrite).

    Host-side helper used for problem loading (`data_type_utils.offset_target_path` evaluates a named frame at q = 0,
    reference `cppflow/data_type_utils.py:65-73`) and as an independent check of `canonicalize`.
    Returns the 4x4 world transform of `link` (default: the last link of the chain).
    """
    q = np.asarray(q, dtype=np.float64).reshape(-1)
    T = np.eye(4)
    if link == spec.base_link:
        return T
    qi = 0
    for j in spec.joints:
        Tj = np.eye(4)
        Tj[:3, :3] = rpy_to_matrix(*j.rpy)
        Tj[:3, 3] = j.xyz
        T = T @ Tj
        if j.jtype != "fixed":
            a = np.asarray(j.axis, dtype=np.float64)
            a = a / np.linalg.norm(a)
            M = np.eye(4)
            if j.jtype == "revolute":
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                M[:3, :3] = np.eye(3) + np.sin(q[qi]) * K + (1 - np.cos(q[qi])) * (K @ K)
            else:
                M[:3, 3] = a * q[qi]
            T = T @ M
            qi += 1
        if link is not None and j.child_link == link:
            return T
    assert link is None, f"link '{link}' not in chain"
    return T
