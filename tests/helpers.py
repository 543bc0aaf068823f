"""Shared test helpers: seeded inputs in the reference's own construction and oracle handles."""

import functools

import numpy as np

from cppflow_amd.robot_model import canonicalize
from cppflow_amd.robot_zoo import ROBOT_SPECS
from oracle.oracle import Oracle


@functools.lru_cache(maxsize=None)
def chain(name):
    return canonicalize(ROBOT_SPECS[name]())


@functools.lru_cache(maxsize=None)
def oracle64(name):
    return Oracle(chain(name), f32=False, threads=8)


@functools.lru_cache(maxsize=None)
def oracle32(name):
    return Oracle(chain(name), f32=True, threads=8)


def f32(a):
    """fp32-representable float64 array (what both oracle builds and the GPU see)."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def random_configs(name, n, seed=0, margin=0.0):
    ch = chain(name)
    rng = np.random.RandomState(seed)
    return f32(rng.uniform(ch.lo + margin, ch.hi - margin, size=(n, ch.ndof)))


def lm_problem(name, S, W, seed=0, noise=0.1):
    """Seeds exactly as tests/optimization_test.py:82,136-137 of the reference builds them:
    q* ~ U(limits), target = FK(q*), x0 = clamp(q* + 0.1 * randn).  target is [W,7] shared by the S seeds."""
    ch = chain(name)
    rng = np.random.RandomState(seed)
    q_star = rng.uniform(ch.lo, ch.hi, size=(W, ch.ndof))
    target = f32(oracle64(name).fk(f32(q_star)))
    x0 = np.tile(q_star[None], (S, 1, 1)) + noise * rng.randn(S, W, ch.ndof)
    x0 = np.clip(x0, ch.lo, ch.hi).reshape(S * W, ch.ndof)
    return f32(x0), target


def collision_free_problem(name, S, W, seed=0, noise=0.005, obstacles=None, bad_seeds=(), bad_noise=0.6):
    """Like `lm_problem`, but the waypoints' q* are drawn collision-free (self and against `obstacles`) and the seeds sit close to
    them, so that after a few LM steps most seeds are VALID (x_is_valid's seed selection has something to select); the seeds in
    `bad_seeds` get `bad_noise` instead and are still far from the target then."""
    ch = chain(name)
    rng = np.random.RandomState(seed)
    cand = f32(rng.uniform(ch.lo + 0.1, ch.hi - 0.1, size=(8 * W, ch.ndof)))
    lo, hi = box_corners([c for c, _ in obstacles], [T for _, T in obstacles]) if obstacles else (None, None)
    m = oracle32(name).masks(cand, lo, hi, ch.lo, ch.hi)
    keep = cand[(m["self_mask"] == 0) & (m["env_mask"] == 0)][:W]
    assert len(keep) == W
    target = f32(oracle64(name).fk(keep))
    sigma = np.full((S, 1, 1), noise)
    sigma[list(bad_seeds)] = bad_noise
    x0 = np.clip(keep[None] + sigma * rng.randn(S, W, ch.ndof), ch.lo, ch.hi).reshape(S * W, ch.ndof)
    return f32(x0), target


def stacked(target, S):
    return np.tile(target, (S, 1))


def box_corners(cuboids, Tcuboids):
    """World-frame corners exactly as the library forms them: fp32 t + fp32 extent."""
    lo, hi = [], []
    for c, T in zip(cuboids, Tcuboids):
        c = np.asarray(c, dtype=np.float32)
        t = np.asarray(T, dtype=np.float32)[:3, 3]
        lo.append((t + c[:3]).astype(np.float64))
        hi.append((t + c[3:]).astype(np.float64))
    return np.array(lo).reshape(-1, 3), np.array(hi).reshape(-1, 3)


def cuboid_obstacle(x, y, z, sx, sy, sz):
    """(cuboid[6], Tcuboid[4,4]) exactly as cppflow/data_type_utils.py:109-124 builds them (element [3,3] left 0)."""
    cuboid = np.array([-sx / 2, -sy / 2, -sz / 2, sx / 2, sy / 2, sz / 2], dtype=np.float32)
    T = np.zeros((4, 4), dtype=np.float32)
    T[:3, :3] = np.eye(3, dtype=np.float32)
    T[0, 3], T[1, 3], T[2, 3] = x, y, z
    return cuboid, T


PANDA_2CUBES = [cuboid_obstacle(0.2, 0.3, 0.4, 0.15, 0.15, 0.15), cuboid_obstacle(-0.25, 0.3, 0.75, 0.15, 0.15, 0.15)]
PANDA_1CUBE = [cuboid_obstacle(0.0, 0.2, 0.7, 0.25, 0.25, 0.25)]


def random_chain_spec(ndof, seed, n_prismatic=1, general_axes=True):
    """A random serial chain in the URDF vocabulary: arbitrary (non-principal) joint axes, random fixed transforms with
    rotations, fixed joints in between, a prismatic joint somewhere, capsules on every other link."""
    from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec

    rng = np.random.RandomState(seed)
    pris = set(rng.choice(ndof, size=n_prismatic, replace=False).tolist()) if n_prismatic else set()
    joints, capsules = [], [CapsuleSpec("base", (0, 0, 0), (0, 0, 0.08), 0.05)]
    for i in range(ndof):
        if i % 3 == 2:  # a fixed joint in the middle of the chain
            joints.append(JointSpec(f"fix{i}", f"flink{i}", tuple(rng.uniform(-0.05, 0.05, 3)), tuple(rng.uniform(-0.5, 0.5, 3)), jtype="fixed"))
        axis = rng.randn(3) if general_axes and i % 2 == 0 else np.eye(3)[rng.randint(3)] * rng.choice([-1.0, 1.0])
        xyz = tuple(rng.uniform(-0.15, 0.15, 3) + np.array([0, 0, 0.12]))
        rpy = tuple(rng.uniform(-1.0, 1.0, 3))
        if i in pris:
            joints.append(JointSpec(f"j{i}", f"link{i}", xyz, rpy, tuple(axis), "prismatic", (-0.1, 0.3)))
        else:
            joints.append(JointSpec(f"j{i}", f"link{i}", xyz, rpy, tuple(axis), "revolute", (-2.5, 2.0 + 0.1 * i)))
        if i % 2 == 1:
            capsules.append(CapsuleSpec(f"link{i}", tuple(rng.uniform(-0.03, 0.03, 3)), tuple(rng.uniform(0.05, 0.12, 3)), 0.03 + 0.01 * (i % 3)))
    joints.append(JointSpec("tool", "tool", (0.02, -0.01, 0.1), (0.3, -0.2, 0.5), jtype="fixed"))
    capsules.append(CapsuleSpec("tool", (0, 0, -0.05), (0, 0.02, 0.02), 0.025))
    return RobotSpec(f"random{ndof}_{seed}", f"random {ndof}-dof chain", "base", joints, capsules, min_link_gap=2)


# ---- hand-derived FK pins ---------------------------------------------------------------------------------
# jrl is absent, so no FK value of the reference can be imported; these are computed ON PAPER from the public URDF constants the
# models cite (franka_description / fetch_description) -- every joint frame of these configurations is a
# signed axis permutation, so the chain is a sum of link offsets along known world axes -- and written down as literals.  They pin
# the canonical rewrite (robot_model.py), the oracle's FK and the HIP FK to the URDF semantics (parent -> joint xyz / rpy, axis),
# independently of any code of this repository.  Pose = [x y z qw qx qy qz], base frame = the spec's base link.
#   Panda, q = 0:            z = 0.333 + 0.316 + 0.384 - 0.107 = 0.926, x = 0.0825 - 0.0825 + 0.088; hand = Rx(pi) Rz(-pi/4)
#   Panda, q4 = -pi/2, q6 = pi/2:  the forearm points along +x: x = 0.0825 + 0.384 + 0.088, z = 0.333 + 0.316 + 0.0825 - 0.107
#   Fetch, q = 0:            x = sum of the link offsets = 1.1281, z = 0.37743 + 0.34858 + 0.06
#   Fetch, torso 0.2, pan pi/2:   the arm points along +y: y = 1.09545, x = -0.086875 + 0.119525, z = 0.78601 + 0.2
#   Fetch, torso 0.2, pan pi/2, elbow pi/2:  the forearm points down: y = 0.117 + 0.219 + 0.133, z = 0.98601 - 0.62645
_C8, _S8 = 0.9238795325112867, 0.3826834323650898  # cos, sin of pi/8
FK_PINS = {
    "panda": [
        ([0, 0, 0, 0, 0, 0, 0], [0.088, 0.0, 0.926, 0.0, _C8, _S8, 0.0]),
        ([0, 0, 0, -np.pi / 2, 0, np.pi / 2, 0], [0.5545, 0.0, 0.6245, 0.0, _C8, _S8, 0.0]),
    ],
    "fetch": [
        ([0, 0, 0, 0, 0, 0, 0, 0], [1.1281, 0.0, 0.78601, 1.0, 0.0, 0.0, 0.0]),
        ([0.2, np.pi / 2, 0, 0, 0, 0, 0, 0], [0.03265, 1.09545, 0.98601, np.sqrt(0.5), 0.0, 0.0, np.sqrt(0.5)]),
        ([0.2, np.pi / 2, 0, 0, np.pi / 2, 0, 0, 0], [0.03265, 0.469, 0.35956, 0.5, -0.5, 0.5, 0.5]),
    ],
    "fetch_arm": [
        ([0, 0, 0, 0, 0, 0, 0], [1.1281, 0.0, 0.78601, 1.0, 0.0, 0.0, 0.0]),
        ([np.pi / 2, 0, 0, np.pi / 2, 0, 0, 0], [0.03265, 0.469, 0.15956, 0.5, -0.5, 0.5, 0.5]),
    ],
}


# ---- hand-derived Jacobian pins at the zero pose ----------------------------------------------------------------------------------
# Same derivation on paper: at q = 0 every joint axis is a signed world axis and every joint origin a sum of link offsets, so the
# geometric Jacobian column of a revolute joint is [axis; axis x (p_ee - origin)] and of a prismatic one [0; axis]  (rows 0:3 angular,
# 3:6 linear, world frame: SURVEY.md 8a a7).  Panda: axes z, y, z, -y, z, -y, -z at origins (0,0,.333), (0,0,.333), (0,0,.649),
# (.0825,0,.649), (0,0,1.033), (0,0,1.033), (.088,0,1.033), p_ee = (.088,0,.926).  Fetch: torso prismatic z, then z, y, x, y, x, y, x
# at x = .03265, .14965, .36865, .50165, .69865, .82315, .96165 (z = .72601 for the pan joint, .78601 beyond), p_ee = (1.1281,0,.78601).
J_PINS = {
    "panda": np.array([
        # j1      j2      j3      j4       j5      j6      j7
        [0.0,     0.0,    0.0,    0.0,     0.0,    0.0,    0.0],     # wx
        [0.0,     1.0,    0.0,   -1.0,     0.0,   -1.0,    0.0],     # wy
        [1.0,     0.0,    1.0,    0.0,     1.0,    0.0,   -1.0],     # wz
        [0.0,     0.593,  0.0,   -0.277,   0.0,    0.107,  0.0],     # vx
        [0.088,   0.0,    0.088,  0.0,     0.088,  0.0,    0.0],     # vy
        [0.0,    -0.088,  0.0,    0.0055,  0.0,    0.088,  0.0],     # vz
    ]),
    "fetch": np.array([
        # torso   pan      lift      uroll  elbow     froll  wflex     wroll
        [0.0,     0.0,     0.0,      1.0,   0.0,      1.0,   0.0,      1.0],
        [0.0,     0.0,     1.0,      0.0,   1.0,      0.0,   1.0,      0.0],
        [0.0,     1.0,     0.0,      0.0,   0.0,      0.0,   0.0,      0.0],
        [0.0,     0.0,     0.0,      0.0,   0.0,      0.0,   0.0,      0.0],
        [0.0,     1.09545, 0.0,      0.0,   0.0,      0.0,   0.0,      0.0],
        [1.0,     0.0,    -0.97845,  0.0,  -0.62645,  0.0,  -0.30495,  0.0],
    ]),
}
J_PINS["fetch_arm"] = J_PINS["fetch"][:, 1:].copy()


# ---- the one forward-kinematics datum of the reference tree ------------------------------------------------------------------------
# tests/planners_test.py:299-309 of the reference: a (commented-out) Panda configuration q0 next to the assertion that
# robot.forward_kinematics(q0) equals the first pose of its target path to atol = 1e-3 -- pose [0.45, 0, 0, 1, 0, 0, 0] + the
# panda__1cube offset [0, 0.5421984559194368, 0.7885155964931997] (`:282-298`).  jrl's Panda (`panda_link0 -> panda_hand`,
# ros2/ros2_publisher.py:60-61) evaluated at an arbitrary seven-joint configuration: the only number in the reference that
# depends on every link offset, every axis and the hand frame of the model at once.
REFERENCE_PANDA_Q0 = [1.267967, 0.711829, -0.811080, -0.810924, -2.637594, 1.767759, 0.083284]
REFERENCE_PANDA_POSE = [0.45, 0.5421984559194368, 0.7885155964931997, 1.0, 0.0, 0.0, 0.0]


def fk_pin_arrays(name):
    q = np.array([p[0] for p in FK_PINS[name]], dtype=np.float64)
    pose = np.array([p[1] for p in FK_PINS[name]], dtype=np.float64)
    return q, pose


def pose_close(got, want, tol_p, tol_q):
    """position within tol_p; quaternion equal up to sign within tol_q"""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    dq = np.minimum(np.abs(got[:, 3:] - want[:, 3:]).max(axis=1), np.abs(got[:, 3:] + want[:, 3:]).max(axis=1))
    return np.abs(got[:, :3] - want[:, :3]).max() <= tol_p and dq.max() <= tol_q


# ---- closed-form cases of the capsule distance functions ---------------------------------------------------------------------------
# (end points exactly representable in fp32; radius 0, so the "distance" the APIs return is the distance between the axis segments)
SEGMENT_KATS = [
    (((0, 0, 0), (1, 0, 0)), ((0, 1, 0), (1, 1, 0)), 1.0),            # parallel, overlapping ranges
    (((0, 0, 0), (1, 0, 0)), ((2, 1, 0), (3, 1, 0)), np.sqrt(2.0)),    # parallel, disjoint ranges: end point to end point
    (((-1, 0, 0), (1, 0, 0)), ((0, -1, 0.5), (0, 1, 0.5)), 0.5),      # skew, crossing in projection
    (((-1, 0, 0), (1, 0, 0)), ((0, -1, 0), (0, 1, 0)), 0.0),           # intersecting
    (((0, 0, 0), (1, 0, 0)), ((2, 0, 0), (4, 0, 0)), 1.0),             # collinear, disjoint
    (((0, 0, 0), (1, 0, 0)), ((0.5, 0.25, 0), (0.5, 2, 0)), 0.25),     # T shape: end point to interior
    (((0.375, 0.5, 0), (0.375, 0.5, 0)), ((0, 0, 0), (0, 0, 0)), 0.625),  # two spheres (zero-length segments)
    (((0, 0, 2), (0, 0, 2)), ((-1, 0, 0), (1, 0, 0)), 2.0),            # sphere to the interior of a segment
]
BOX_KATS = [  # against the unit box [0, 1]^3
    (((2, 0.5, 0.5), (3, 0.5, 0.5)), 1.0),               # along an axis, off a face
    (((2, 2, 0.5), (3, 3, 0.5)), np.sqrt(2.0)),          # off an edge: end point to the edge
    (((2, 2, 2), (2, 2, 2)), np.sqrt(3.0)),              # a sphere off a corner
    (((-1, 2, 0.5), (2, -1, 0.5)), 0.0),                 # passes through the box
    (((0.5, 0.5, 0.5), (0.625, 0.5, 0.5)), 0.0),         # inside
    (((-1, 1.5, 0.5), (2, 1.5, 0.5)), 0.5),              # parallel to a face, overhanging both sides
    (((1.5, 1.5, -1), (1.5, 1.5, 2)), np.sqrt(0.5)),     # parallel to an edge
    (((2, 0, 0.5), (0, 2, 0.5)), 0.0),                   # grazes the edge x = y = 1 exactly
]


def two_capsule_spec(c0, c1):
    """three coincident revolute joints about z, a capsule c0 on the base and c1 on the last link: at q = 0 both sit where given"""
    from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec

    joints = [JointSpec(f"j{i}", f"l{i}", (0, 0, 0), (0, 0, 0), (0, 0, 1), "revolute", (-1.0, 1.0)) for i in range(3)]
    joints.append(JointSpec("tool", "tool", (0, 0, 0.125), (0, 0, 0), jtype="fixed"))
    caps = [CapsuleSpec("base", c0[0], c0[1], 0.0), CapsuleSpec("l2", c1[0], c1[1], 0.0)]
    return RobotSpec("kat", "two capsules", "base", joints, caps, collision_pairs=[(0, 1)])


def two_capsule_chain(c0, c1):
    from cppflow_amd.robot_model import canonicalize

    return canonicalize(two_capsule_spec(c0, c1))
