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
