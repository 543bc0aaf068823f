"""Mask-only fused launches on INCOHERENT rows (pytest -m gpu).  The fused kernel's broad phase is wave-uniform: an exact capsule
test runs for the 64 rows of a wavefront as soon as one of them is within reach (csrc/kernels_collision.h: cull_far), so which exact
tests a row's wavefront executes depends on its neighbours -- the masks must not.  Bar: BIT-EXACT against the canonical-order fp32
oracle (reference: cppflow/collision_detection.py:39-68, search.py:46-54) on independent random configurations (a wavefront then runs
~26 exact tests per row where its rows need ~3, profiles/r5_cull_stats.txt), on consecutive waypoints of a smooth path (coherent),
on both within one launch, on partial wavefronts and tiny launches, for every shipped robot; and a row's masks are the same whatever
rows share its wavefront.  `exact_tests_in_wave` recomputes, from the oracle's capsule end points and the kernel's broad-phase thresholds,
how many exact tests a wavefront executes, so that the test knows which regime each input is in.  (Written with the round-5 experiment
that compacted the surviving (row, test) items into an LDS queue -- bit-exact, slower, not taken: profiles/r5_ab_coll_queue.txt.)"""

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def exact_tests_in_wave(name, x, lo, hi):
    """per wavefront of 64 consecutive rows: exact tests the wavefront executes (any row within reach) and (row, test) items its rows need"""
    from cppflow_amd import gen_robots as G

    ch = H.chain(name)
    ep = np.asarray(H.oracle32(name).capsule_endpoints(x))
    n = (x.shape[0] // 64) * 64
    _, _, _, _, half = G.capsule_centred(ch.cap_p0, ch.cap_p1)
    r32 = ch.cap_r.astype(np.float32)
    c = 0.5 * (ep[:n, :, :3].astype(np.float64) + ep[:n, :, 3:])
    tests = np.zeros(n // 64, dtype=np.int64)
    items = np.zeros(n // 64, dtype=np.int64)
    for a, b in ch.pairs:
        w = (((c[:, a] - c[:, b]) ** 2).sum(-1) <= float(G.cull_threshold(half[a] + half[b] + float(r32[a]) + float(r32[b])))).reshape(-1, 64)
        tests += w.any(1)
        items += w.sum(1)
    for o in range(len(lo)):
        for k in range(c.shape[1]):
            e = c[:, k] - np.clip(c[:, k], lo[o], hi[o])
            w = ((e**2).sum(-1) <= float(G.cull_threshold(half[k] + float(r32[k])))).reshape(-1, 64)
            tests += w.any(1)
            items += w.sum(1)
    return tests, items / 64.0


def setup_robot(name):
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    return rb, lo, hi


def fused_masks(rb, x, target, n_steps=1):
    res = rb.lm_pose_steps(dev(x), dev(target), n_steps=n_steps, want_collisions=True, **LM)  # masks only: the launch kind with a broad phase
    torch.cuda.synchronize()
    return res


def check_against_oracle(name, rb, res, lo, hi):
    jl_lo, jl_hi = rb.padded_joint_limits()
    x = res["x"].cpu().numpy().astype(np.float64)
    want = H.oracle32(name).masks(x, lo, hi, jl_lo, jl_hi)
    for k in ("self_mask", "env_mask", "jlim_mask"):
        assert np.array_equal(res[k].cpu().numpy().reshape(-1).astype(np.uint8), want[k]), (name, k)
    assert np.array_equal(res["ext_cost"].cpu().numpy().astype(np.float64), want["ext_cost"]), name
    return x, want


def restore(rb):
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["panda", "fetch", "fetch_arm", "chain12"])
def test_masks_bit_exact_on_independent_random_configurations(name):
    rb, lo, hi = setup_robot(name)
    n = 64 * 96
    _, target = H.lm_problem(name, 96, 64, seed=11)  # (targets of reachable poses; one step from a random configuration is a random configuration)
    q = H.random_configs(name, n, seed=12)
    res = fused_masks(rb, q, target)
    x, want = check_against_oracle(name, rb, res, lo, hi)
    tests, need = exact_tests_in_wave(name, x, lo, hi)
    if name == "panda":  # the incoherent regime (a compact arm like fetch_arm needs most of its pairs on every row anyway)
        assert np.median(tests) > 3 * np.median(need), (name, np.median(tests), np.median(need))
    assert np.median(tests) >= np.median(need)
    assert 0.005 < want["self_mask"].mean() < 0.995 or H.chain(name).pairs.shape[0] == 0
    # the standalone mask kernel at the same x
    alone = rb.collision_masks(res["x"].reshape(96, 64, -1))
    for k in ("self_mask", "env_mask", "jlim_mask"):
        assert torch.equal(res[k].view(torch.bool), alone[k].reshape(-1).view(torch.bool)), (name, k)
    restore(rb)


def smooth_paths(name, S, W, seed):
    """S seeds following ONE smooth joint-space path (consecutive waypoints a few millirad apart): coherent wavefronts"""
    ch = H.chain(name)
    rng = np.random.RandomState(seed)
    qa = rng.uniform(ch.lo + 0.3, ch.hi - 0.3)
    qb = np.clip(qa + rng.uniform(-0.5, 0.5, size=ch.ndof), ch.lo, ch.hi)
    path = qa[None] + (qb - qa)[None] * np.linspace(0.0, 1.0, W)[:, None]
    x = np.clip(path[None] + 0.01 * rng.randn(S, 1, ch.ndof), ch.lo, ch.hi).reshape(S * W, ch.ndof)
    return H.f32(x), H.f32(H.oracle64(name).fk(H.f32(path)))


def test_coherent_and_incoherent_wavefronts_in_one_launch():
    name = "panda"
    rb, lo, hi = setup_robot(name)
    S, W = 40, 64
    x_path, target = smooth_paths(name, S, W, seed=21)
    q = H.random_configs(name, S * W, seed=22)
    mix = x_path.reshape(S, W, -1).copy()
    mix[::2] = q.reshape(S, W, -1)[::2]  # every other wavefront: independent rows
    res = fused_masks(rb, mix.reshape(S * W, -1), target, n_steps=1)
    xk, want = check_against_oracle(name, rb, res, lo, hi)
    tests, need = exact_tests_in_wave(name, xk, lo, hi)
    assert np.median(tests[::2]) > 3 * np.median(need[::2]) and np.median(tests[1::2]) < 2 * np.median(need[1::2]) + 1, (tests[:4], need[:4])
    restore(rb)


def test_partial_last_wavefront_and_tiny_launches():
    name = "panda"
    rb, lo, hi = setup_robot(name)
    for n in (64 * 9 + 17, 64, 5, 64 * 3 + 63):
        q = H.random_configs(name, n, seed=30 + n)
        W = n  # one "seed" of n waypoints: any n is a legal launch
        _, target = H.lm_problem(name, 1, W, seed=31)
        res = fused_masks(rb, q, target)
        check_against_oracle(name, rb, res, lo, hi)
    restore(rb)


def test_masks_do_not_depend_on_the_wavefront_a_row_is_in():
    name = "panda"
    rb, lo, hi = setup_robot(name)
    S, W = 32, 64
    x0, target = H.lm_problem(name, S, W, seed=41)
    q = H.random_configs(name, S * W, seed=42)
    rows = np.arange(S * W) % 3 == 0
    x = np.where(rows[:, None], q, x0)  # every third row an independent configuration
    tgt = target
    res = fused_masks(rb, x, tgt, n_steps=1)
    # the same rows in another order (targets are per waypoint index: permute WITHIN the waypoint index, i.e. across seeds)
    rng = np.random.RandomState(5)
    perm = np.stack([rng.permutation(S) for _ in range(W)], axis=1)  # perm[s, w]: which seed's row sits at (s, w)
    xp = x.reshape(S, W, -1)[perm, np.arange(W)[None, :]].reshape(S * W, -1)
    resp = fused_masks(rb, xp, tgt, n_steps=1)
    for k in ("self_mask", "env_mask", "jlim_mask", "ext_cost", "x"):
        a = res[k].cpu().numpy().reshape(S, W, -1)[perm, np.arange(W)[None, :]].reshape(S * W, -1)
        assert np.array_equal(a, resp[k].cpu().numpy().reshape(S * W, -1)), k
    restore(rb)
