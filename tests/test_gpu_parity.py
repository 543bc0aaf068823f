"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Every call goes through the C ABI of libcppflow_hip.so.

Bars (BASELINE.md section 5 / SURVEY.md 8d):
  * FK, capsule distances, collision / joint-limit masks, search cost: BIT-EXACT against the canonical-order fp32 oracle
  * Jacobian, pose error: atol 1e-5 against the fp64 oracle (reference's own J / r tolerance, tests/optimization_test.py:96-98)
  * x after one LM step: atol 5e-3 against the fp64 oracle in reference order (tests/optimization_test.py:99), on rows
    whose damped system is not near-singular (the reference's own fp32 LU is noise there)
  * final pose error after K steps: within 1e-5 (m, rad) of the fp64 oracle's
"""

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

ROBOTS = ["panda", "fetch", "fetch_arm", "chain12"]
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE, lm_hyper_parameters.py:119-126


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda:0")


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def robots():
    from cppflow_amd.robots import get_robot

    return {n: get_robot(n) for n in ROBOTS}


@pytest.mark.parametrize("name", ROBOTS)
def test_fk_bit_exact_vs_fp32_oracle(robots, name):
    q = H.random_configs(name, 4096, seed=1)
    got = host(robots[name].forward_kinematics(dev(q)))
    want = H.oracle32(name).fk(q)
    assert np.array_equal(got[:, :3], want[:, :3]), np.abs(got[:, :3] - want[:, :3]).max()
    assert np.array_equal(got[:, 3:], want[:, 3:]), np.abs(got[:, 3:] - want[:, 3:]).max()
    truth = H.oracle64(name).fk(q)
    assert np.abs(got[:, :3] - truth[:, :3]).max() < 2e-6


@pytest.mark.parametrize("name", ROBOTS)
def test_jacobian_and_pose_error(robots, name):
    S, W = 8, 64
    x0, target = H.lm_problem(name, S, W, seed=2)
    J = host(robots[name].jacobian(dev(x0)))
    assert np.abs(J - H.oracle64(name).jacobian(x0)).max() < 1e-5
    e, cur = robots[name].pose_errors(dev(x0), dev(target))
    e_want, cur_want = H.oracle64(name).pose_errors(x0, H.stacked(target, S))
    assert np.abs(host(e)[:, :, 0] - e_want).max() < 1e-5
    assert np.abs(host(cur)[:, :3] - cur_want[:, :3]).max() < 2e-6


@pytest.mark.parametrize("name", ROBOTS)
def test_single_lm_step_matches_reference_order_oracle(robots, name):
    """levenberg_marquardt_only_pose (no clamp), return_residual=True: x_new, scaled J, scaled e."""
    S, W = 16, 64
    x0, target = H.lm_problem(name, S, W, seed=3)
    res = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, return_residual=True, **LM)
    xo, Jo, eo, fails = H.oracle64(name).lm_step(x0, H.stacked(target, S), solver=0, **LM)
    assert fails == 0
    assert np.abs(host(res["J"]) - Jo).max() < 1e-5
    assert np.abs(host(res["e"])[:, :, 0] - eo).max() < 1e-5
    # x parity on rows whose damped system is not near-singular: sigma_min(J_scaled) >= 2e-2, i.e.
    # cond(J J^T + lambda I) <~ 6e4.  Below that the reference's own fp32 LU result is noise (SURVEY.md fact 0.5:
    # 0.02-0.04 rad median null-space error), so there is nothing to be in parity with.
    smin = np.linalg.svd(Jo, compute_uv=False)[:, -1]
    ok = smin >= 2e-2
    assert ok.mean() > 0.85, ok.mean()
    diff = np.abs(host(res["x"]) - xo)
    assert diff[ok].max() < 5e-3, diff[ok].max()
    assert np.isfinite(host(res["x"])).all()


@pytest.mark.parametrize("name", ROBOTS)
def test_fused_k_steps_final_pose_error(robots, name):
    S, W, K = 16, 64, 10
    x0, target = H.lm_problem(name, S, W, seed=4)
    res = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=K, clamp=True, want_errors=True, **LM)
    x_gpu = host(res["x"])
    o = H.oracle64(name)
    x_orc = o.lm_steps(x0, H.stacked(target, S), K, solver=0, **LM)
    pe_o, re_o = o.pose_metrics_exact(x_orc, H.stacked(target, S))
    # the metrics the kernel reports are those of its own x (checked against the oracle evaluated at that x) ...
    pe_at, re_at = o.pose_metrics_exact(x_gpu, H.stacked(target, S))
    assert np.abs(host(res["pos_err_m"]) - pe_at).max() < 1e-5
    assert np.abs(host(res["rot_err_rad"]) - re_at).max() < 1e-5
    # (the reference-order formula 2*acos(clamp(dot)) agrees with that to the formula's own sensitivity to the fp32 norm
    #  of the target quaternion: 2e-7 / sin(theta/2) <= 4.5e-4 rad above the clamp floor)
    _, re_ref = o.pose_metrics(x_gpu, H.stacked(target, S))
    assert np.abs(host(res["rot_err_rad"]) - re_ref).max() < 4.5e-4
    # ... and on rows where the oracle converged, the build converged to the same pose error within 1e-5
    conv = (pe_o < 1e-4) & (re_o < 1.2e-3)
    assert conv.mean() > 0.9, conv.mean()
    # rows at the fp32 floor after K steps: within 1e-5; rows between 5e-6 and 1e-4 are still contracting by a factor per
    # step, so two arithmetics (fp64 oracle / fp32 kernel, either kernel shape) differ there in proportion: factor bound
    settled = conv & (pe_o < 5e-6)
    assert settled.sum() > 0.8 * conv.sum()
    assert np.abs(host(res["pos_err_m"]) - pe_o)[settled].max() < 1e-5
    assert np.abs(host(res["rot_err_rad"]) - re_o)[settled].max() < 1e-5
    assert (host(res["pos_err_m"])[conv] <= 3.0 * pe_o[conv] + 1e-5).all()
    # joint limits hold exactly
    ch = H.chain(name)
    assert (x_gpu >= ch.lo - 0).all() and (x_gpu <= ch.hi + 0).all()


def _obstacles_for(name):
    if name in ("panda", "chain12"):
        return H.PANDA_2CUBES
    return [H.cuboid_obstacle(0.7, 0.1, 0.8, 0.3, 0.3, 0.3)]


@pytest.mark.parametrize("name", ROBOTS)
def test_collision_masks_bit_exact(robots, name):
    rb = robots[name]
    S, W = 32, 64
    q = H.random_configs(name, S * W, seed=5)
    obs = _obstacles_for(name)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)  # search.py:20-21
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    jl_lo, jl_hi = rb.padded_joint_limits()
    want = H.oracle32(name).masks(q, lo, hi, jl_lo, jl_hi)
    got = rb.collision_masks(dev(q).reshape(S, W, -1), want_min_dists=True)
    for k in ("self_mask", "env_mask", "jlim_mask"):
        assert np.array_equal(got[k].cpu().numpy().reshape(-1).astype(np.uint8), want[k]), k
    assert np.array_equal(host(got["ext_cost"]).reshape(-1), want["ext_cost"])
    assert np.array_equal(host(got["min_self"]).reshape(-1), want["min_self"])
    assert np.array_equal(host(got["min_env"]).reshape(-1), want["min_env"])
    # the masks are not trivially all-0 / all-1
    assert 0.01 < want["self_mask"].mean() < 0.99
    # fp64 truth: identical masks away from |dist| < 1e-5
    truth = H.oracle64(name).masks(q, lo, hi, jl_lo, jl_hi)
    far = np.abs(truth["min_self"]) > 1e-5
    assert np.array_equal(want["self_mask"][far], truth["self_mask"][far])
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_distance_matrices_bit_exact(robots, name):
    rb = robots[name]
    q = H.random_configs(name, 2048, seed=6)
    got = host(rb.self_collision_distances(dev(q)))
    assert np.array_equal(got, H.oracle32(name).self_dists(q))
    assert np.abs(got - H.oracle64(name).self_dists(q)).max() < 5e-6
    cuboid, T = H.PANDA_1CUBE[0]
    lo, hi = H.box_corners([cuboid], [T])
    got = host(rb.env_collision_distances(dev(q), torch.tensor(cuboid), torch.tensor(T)))
    assert np.array_equal(got, H.oracle32(name).env_dists(q, lo[0], hi[0]))
    assert np.abs(got - H.oracle64(name).env_dists(q, lo[0], hi[0])).max() < 5e-6


def test_fused_masks_equal_standalone_masks(robots):
    """The fused launch's masks are those of its own x_out (property P3 of SURVEY.md section 4, restated)."""
    rb = robots["panda"]
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    S, W = 16, 64
    x0, target = H.lm_problem("panda", S, W, seed=7)
    res = rb.lm_pose_steps(dev(x0), dev(target), n_steps=5, want_errors=True, want_collisions=True, want_min_dists=True, **LM)
    alone = rb.collision_masks(res["x"].reshape(S, W, -1), want_min_dists=True)
    for k in ("self_mask", "env_mask", "jlim_mask"):
        assert torch.equal(res[k].view(torch.bool), alone[k].reshape(-1)), k
    for k in ("ext_cost", "min_self", "min_env"):
        assert torch.equal(res[k], alone[k].reshape(-1)), k
    # and bit-exact against the fp32 oracle evaluated at the same x
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    jl_lo, jl_hi = rb.padded_joint_limits()
    want = H.oracle32("panda").masks(host(res["x"]), lo, hi, jl_lo, jl_hi)
    assert np.array_equal(res["self_mask"].cpu().numpy(), want["self_mask"])
    assert np.array_equal(res["env_mask"].cpu().numpy(), want["env_mask"])
    assert np.array_equal(host(res["ext_cost"]), want["ext_cost"])
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["fetch", "panda"])
def test_seed_validity(robots, name):
    S, W = 12, 59
    x0, target = H.lm_problem(name, S, W, seed=8, noise=0.02)
    got = host(robots[name].seed_validity(dev(x0), dev(target)))
    want = H.oracle64(name).seed_validity(x0, H.stacked(target, S), S, W)
    assert np.abs(got[:, 0] - want[:, 0]).max() < 1e-3  # cm
    assert np.abs(got[:, 1] - want[:, 1]).max() < 1e-3  # deg
    assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-3  # deg
    assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-4  # cm


def test_clamp_in_place_and_edge_sizes(robots):
    rb = robots["fetch"]
    ch = H.chain("fetch")
    x = dev(np.random.RandomState(0).uniform(-5, 5, size=(1000, 8)))
    x_before = host(x)
    ptr = x.data_ptr()
    y = rb.clamp_to_joint_limits(x)
    assert y.data_ptr() == ptr
    assert np.array_equal(host(x), H.oracle32("fetch").clamp(x_before))
    assert (host(x) >= ch.lo).all() and (host(x) <= ch.hi).all()
    # empty and ragged (n not a multiple of the workgroup) inputs
    assert rb.forward_kinematics(torch.empty((0, 8), device="cuda:0")).shape == (0, 7)
    for n in (1, 63, 257):
        q = H.random_configs("fetch", n, seed=n)
        assert np.array_equal(host(rb.forward_kinematics(dev(q))), H.oracle32("fetch").fk(q))


def test_contract_violations_raise(robots):
    rb = robots["panda"]
    with pytest.raises(RuntimeError):
        rb.forward_kinematics(torch.zeros((4, 7)))  # CPU tensor: no fallback
    with pytest.raises(AssertionError):
        rb.forward_kinematics(torch.zeros((4, 8), device="cuda:0"))
    with pytest.raises(AssertionError):
        rb.lm_pose_steps(torch.zeros((10, 7), device="cuda:0"), torch.zeros((3, 7), device="cuda:0"), 1e-6, 3.5, 0.35)
    cuboid, T = H.cuboid_obstacle(0, 0, 0, 1, 1, 1)
    T = T.copy()
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = 0.0, -1.0, 1.0, 0.0  # rotated cuboid: rejected like data_type_utils.py:108
    with pytest.raises(AssertionError):
        rb.set_obstacles([cuboid], [T])
    rb.set_obstacles([], [])


@pytest.mark.parametrize("name", ROBOTS)
def test_specialised_kernels_equal_generic_kernels(robots, name):
    """The robot-specialised instantiations (compile-time chain tables, capsules in registers) and the generic ones
    (kernel-argument constants, capsules in LDS) follow the same canonical operation order: identical outputs."""
    from cppflow_amd import _hip

    rb = robots[name]
    assert _hip.lib().cppf_robot_specialization(rb._handle(torch.device("cuda:0"))) >= 0
    obs = _obstacles_for(name)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    S, W = 8, 64
    x0, target = H.lm_problem(name, S, W, seed=11)
    kw = dict(n_steps=4, want_errors=True, want_collisions=True, want_min_dists=True, return_residual=True, **LM)
    a = rb.lm_pose_steps(dev(x0), dev(target), **kw)
    ca = rb.collision_masks(dev(x0).reshape(S, W, -1), want_min_dists=True)
    try:
        rb.debug_set("force_generic", 1)
        b = rb.lm_pose_steps(dev(x0), dev(target), **kw)
        cb = rb.collision_masks(dev(x0).reshape(S, W, -1), want_min_dists=True)
    finally:
        rb.debug_set("force_generic", 0)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for k in ca:
        assert torch.equal(ca[k], cb[k]), k
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("ndof,seed", [(3, 4), (4, 5), (5, 6), (6, 0), (7, 1), (8, 2), (9, 7), (10, 8), (12, 3)])
def test_arbitrary_chains_through_the_generic_kernels(ndof, seed):
    """Descriptions that match no generated table (random chains with general joint axes, rotated fixed transforms, a
    prismatic joint, fixed joints inside the chain) run the generic kernels: FK / masks bit-exact with the fp32 oracle,
    LM parity as for the shipped robots."""
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    spec = H.random_chain_spec(ndof, seed)
    rb = Robot(spec, specialize=False)  # (the run-time-specialised kernels: tests/test_gpu_round2.py)
    ch = canonicalize(spec)
    o64, o32 = Oracle(ch, f32=False), Oracle(ch, f32=True)
    assert _hip.lib().cppf_robot_specialization(rb._handle(torch.device("cuda:0"))) == -1
    rng = np.random.RandomState(seed)
    q = H.f32(rng.uniform(ch.lo, ch.hi, size=(2048, ndof)))
    assert np.array_equal(host(rb.forward_kinematics(dev(q))), o32.fk(q))
    assert np.abs(host(rb.jacobian(dev(q))) - o64.jacobian(q)).max() < 1e-5
    obs = [H.cuboid_obstacle(0.1, 0.1, 0.5, 0.3, 0.3, 0.3)]
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    jl_lo, jl_hi = rb.padded_joint_limits()
    want = o32.masks(q, lo, hi, jl_lo, jl_hi)
    for want_min in (False, True):
        got = rb.collision_masks(dev(q).reshape(32, 64, ndof), want_min_dists=want_min)
        for k in ("self_mask", "env_mask", "jlim_mask"):
            assert np.array_equal(got[k].cpu().numpy().reshape(-1).astype(np.uint8), want[k]), (k, want_min)
        assert np.array_equal(host(got["ext_cost"]).reshape(-1), want["ext_cost"])
    assert np.array_equal(host(got["min_self"]).reshape(-1), want["min_self"])
    assert np.array_equal(host(got["min_env"]).reshape(-1), want["min_env"])
    assert want["self_mask"].any() or want["env_mask"].any()
    # LM: K fused steps converge like the oracle's
    S, W, K = 8, 64, 10
    q_star = H.f32(rng.uniform(ch.lo, ch.hi, size=(W, ndof)))
    target = H.f32(o64.fk(q_star))
    x0 = H.f32(np.clip(q_star[None] + 0.1 * rng.randn(S, W, ndof), ch.lo, ch.hi).reshape(S * W, ndof))
    r = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, **LM)
    x_o = o64.lm_steps(x0, H.stacked(target, S), K)
    pe_o, re_o = o64.pose_metrics_exact(x_o, H.stacked(target, S))
    conv = (pe_o < 1e-4) & (re_o < 1.2e-3)
    assert conv.mean() > 0.8
    # Rows the oracle converges on end within 1e-5 of its pose error.  The basins of attraction of the LM iteration are
    # fractal: from a 0.1 rad perturbation a 1-ulp difference in an early residual can send an isolated row to a different
    # solution branch (in either implementation), so a stray row in 200 is tolerated; it must still be a valid result.
    d_pos = np.abs(host(r["pos_err_m"]) - pe_o)[conv]
    d_rot = np.abs(host(r["rot_err_rad"]) - re_o)[conv]
    same = (d_pos < 1e-5) & (d_rot < 1e-5)
    assert same.mean() >= 0.995, (same.mean(), d_pos.max(), d_rot.max())
    assert np.isfinite(host(r["x"])).all()
