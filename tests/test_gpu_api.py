"""GPU tests of the reference-named Python surface (each call lands on the C ABI) and of the committed golden fixtures."""

import os

import numpy as np
import pytest
import torch

from cppflow_amd import _hip
from tests import helpers as H

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def robots():
    from cppflow_amd.robots import get_robot

    return {n: get_robot(n) for n in ("panda", "fetch", "fetch_arm", "chain12")}


@pytest.mark.parametrize("name", ["panda", "fetch", "fetch_arm", "chain12"])
def test_golden_vectors(robots, name):
    """tests/golden/lm_golden_<robot>.npz (fp64 oracle, reference operation order; made by tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, f"lm_golden_{name}.npz"))
    rb = robots[name]
    S, W, K = int(z["S"]), int(z["W"]), int(z["K"])
    x0, target = dev(z["x0"]), dev(z["target"])
    assert np.abs(host(rb.forward_kinematics(x0))[:, :3] - z["fk"][:, :3]).max() < 2e-6
    e, _ = rb.pose_errors(x0, target)
    assert np.abs(host(e)[:, :, 0] - z["e"]).max() < 1e-5
    r1 = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=1, clamp=False, return_residual=True)
    assert np.abs(host(r1["J"]) - z["J_scaled"]).max() < 1e-5
    assert np.abs(host(r1["e"])[:, :, 0] - z["e_scaled"]).max() < 1e-5
    ok = np.linalg.svd(z["J_scaled"], compute_uv=False)[:, -1] >= 2e-2
    assert np.abs(host(r1["x"]) - z["x_step1"])[ok].max() < 5e-3
    rK = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=K, want_errors=True)
    conv = (z["pos_err_K"] < 1e-4) & (z["rot_err_K"] < 1.2e-3)
    assert conv.mean() > 0.85
    assert np.abs(host(rK["pos_err_m"]) - z["pos_err_K"])[conv].max() < 1e-5
    assert np.abs(host(rK["rot_err_rad"]) - z["rot_err_K"])[conv].max() < 1e-5
    # collision: distances within fp32 rounding of the fp64 golden values, masks bit-exact with the fp32 golden masks
    q = dev(z["q_coll"])
    assert np.abs(host(rb.self_collision_distances(q)) - z["self_dists"]).max() < 5e-6
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    ch = H.chain(name)
    rb.set_padded_joint_limits((ch.lo, ch.hi))
    m = rb.collision_masks(q.reshape(1, -1, rb.ndof), want_min_dists=True)
    assert np.array_equal(m["self_mask"].cpu().numpy().reshape(-1).astype(np.uint8), z["self_mask"])
    assert np.array_equal(m["env_mask"].cpu().numpy().reshape(-1).astype(np.uint8), z["env_mask"])
    assert np.array_equal(host(m["min_self"]).reshape(-1), z["min_self_f32"])
    assert np.array_equal(host(m["min_env"]).reshape(-1), z["min_env_f32"])
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


def test_joint_limit_margin_mask_kat_on_gpu(robots):
    """tests/search_test.py:22-57 of the reference, through joint_limit_almost_violations_3d."""
    from cppflow_amd.search import joint_limit_almost_violations_3d

    pi = np.pi
    qs = torch.zeros((2, 3, 8), device=DEV)
    qs[0, 0] = torch.tensor([0.051, 0, 0, 0, 0, 0, 0, 0])
    qs[0, 1] = torch.tensor([0.38615 - 0.001, 0, 0, 0, 0, 0, 0, 0])
    qs[0, 2] = torch.tensor([0.38615 - 0.051, 0, 0, 0, 0, 0, 0, 0])
    qs[1, 0] = torch.tensor([0.38615 - 0.051, 0, 0, -pi, 0, 0, 0, 0])
    qs[1, 1] = torch.tensor([0.38615 - 0.051, 0, 0, -pi + 0.11, 0, 0, 0, 0])
    qs[1, 2] = torch.tensor([0.38615 - 0.051, 0, 0, -pi + 0.11, 0, 0, 0, pi - 0.25])
    got = joint_limit_almost_violations_3d(robots["fetch"], qs, eps_revolute=0.1, eps_prismatic=0.05)
    assert got.dtype == torch.float32
    torch.testing.assert_close(got.cpu(), torch.tensor([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0]]))


def test_pose_residual_kat_on_gpu(robots):
    """tests/optimization_utils_test.py:344-402 of the reference through levenberg_marquardt_only_pose."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_POSE, OptimizationParameters
    from cppflow_amd.optimization import OptimizationProblem, OptimizationState, levenberg_marquardt_only_pose
    from cppflow_amd.optimization_utils import get_6d_pose_errors

    fetch = robots["fetch"]
    qs = dev([[-0.05, 0, 0, 0, 0, 0, 0, 0], [0.25, 0, 0, 0, 0, 0, 0, 0], [0.1, 0, 0, 0, 0, 0, 0, 0]])
    target = fetch.forward_kinematics(dev([[0.05, 0, 0, 0, 0, 0, 0, 0], [0.2, 0, 0, 0, 0, 0, 0, 0], [0.1, 0, 0, 0, 0, 0, 0, 0]]))
    params = OptimizationParameters(**{**ALT_LOSS_V2_1_POSE.__dict__, "alpha_position": 0.25, "alpha_rotation": 1.5})
    problem = problem_from_arrays(fetch, target.cpu().numpy(), device=DEV)
    op = OptimizationProblem(problem, problem.constraints, qs, problem.target_path, 0, 1, None)
    x_new, J, r = levenberg_marquardt_only_pose(op, OptimizationState(qs.clone(), 0, 0.0), params, return_residual=True)
    expected = torch.tensor([[0, 0, 0, 0, 0, 0.1 * 0.25], [0, 0, 0, 0, 0, -0.05 * 0.25], [0, 0, 0, 0, 0, 0.0]])
    torch.testing.assert_close(r[:, :, 0].cpu(), expected, atol=1e-6, rtol=0)
    assert J.shape == (3, 6, 8) and x_new.shape == (3, 8)
    torch.testing.assert_close(J[:, :, 0].cpu(), torch.tensor([0, 0, 0, 0, 0, 0.25]).repeat(3, 1), atol=1e-6, rtol=0)
    # the step reduces the height error (the stretched-out zero pose is singular, so only the residual is pinned)
    e_after, _ = get_6d_pose_errors(fetch, x_new, target)
    assert float(e_after[0, 5, 0].abs()) < 0.1 and float(e_after[1, 5, 0].abs()) < 0.05
    e, cur = get_6d_pose_errors(fetch, qs, target)
    assert e.shape == (3, 6, 1) and cur.shape == (3, 7)
    torch.testing.assert_close(e[:, 5, 0].cpu(), torch.tensor([0.1, -0.05, 0.0]), atol=1e-6, rtol=0)


def test_batched_masks_equal_per_path_masks(robots):
    """Property P3: tests/collision_checking_test.py:27-56 of the reference (Panda, 1 cube, 50 paths x 5 waypoints)."""
    from cppflow_amd.collision_detection import (env_colliding_configs_capsule, get_only_non_colliding_qpaths,
                                                 qpaths_batched_collisions, qpaths_batched_env_collisions,
                                                 qpaths_batched_self_collisions, self_colliding_configs_capsule)  # fmt: skip
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.problems_synthetic import PANDA_1CUBE_OBSTACLES

    rb = robots["panda"]
    n, k = 5, 50
    target = H.oracle64("panda").fk(H.random_configs("panda", n, seed=1))
    problem = problem_from_arrays(rb, target, PANDA_1CUBE_OBSTACLES, device=DEV)
    np.random.seed(0)
    qpaths = [dev(rb.sample_joint_angles(n)) for _ in range(k)]
    safe_gt = [qp for qp in qpaths if not self_colliding_configs_capsule(problem, qp).any()]
    safe_gt = [qp for qp in safe_gt if not env_colliding_configs_capsule(problem, qp).any()]
    assert 0 < len(safe_gt) < k
    q = torch.stack(qpaths)
    sv, ev = qpaths_batched_self_collisions(problem, q), qpaths_batched_env_collisions(problem, q)
    assert sv.dtype == torch.bool and sv.shape == (k, n) and ev.shape == (k, n)
    returned = get_only_non_colliding_qpaths(qpaths, sv, ev)
    assert len(returned) == len(safe_gt) and all(torch.equal(a, b) for a, b in zip(safe_gt, returned))
    s2, e2 = qpaths_batched_collisions(problem, q)
    assert torch.equal(s2, sv) and torch.equal(e2, ev)
    rb.set_obstacles([], [])


def test_x_is_valid_and_run_lm_optimization(robots):
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.optimization import run_lm_optimization, run_lm_pose_refinement
    from cppflow_amd.optimization_utils import x_is_valid

    rb = robots["fetch_arm"]
    z = np.load(os.path.join(GOLDEN, "reference_paths.npz"))
    target = z["fetch_arm__s__truncated"]  # BASELINE config C1: 59 waypoints
    W = target.shape[0]
    problem = problem_from_arrays(rb, target, device=DEV)
    # a smooth joint-space seed: solve the first waypoint from several starts, then track the path warm-started
    o = H.oracle64("fetch_arm")
    ch = H.chain("fetch_arm")
    rng = np.random.RandomState(0)
    starts = rng.uniform(ch.lo * 0.5, ch.hi * 0.5, size=(64, 7))
    sol = o.lm_steps(H.f32(starts), np.tile(target[:1].astype(np.float64), (64, 1)), 40, lm_lambda=1e-4)
    pe, _ = o.pose_metrics_exact(sol, np.tile(target[:1].astype(np.float64), (64, 1)))
    q = sol[int(np.argmin(pe))][None]
    assert pe.min() < 1e-5
    path = []
    for w in range(W):
        q = o.lm_steps(H.f32(q), target[w : w + 1].astype(np.float64), 10, lm_lambda=1e-4)
        path.append(q[0])
    q_star = np.array(path)
    x_seed = dev(np.clip(q_star + 0.002 * rng.randn(W, 7), ch.lo, ch.hi))
    # the noisy seed is not valid (pose error), the refined one is
    x_sol, idx, flags = x_is_valid(problem, problem.constraints, problem.target_path, x_seed, parallel_count=1)
    assert x_sol is None and idx is None and flags[0] is False
    res = run_lm_optimization(problem, x_seed, tmax_sec=30.0, max_n_steps=20, return_if_valid_after_n_steps=15,
                              convergence_threshold=0.3, parallel_count=1, verbosity=0)  # fmt: skip
    pos_cm, rot_deg = (100 * t for t in rb.pose_error_metrics(res.x_opt, problem.target_path)[:1]), None
    assert res.x_opt.shape == (W, 7)
    pe_m, re_rad = rb.pose_error_metrics(res.x_opt, problem.target_path)
    assert float(pe_m.max()) * 100 < problem.constraints.max_allowed_position_error_cm
    assert float(torch.rad2deg(re_rad).max()) < problem.constraints.max_allowed_rotation_error_deg
    del pos_cm, rot_deg
    # batched form: 3 seeds at once, result identical to running each seed alone (rows are independent)
    seeds = torch.cat([x_seed, dev(np.clip(q_star + 0.004 * rng.randn(W, 7), ch.lo, ch.hi)), x_seed.clone()])
    batched = run_lm_pose_refinement(problem, seeds, n_steps=6)
    alone = run_lm_pose_refinement(problem, seeds[W : 2 * W].contiguous(), n_steps=6)
    assert torch.equal(batched.x[W : 2 * W], alone.x) and torch.equal(batched.ext_cost[1], alone.ext_cost[0])
    assert torch.equal(batched.x[:W], batched.x[2 * W :])
    assert batched.pos_err_m.shape == (3, W) and batched.self_mask.dtype == torch.bool
    x_sol, idx, flags = x_is_valid(problem, problem.constraints, problem.target_path, batched.x, parallel_count=3)
    assert flags[0] and flags[1]


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_baseline_configs_on_reference_paths(robots, cfg):
    """BASELINE.json configs 2-4 at their full sizes on the reference's own target paths (tests/golden/reference_paths.npz),
    checked through size-independent properties: joint limits hold, converged rows reproduce the target pose, the fused
    masks equal the standalone masks, and seed-sharding the batch changes nothing (the multi-GPU partition)."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.optimization import run_lm_pose_refinement
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES

    z = np.load(os.path.join(GOLDEN, "reference_paths.npz"))
    name, key, S, obs = {
        "C2": ("panda", "panda__1cube_first64", 128, []),
        "C3": ("fetch", "fetch__hello_first256", 512, []),
        "C4": ("panda", "panda__2cubes_resampled256", 1024, PANDA_2CUBES_OBSTACLES),
    }[cfg]
    rb = robots[name]
    target = z[key]
    W = target.shape[0]
    problem = problem_from_arrays(rb, target, obs, device=DEV)
    ch = H.chain(name)
    g = torch.Generator().manual_seed(0)
    lo, hi = torch.tensor(ch.lo, dtype=torch.float32), torch.tensor(ch.hi, dtype=torch.float32)
    # seeds as SURVEY.md 8(d) builds them: per (seed, waypoint) an IK branch q* (heavily damped LM from a random start),
    # then x0 = clamp(q* + 0.1 randn) -- the construction of the reference's tests/optimization_test.py:82
    x_rand = (lo + (hi - lo) * (0.15 + 0.7 * torch.rand((S * W, rb.ndof), generator=g))).to(DEV)
    ik = rb.lm_pose_steps(x_rand, problem.target_path, 1e-2, 3.5, 0.35, n_steps=60, want_errors=True)
    solved = ((ik["pos_err_m"] < 1e-4) & (ik["rot_err_rad"] < 1.75e-3)).view(S, W)
    assert float(solved.float().mean()) > 0.3
    noise = 0.1 * torch.randn((S * W, rb.ndof), generator=g).to(DEV)
    x0 = torch.minimum(torch.maximum(ik["x"] + noise, lo.to(DEV)), hi.to(DEV)).contiguous()
    r = run_lm_pose_refinement(problem, x0, n_steps=20)
    x = r.x
    assert bool(((x >= lo.to(DEV)) & (x <= hi.to(DEV))).all())
    conv = (r.pos_err_m < 1e-4) & (r.rot_err_rad < 1.75e-3)
    assert float(conv[solved].float().mean()) > 0.85, float(conv[solved].float().mean())
    pe, re = rb.pose_error_metrics(x, problem.target_path)
    assert torch.equal(pe.view(S, W), r.pos_err_m) and torch.equal(re.view(S, W), r.rot_err_rad)
    cur = rb.forward_kinematics(x)
    tgt = problem.target_path.repeat(S, 1)
    assert float((cur[:, :3] - tgt[:, :3]).norm(dim=1)[conv.view(-1)].max()) < 1e-4
    alone = rb.collision_masks(x.view(S, W, -1))
    assert torch.equal(alone["self_mask"], r.self_mask) and torch.equal(alone["env_mask"], r.env_mask)
    assert torch.equal(alone["jlim_mask"], r.jlim_mask) and torch.equal(alone["ext_cost"], r.ext_cost)
    expect_cost = 100.0 * r.jlim_mask.float() + 1000.0 * r.env_mask.float() + 1000.0 * r.self_mask.float()
    assert torch.equal(expect_cost, r.ext_cost)  # cppflow/search.py:146-150
    half = S // 2
    a = run_lm_pose_refinement(problem, x0[: half * W].contiguous(), n_steps=20)
    b = run_lm_pose_refinement(problem, x0[half * W :].contiguous(), n_steps=20)
    assert torch.equal(torch.cat([a.x, b.x]), x) and torch.equal(torch.cat([a.packed[: 4 * half * W], b.packed[: 4 * half * W]]).view(torch.float32), r.ext_cost.view(-1))
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


def test_c5_shape_runs(robots):
    """BASELINE config 5 geometry (12-DoF chain, 512 waypoints) at a reduced seed count: every row finite and clamped."""
    rb = robots["chain12"]
    S, W = 64, 512
    x0, target = H.lm_problem("chain12", S, W, seed=5)
    r = rb.lm_pose_steps(dev(x0), dev(target), 1e-6, 3.5, 0.35, n_steps=10, want_errors=True)
    assert bool(torch.isfinite(r["x"]).all()) and float((r["pos_err_m"] < 1e-4).float().mean()) > 0.9


@pytest.mark.parametrize("name,k,T", [("panda", 175, 59), ("fetch", 64, 40), ("panda", 1024, 64), ("chain12", 7, 5)])
def test_dp_search_on_device_matches_oracle(robots, name, k, T):
    """cppf_dp_search vs the fp32 oracle restatement of cppflow/search.py:128-191: cost table and chosen candidates
    bit-exact (same operation order, first minimal index), best path identical."""
    from cppflow_amd.search import dp_search

    rb = robots[name]
    rng = np.random.RandomState(k + T)
    ch = H.chain(name)
    # k candidate paths that are noisy copies of a few smooth branches, so the optimum actually switches candidates
    base = rng.uniform(ch.lo, ch.hi, size=(4, 1, rb.ndof)) + 0.3 * np.cumsum(rng.randn(4, T, rb.ndof) * 0.1, axis=1)
    q = H.f32(np.clip(base[rng.randint(0, 4, size=k)] + 0.02 * rng.randn(k, T, rb.ndof), ch.lo, ch.hi))
    ext = ((rng.rand(k, T) < 0.15) * 1000.0 + (rng.rand(k, T) < 0.1) * 100.0).astype(np.float32)
    path, idx, costsT = rb.dp_search(dev(q), dev(ext))
    want_idx, want_costs = H.oracle32(name).dp_search(q, ext)
    assert np.array_equal(host(costsT).T, want_costs)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(host(path), q[want_idx, np.arange(T)])
    assert len(set(want_idx.tolist())) > 1 or k < 8
    # the reference-named entry point with the masks instead of the cost matrix
    z = torch.zeros((k, T), dtype=torch.bool, device=DEV)
    p2 = dp_search(rb, dev(q), z, z)
    jl = rb.collision_masks(dev(q), only=("jlim",))  # padding unset -> all zero
    assert p2.shape == (T, rb.ndof) and not bool(jl["jlim_mask"].any())


def test_edge_sizes(robots):
    """Empty, single-row, single-waypoint and ragged inputs (the reference's functions accept any k, T >= 1)."""
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays

    rb = robots["panda"]
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    empty = torch.empty((0, 7), device=DEV)
    tgt1 = rb.forward_kinematics(dev(H.random_configs("panda", 1, seed=0)))
    assert rb.jacobian(empty).shape == (0, 6, 7) and rb.self_collision_distances(empty).shape == (0, rb.n_collision_pairs)
    r = rb.lm_pose_steps(empty, tgt1, 1e-6, 3.5, 0.35, n_steps=3, want_errors=True, want_collisions=True)
    assert r["x"].shape == (0, 7) and r["ext_cost"].shape == (0,)
    assert rb.collision_masks(torch.empty((0, 5, 7), device=DEV))["self_mask"].shape == (0, 5)
    # W = 1 (a single target for every row), n = 1, and n = 257 (one lane into a second workgroup)
    for n in (1, 257):
        x = dev(H.random_configs("panda", n, seed=n))
        r = rb.lm_pose_steps(x, tgt1, 1e-6, 3.5, 0.35, n_steps=2, want_errors=True, want_collisions=True)
        want = H.oracle64("panda").lm_steps(host(x), np.tile(host(tgt1), (n, 1)), 2)
        ok = np.abs(want - host(x)).max(axis=1) < 0.5
        if ok.any():
            assert np.abs(host(r["x"]) - want)[ok].max() < 5e-3
        assert r["self_mask"].shape == (n,) and bool(torch.isfinite(r["x"]).all())
        m = H.oracle32("panda").masks(host(r["x"]), *H.box_corners([c for c, _ in obs], [T for _, T in obs]), None, None)
        assert np.array_equal(r["self_mask"].cpu().numpy(), m["self_mask"]) and np.array_equal(r["env_mask"].cpu().numpy(), m["env_mask"])
    # dp_search with a single candidate / a single timestep
    q = dev(H.random_configs("panda", 6, seed=3)).reshape(1, 6, 7)
    path, idx, _ = rb.dp_search(q, torch.zeros((1, 6), device=DEV))
    assert torch.equal(path, q[0]) and idx.tolist() == [0] * 6
    q = dev(H.random_configs("panda", 5, seed=4)).reshape(5, 1, 7)
    ext = dev([[3.0], [1.0], [2.0], [1.0], [5.0]])
    path, idx, _ = rb.dp_search(q, ext)
    assert idx.tolist() == [1] and torch.equal(path[0], q[1, 0])  # first minimal index
    # seed_validity with W = 1: no joint deltas
    sv = rb.seed_validity(dev(H.random_configs("panda", 3, seed=5)), tgt1)
    assert sv.shape == (3, 4) and float(sv[:, 2:].abs().max()) == 0.0
    rb.set_obstacles([], [])


def test_c5_full_size(robots):
    """BASELINE config 5 at full size on one GPU: 12-DoF chain, 4096 seeds x 512 waypoints = 2 097 152 rows, 2 cuboids.
    Checked by sampling: 4096 random rows of the result against the oracle evaluated at the kernel's own x."""
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays

    rb = robots["chain12"]
    S, W = 4096, 512
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    ch = H.chain("chain12")
    g = torch.Generator().manual_seed(0)
    lo, hi = torch.tensor(ch.lo, dtype=torch.float32), torch.tensor(ch.hi, dtype=torch.float32)
    q_star = lo + (hi - lo) * torch.rand((W, 12), generator=g)
    target = rb.forward_kinematics(q_star.to(DEV))
    x0 = torch.minimum(torch.maximum(q_star[None] + 0.1 * torch.randn((S, W, 12), generator=g), lo), hi).reshape(S * W, 12).to(DEV)
    r = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=10, want_errors=True, want_collisions=True, want_min_dists=True)
    assert float((r["pos_err_m"] < 1e-4).float().mean()) > 0.9
    rows = torch.randint(0, S * W, (4096,), generator=g)
    x = host(r["x"][rows.to(DEV)])
    tgt = host(target)[(rows % W).numpy()]
    pe, re = H.oracle64("chain12").pose_metrics_exact(x, tgt)
    assert np.abs(host(r["pos_err_m"][rows.to(DEV)]) - pe).max() < 1e-5
    assert np.abs(host(r["rot_err_rad"][rows.to(DEV)]) - re).max() < 1e-5
    lo_b, hi_b = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    m = H.oracle32("chain12").masks(x, lo_b, hi_b, None, None)
    assert np.array_equal(r["self_mask"][rows.to(DEV)].cpu().numpy(), m["self_mask"])
    assert np.array_equal(r["env_mask"][rows.to(DEV)].cpu().numpy(), m["env_mask"])
    assert np.array_equal(host(r["min_self"][rows.to(DEV)]), m["min_self"])
    # the K-step LM result itself on the sampled rows against the fp64 reference-order oracle
    o64 = H.oracle64("chain12")
    x0_s = host(x0[rows.to(DEV)])
    x_o = o64.lm_steps(x0_s, tgt, 10, 1e-6, 3.5, 0.35, solver=0)
    pe_o, re_o = o64.pose_metrics_exact(x_o, tgt)
    pe_g, re_g = host(r["pos_err_m"][rows.to(DEV)]), host(r["rot_err_rad"][rows.to(DEV)])
    conv, conv_g = (pe_o < 1e-4) & (re_o < 1.2e-3), (pe_g < 1e-4) & (re_g < 1.2e-3)
    assert conv.mean() > 0.9 and abs(conv_g.mean() - conv.mean()) < 0.01
    settled = conv & conv_g & (pe_o < 5e-6)
    assert settled.sum() > 0.8 * conv.sum()
    assert np.abs(pe_g - pe_o)[settled].max() < 1e-5 and np.abs(re_g - re_o)[settled].max() < 1e-5
    rb.set_obstacles([], [])


def _full_params(**kw):
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters

    d = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    d.update(kw)
    return OptimizationParameters(**d)


@pytest.mark.parametrize("name,variant", [("panda", "diff_preset"), ("fetch", "diff_preset"), ("panda", "pose_only"),
                                           ("panda", "everything"), ("chain12", "diff_preset")])
def test_coupled_lm_step_matches_dense_reference_order_oracle(robots, name, variant):
    """cppf_lm_full_step (block-tridiagonal elimination per trajectory) vs the oracle's dense restatement of
    levenberg_marquardt_full + LmResidualFns.get_r_and_J (cppflow/optimization.py:95-144, optimization_utils.py:486-731)."""
    rb = robots[name]
    S, T = 3, 24
    rng = np.random.RandomState(7)
    ch = H.chain(name)
    obs = H.PANDA_2CUBES if name in ("panda", "chain12") else [H.cuboid_obstacle(0.7, 0.1, 0.8, 0.3, 0.3, 0.3)]
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    # smooth trajectories; the first one is anchored at a configuration that collides (with itself or an obstacle), so
    # that the stacked residual of trajectory 0 contains active collision rows
    cand = H.random_configs(name, 4000, seed=11)
    m = H.oracle64(name).masks(cand, lo, hi, None, None)
    hit = cand[np.flatnonzero((m["self_mask"] | m["env_mask"]) > 0)[0]]
    base = np.clip(hit[None, :] + np.cumsum(0.02 * rng.randn(T, rb.ndof), axis=0), ch.lo, ch.hi)
    # the S seeds are small perturbations of one trajectory: they share the target path, as seeds of one problem do
    x = H.f32(np.clip(base[None] + 0.003 * rng.randn(S, T, rb.ndof), ch.lo, ch.hi).reshape(S * T, rb.ndof))
    target = H.f32(H.oracle64(name).fk(H.f32(base)) + np.concatenate([0.002 * rng.randn(T, 3), np.zeros((T, 4))], axis=1))
    if variant == "diff_preset":
        pm = _full_params()
    elif variant == "pose_only":
        pm = _full_params(use_pose=True, alpha_position=3.5, alpha_rotation=0.35, use_differencing=False,
                          use_virtual_configs=False, use_self_collisions=False, use_env_collisions=False)
    else:
        pm = _full_params(use_pose=True, alpha_position=1.1, alpha_rotation=1.0, alpha_self_collision=0.05,
                          alpha_env_collision=0.03, alpha_differencing=0.01, alpha_differencing_prismatic_scaling=2.0)
    xv = H.f32(x + 0.01 * rng.randn(*x.shape)) if variant == "everything" else None
    pm.virtual_configs = dev(xv) if xv is not None else torch.tensor([])
    got = host(rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=pm.virtual_configs))
    # the other elimination order (waypoint after waypoint instead of parallel cyclic reduction over the waypoints) must
    # land on the same step
    rb.debug_set("pcr_max_rows", 0)
    try:
        sequential = host(rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=pm.virtual_configs))
        # ... and so must the one-wavefront-per-trajectory kernel the row-per-lane one replaced (d <= 8 only)
        rb.debug_set("full_rows", 0)
        per_wave = host(rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=pm.virtual_configs))
    finally:
        rb.debug_set("pcr_max_rows", -1)
        rb.debug_set("full_rows", 1)
    want, r = H.oracle64(name).lm_full_step(x, target, pm, S, T, virtual_configs=xv, boxes_lo=lo, boxes_hi=hi, return_residual=True)
    n_fixed = (6 * T if pm.use_pose else 0) + ((T - 1) * rb.ndof if pm.use_differencing else 0) + (8 * rb.ndof if pm.use_virtual_configs else 0)
    if variant != "pose_only":
        assert r.shape[0] > n_fixed, "the case must contain active collision rows"
    step = np.abs(want - x).max()
    assert step > 1e-4
    if pm.use_pose:
        # With the pose block the d x d blocks are J^T J + small diagonal: rank 6 of 7, cond ~1e6-1e7 -- in fp32 (the
        # reference's dtype too) the step carries null-space noise, so parity is stated in task space
        # and, in joint space, at the reference's own inter-formulation scale on well-conditioned rows.
        Js = H.oracle64(name).lm_step(x, H.stacked(target, S), lm_lambda=pm.lm_lambda, alpha_position=pm.alpha_position,
                                      alpha_rotation=pm.alpha_rotation)[1]
        ok = np.linalg.svd(Js, compute_uv=False)[:, -1] >= 2e-2
        assert ok.mean() > 0.5
        assert np.abs(np.einsum("nij,nj->ni", Js, got - want))[ok].max() < 2e-3
        assert np.abs(got - want)[ok].max() < 3e-2
        if variant == "pose_only":
            # property P1 (tests/optimization_test.py:74-100): coupled step with only the pose block == batched step
            batched = host(rb.lm_pose_steps(dev(x), dev(target), 1e-6, 3.5, 0.35, n_steps=1, clamp=False)["x"])
            assert np.abs(np.einsum("nij,nj->ni", Js, got - batched))[ok].max() < 2e-3
            assert np.abs(want - batched)[ok].max() < 5e-3  # dual-form kernel vs the dense fp64 formulation
    else:
        assert np.abs(got - want).max() < 2e-4 + 2e-3 * step, (np.abs(got - want).max(), step)
        assert np.abs(sequential - want).max() < 2e-4 + 2e-3 * step, (np.abs(sequential - want).max(), step)
        assert np.abs(per_wave - want).max() < 2e-4 + 2e-3 * step, (np.abs(per_wave - want).max(), step)
    rb.set_obstacles([], [])


@pytest.mark.parametrize("name,T", [("panda", 256), ("fetch", 300), ("panda", 1), ("panda", 2), ("fetch_arm", 59)])
def test_coupled_step_parallel_in_time_equals_sequential_elimination(robots, name, T):
    """cppf_lm_full_step at the reference's cadence (one trajectory, optimization.py:128) runs parallel cyclic reduction over
    the waypoints; at full path lengths (beyond what the dense oracle handles in seconds) it must agree with the
    waypoint-after-waypoint elimination of the same system, and with the oracle on the short path."""
    rb, ch = robots[name], H.chain(name)
    rng = np.random.RandomState(T)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    pm = _full_params()
    pm.virtual_configs = torch.tensor([])
    S = 2
    base = np.clip(rng.uniform(ch.lo, ch.hi)[None, :] * 0.5 + np.cumsum(0.03 * rng.randn(T, rb.ndof), axis=0), ch.lo, ch.hi)
    x = H.f32(np.clip(base[None] + 0.01 * rng.randn(S, T, rb.ndof), ch.lo, ch.hi).reshape(S * T, rb.ndof))
    target = H.f32(H.oracle64(name).fk(H.f32(base)))
    if 2 * pm.n_virtual_configs >= T:
        pm = _full_params(use_virtual_configs=False)
        pm.virtual_configs = torch.tensor([])
    pcr = host(rb.lm_full_step(dev(x), dev(target), pm))
    # the same reduction with its state in the caller's workspace instead of LDS (what W > 256 runs) and in LDS with one lane
    # per waypoint: the same arithmetic, bit for bit; the default (two half-workgroups per waypoint, the t + s side accumulated
    # separately) differs from them by rounding only
    try:
        rb.debug_set("pcr_lds", 0)
        pcr_ws = host(rb.lm_full_step(dev(x), dev(target), pm))
        rb.debug_set("pcr_lds", 1)
        pcr_one = host(rb.lm_full_step(dev(x), dev(target), pm))
    finally:
        rb.debug_set("pcr_lds", 2)
    assert np.array_equal(pcr_one, pcr_ws)
    assert np.abs(pcr - pcr_ws).max() < 1e-6 + 1e-4 * np.abs(pcr_ws - x).max()
    rb.debug_set("pcr_max_rows", 0)
    try:
        seq = host(rb.lm_full_step(dev(x), dev(target), pm))
    finally:
        rb.debug_set("pcr_max_rows", -1)
    step = np.abs(seq - x).max()
    assert np.isfinite(pcr).all() and np.abs(pcr - seq).max() < 1e-5 + 1e-3 * step, (np.abs(pcr - seq).max(), step)
    # eight trajectories per wavefront (the default beyond the parallel-in-time range) vs one wavefront per trajectory, on a
    # ragged count of trajectories: 11 = one full wavefront of eight + three groups of a second one
    S11 = 11
    x11 = H.f32(np.clip(base[None] + 0.01 * rng.randn(S11, T, rb.ndof), ch.lo, ch.hi).reshape(S11 * T, rb.ndof))
    rb.debug_set("pcr_max_rows", 0)
    try:
        rows = host(rb.lm_full_step(dev(x11), dev(target), pm))
        rb.debug_set("full_rows", 0)
        wave = host(rb.lm_full_step(dev(x11), dev(target), pm))
    finally:
        rb.debug_set("pcr_max_rows", -1)
        rb.debug_set("full_rows", 1)
    step11 = np.abs(wave - x11).max()
    assert np.isfinite(rows).all() and np.abs(rows - wave).max() < 1e-5 + 1e-3 * step11, (np.abs(rows - wave).max(), step11)
    # the oracle at EVERY path length: the same residual rows in band storage, banded Cholesky (oracle/lmik_oracle.c:
    # orc_lm_full_step_banded; held equal to the reference's dense formulation at T <= 64 by tests/test_oracle_kats.py)
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    want = H.oracle64(name).lm_full_step(x, target, pm, S, T, boxes_lo=lo, boxes_hi=hi, banded=True)
    assert np.abs(pcr - want).max() < 2e-4 + 2e-3 * step
    assert np.abs(seq - want).max() < 2e-4 + 2e-3 * step
    rb.set_obstacles([], [])


def test_alternating_loop_with_differencing(robots):
    """run_lm_optimization end to end on BASELINE config C1's path: pose steps until the pose is valid, then coupled
    differencing steps; the result is valid and no rougher than what the pose-only loop returns."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.evaluation_utils import angular_changes
    from cppflow_amd.optimization import run_lm_optimization

    rb = robots["fetch_arm"]
    z = np.load(os.path.join(GOLDEN, "reference_paths.npz"))
    target = z["fetch_arm__s__truncated"]
    W = target.shape[0]
    problem = problem_from_arrays(rb, target, device=DEV)
    o, ch, rng = H.oracle64("fetch_arm"), H.chain("fetch_arm"), np.random.RandomState(0)
    starts = rng.uniform(ch.lo * 0.5, ch.hi * 0.5, size=(64, 7))
    t0 = np.tile(target[:1].astype(np.float64), (64, 1))
    sol = o.lm_steps(H.f32(starts), t0, 40, lm_lambda=1e-4)
    q = sol[int(np.argmin(o.pose_metrics_exact(sol, t0)[0]))][None]
    path = []
    for w in range(W):
        q = o.lm_steps(H.f32(q), target[w : w + 1].astype(np.float64), 10, lm_lambda=1e-4)
        path.append(q[0])
    x_seed = dev(np.clip(np.array(path) + 0.01 * rng.randn(W, 7), ch.lo, ch.hi))
    kw = dict(tmax_sec=60.0, max_n_steps=20, return_if_valid_after_n_steps=15, convergence_threshold=0.3, verbosity=0)
    a = run_lm_optimization(problem, x_seed, on_pose_valid="stop", **kw)
    b = run_lm_optimization(problem, x_seed, on_pose_valid="differencing", **kw)
    assert b.is_valid and b.n_steps_taken >= a.n_steps_taken
    tl = lambda x: float(angular_changes(x).abs().sum())  # noqa: E731
    assert tl(b.x_opt) <= tl(a.x_opt) + 1e-3
    pe, re = rb.pose_error_metrics(b.x_opt, problem.target_path)
    assert float(pe.max()) * 100 < 0.01 and float(torch.rad2deg(re).max()) < 0.1


def test_planner_end_to_end(robots):
    """CppFlowPlanner.generate_plan on a problem file in the reference's format (panda__line: a straight 20 cm line next
    to a cuboid): candidates -> collision masks -> dp_search -> LM optimisation, all on the device; the plan is valid."""
    from cppflow_amd.data_type_utils import problem_from_filename
    from cppflow_amd.data_types import PlannerSettings
    from cppflow_amd.planners import CppFlowPlanner, LmIkSeedProvider, PlannerSearcher

    problem = problem_from_filename(None, "panda__line", robot=None, device=DEV)
    settings = PlannerSettings(k=64, tmax_sec=30.0, anytime_mode_enabled=False, verbosity=0)
    searcher = PlannerSearcher(settings, problem.robot, LmIkSeedProvider(seed=1))
    s = searcher.generate_plan(problem)
    assert s.plan.q_path.shape == (problem.n_timesteps, 7) and s.timing.dp_search > 0
    planner = CppFlowPlanner(settings, problem.robot, LmIkSeedProvider(seed=1))
    r = planner.generate_plan(problem)
    plan = r.plan
    assert plan.is_valid, str(plan)
    assert plan.max_positional_error_cm < 0.01 and plan.max_rotational_error_deg < 0.1 and plan.mjac_deg < 7.0
    assert not bool(plan.self_colliding_per_ts.any()) and not bool(plan.env_colliding_per_ts.any())
    # the optimised plan is at least as accurate as the raw search path
    assert plan.max_positional_error_cm <= s.plan.max_positional_error_cm + 1e-4  # both sit at the fp32 floor (~2e-5 cm)
    problem.robot.set_obstacles([], [])
    problem.robot.set_joint_limit_padding(None, None)


def test_seed_summary_matches_oracle_pieces(robots):
    """cppf_seed_summary reduces the fused launch's own per-row outputs: compare with the same reductions done in numpy on
    those outputs, and its four validity maxima with cppf_seed_validity (which recomputes FK)."""
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays

    rb = robots["fetch"]
    obs = obstacle_arrays([(0.7, 0.1, 0.8, 0.3, 0.3, 0.3)])
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    S, W = 37, 59
    x0, target = H.lm_problem("fetch", S, W, seed=31)
    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=DEV)
    r = rb.lm_pose_steps(dev(x0), dev(target), 1e-6, 3.5, 0.35, n_steps=3, packed_out=packed)
    got = host(rb.seed_summary(r["x"], packed, S, W))
    pe = host(r["pos_err_m"]).reshape(S, W)
    re = host(r["rot_err_rad"]).reshape(S, W)
    np.testing.assert_allclose(got[:, 0], 100 * pe.max(1), rtol=1e-6)
    np.testing.assert_allclose(got[:, 1], np.rad2deg(re.max(1)), rtol=1e-6)
    sv = host(rb.seed_validity(r["x"], dev(target)))
    np.testing.assert_allclose(got[:, :4], sv, rtol=1e-5, atol=1e-6)
    for col, key in ((4, "self_mask"), (5, "env_mask"), (6, "jlim_mask")):
        assert np.array_equal(got[:, col], r[key].cpu().numpy().reshape(S, W).sum(1).astype(np.float64))
    np.testing.assert_allclose(got[:, 7], host(r["ext_cost"]).reshape(S, W).sum(1), rtol=1e-6)
    want = H.oracle64("fetch").seed_validity(host(r["x"]), H.stacked(target, S), S, W)
    assert np.abs(got[:, 2:4] - want[:, 2:4]).max() < 1e-3
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["panda", "fetch"])
@pytest.mark.parametrize("generic", [False, True])
def test_in_launch_seed_summary_equals_separate_reduction(robots, name, generic):
    """cppf_lm_outputs.seed_summary: the fused launch's own per-seed epilogue (W = 64, 128, 256: whole seeds per workgroup,
    partially filled last workgroups, seeds spanning 1 / 2 / 4 wavefronts) and the fall-back (any other W)
    both equal cppf_seed_summary over the same launch's per-row outputs bit for bit -- every reduction in it is a max or an
    exact sum.  Also without any per-row output buffer (the summary alone implies the collision stage)."""
    from cppflow_amd.problems_synthetic import obstacle_arrays

    rb = robots[name]
    obs = H.PANDA_2CUBES if name == "panda" else obstacle_arrays([(0.7, 0.1, 0.8, 0.3, 0.3, 0.3)])
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    rb.debug_set("force_generic", int(generic))
    try:
        for S, W in ((5, 1), (9, 2), (7, 32), (5, 64), (3, 128), (3, 256), (4, 100), (2, 300)):
            x0, target = H.lm_problem(name, S, W, seed=100 + W)
            x0[: W // 2] = 0.0  # a stretch of colliding / limit-hugging rows so that the counts are not all zero
            packed = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=DEV)
            fused = torch.full((S, 8), -1.0, dtype=torch.float32, device=DEV)
            r = rb.lm_pose_steps(dev(x0), dev(target), 1e-6, 3.5, 0.35, n_steps=2, packed_out=packed, summary_out=fused)
            want = rb.seed_summary(r["x"], packed, S, W)
            assert torch.equal(fused, want), (S, W, fused, want)
            if W in (64, 128, 256):
                alone = torch.full((S, 8), -1.0, dtype=torch.float32, device=DEV)
                r2 = rb.lm_pose_steps(dev(x0), dev(target), 1e-6, 3.5, 0.35, n_steps=2, summary_out=alone)
                assert torch.equal(alone, want) and torch.equal(r2["x"], r["x"]), (S, W)
            else:
                with pytest.raises(AssertionError, match="seed_summary"):
                    rb.lm_pose_steps(dev(x0), dev(target), 1e-6, 3.5, 0.35, n_steps=2,
                                     summary_out=torch.empty((S, 8), dtype=torch.float32, device=DEV))  # fmt: skip
        assert float(want[:, 4:].sum()) >= 0
    finally:
        rb.debug_set("force_generic", 0)
        rb.set_obstacles([], [])
        rb.set_joint_limit_padding(None, None)


def test_entry_points_are_hip_graph_capturable(robots):
    """No entry point allocates, frees or synchronises, so a sequence of launches can be captured into a HIP graph (here:
    the reference's per-iteration cadence -- five K = 1 { step ; clamp } launches ping-ponging two buffers, then the collision
    masks) and replayed; the replay reproduces the eager result bit for bit."""
    rb = robots["panda"]
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    S, W = 64, 64
    x0, target = H.lm_problem("panda", S, W, seed=41)
    x_in, tgt = dev(x0), dev(target)
    bufs = [torch.empty_like(x_in), torch.empty_like(x_in)]
    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=DEV)
    plans = [rb.lm_launch_plan(src, tgt, 1e-6, 3.5, 0.35, n_steps=1, x_out=dst,
                               packed_out=packed if last else None)
             for src, dst, last in ((x_in, bufs[0], False), (bufs[0], bufs[1], False), (bufs[1], bufs[0], False),
                                    (bufs[0], bufs[1], False), (bufs[1], bufs[0], True))]  # fmt: skip

    def run():
        for p in plans:
            p.launch()

    run()
    torch.cuda.synchronize()
    eager_x, eager_packed = bufs[0].clone(), packed.clone()
    bufs[0].zero_(), bufs[1].zero_(), packed.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    bufs[0].zero_(), bufs[1].zero_(), packed.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(bufs[0], eager_x) and torch.equal(packed, eager_packed)
    want = H.oracle64("panda").lm_steps(x0, H.stacked(target, S), 5)
    ok = np.abs(want - x0).max(axis=1) < 0.5
    # five single steps against the fp64 oracle: identical in the typical row, bounded on the rows whose 7th (null-space)
    # direction is only held by the damping
    d = np.abs(host(eager_x) - want)[ok]
    assert np.median(d) < 1e-5 and d.max() < 5e-2
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["panda", "fetch", "chain12"])
def test_plan_metrics_match_oracle(robots, name):
    """cppf_plan_metrics (one wavefront per path) vs orc_plan_metrics (cppflow/data_types.py:140-264 restated): maxima and
    counts to fp32 rounding of the per-row errors, means and lengths to the rounding of a 64-lane tree sum."""
    rb, ch, orc = robots[name], H.chain(name), H.oracle64(name)
    rng = np.random.RandomState(8)
    for S, W in ((7, 59), (3, 200), (4, 1)):
        base = rng.uniform(ch.lo, ch.hi, size=(S, 1, ch.ndof))
        x = np.clip(base + np.cumsum(0.03 * rng.randn(S, W, ch.ndof), axis=1), ch.lo - 0.01, ch.hi + 0.01)
        x = H.f32(x.reshape(S * W, ch.ndof))
        target = H.f32(orc.fk(H.f32(x[:W] + 0.002)))
        sm, em = rng.rand(S * W) < 0.1, rng.rand(S * W) < 0.3
        q_init = H.f32(x[0] + 0.03)
        want = orc.plan_metrics(x, H.stacked(target, S), S, W, sm, em, q_init)
        got = host(rb.plan_metrics(dev(x), dev(target), torch.tensor(sm, device=DEV), torch.tensor(em, device=DEV),
                                   dev(q_init)))  # fmt: skip
        assert got.shape == (S, 16)
        np.testing.assert_allclose(got[:, [0, 1]], want[:, [0, 1]], rtol=2e-3, atol=2e-4)  # cm; fp32 FK noise ~1e-6 m
        np.testing.assert_allclose(got[:, [2, 3]], want[:, [2, 3]], rtol=2e-3, atol=2.6e-2)  # deg; acos-clamp floor region
        np.testing.assert_allclose(got[:, 4:8], want[:, 4:8], rtol=1e-5, atol=1e-5)
        assert np.array_equal(got[:, 8:11], want[:, 8:11])
        np.testing.assert_allclose(got[:, 11], want[:, 11], rtol=1e-6)
        assert np.all(got[:, 12:] == 0)
        none = host(rb.plan_metrics(dev(x), dev(target)))
        assert np.all(none[:, 9:12] == 0) and np.array_equal(none[:, :9], got[:, :9])
        sv = host(rb.seed_validity(dev(x), dev(target)))
        assert np.array_equal(sv, got[:, [0, 2, 4, 5]])  # the same maxima as cppf_seed_validity, bit for bit


def test_plans_from_qpaths_batch_equals_single(robots):
    """plans_from_qpaths: every candidate path of a problem evaluated in one go; each Plan equals plan_from_qpath of that
    path alone and its device-side scalars equal the torch fall-backs of the Plan container."""
    from cppflow_amd.data_type_utils import plan_from_qpath, plans_from_qpaths, problem_from_arrays
    from cppflow_amd.data_types import Plan

    rb = robots["fetch"]
    S, W = 6, 40
    x0, target = H.lm_problem("fetch", S, W, seed=12)
    problem = problem_from_arrays(rb, H.f32(target), [(0.9, 0.0, 0.6, 0.3, 0.3, 0.3)], device=DEV)
    r = rb.lm_pose_steps(dev(x0), problem.target_path, 1e-6, 3.5, 0.35, n_steps=15)
    q = r["x"].view(S, W, -1)
    plans = plans_from_qpaths(q, problem)
    assert len(plans) == S
    for s in (0, S - 1):
        one = plan_from_qpath(q[s], problem)
        assert torch.equal(one.metrics, plans[s].metrics) and torch.equal(one.pose_path, plans[s].pose_path)
        assert one.is_valid == plans[s].is_valid
        fields = {k: getattr(one, k) for k in ("q_path", "q_path_revolute", "q_path_prismatic", "pose_path", "target_path",
                  "robot_joint_limits", "self_colliding_per_ts", "env_colliding_per_ts", "positional_errors",
                  "rotational_errors", "provided_initial_configuration", "constraints")}  # fmt: skip
        torch_only = Plan(**fields)
        for name in ("max_positional_error_cm", "mean_positional_error_cm", "max_rotational_error_deg", "mean_rotational_error_deg",
                     "mjac_deg", "mjac_cm", "path_length_rad", "path_length_m"):  # fmt: skip
            assert getattr(one, name) == pytest.approx(getattr(torch_only, name), rel=1e-4, abs=1e-6), name
        assert one.joint_limits_violated == torch_only.joint_limits_violated and "Plan {" in str(one)
    rb.set_obstacles([], [])


def test_large_batch_indexing(robots):
    """16.8 M rows (16 384 seeds x 1 024 waypoints, 470 MB of x): every index is past 2^24 somewhere and the J / e outputs are
    past 2^31 BYTES; the first and last rows must equal the same rows run alone, bit for bit (rows are independent), and
    the in-launch per-seed summary must equal the separate reduction."""
    rb = robots["panda"]
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    S, W, d = 16384, 1024, rb.ndof
    g = torch.Generator(device=DEV).manual_seed(5)
    lo = torch.tensor([l for l, _ in rb.actuated_joints_limits], device=DEV)
    hi = torch.tensor([u for _, u in rb.actuated_joints_limits], device=DEV)
    q_star = lo + (hi - lo) * torch.rand((W, d), generator=g, device=DEV)
    target = rb.forward_kinematics(q_star)
    x0 = torch.empty((S, W, d), dtype=torch.float32, device=DEV)
    x0.normal_(0.0, 0.1, generator=g)
    x0 += q_star[None]
    x0 = torch.minimum(torch.maximum(x0, lo), hi).view(S * W, d)
    n = S * W
    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
    summary = torch.empty((S, 8), dtype=torch.float32, device=DEV)
    r = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=3, packed_out=packed, summary_out=summary, return_residual=True)
    assert r["J"].numel() * 4 > 2**31 and bool(torch.isfinite(r["x"]).all())
    for sl in (slice(0, 2 * W), slice(n - 2 * W, n)):
        pk = torch.empty(rb.PACKED_BYTES_PER_ROW * 2 * W, dtype=torch.uint8, device=DEV)
        a = rb.lm_pose_steps(x0[sl].contiguous(), target, 1e-6, 3.5, 0.35, n_steps=3, packed_out=pk, return_residual=True)
        for key in ("x", "J", "e", "ext_cost", "pos_err_m", "rot_err_rad", "self_mask", "env_mask", "jlim_mask"):
            assert torch.equal(a[key], r[key][sl]), key
    assert torch.equal(summary, rb.seed_summary(r["x"], packed, S, W))
    masks = rb.collision_masks(r["x"].view(S, W, d))
    assert torch.equal(masks["ext_cost"].view(-1), r["ext_cost"])
    del r, packed, x0, masks
    torch.cuda.empty_cache()
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_distance_jacobians_match_oracle(robots, name):
    """Robot.self_collision_distances_jacobian / env_collision_distances_jacobian (jrl API, call sites
    cppflow/optimization_utils.py:670, 710) vs the fp64 oracle's analytic gradients (themselves checked against finite
    differences on the CPU); the distances returned alongside equal the standalone distance kernels bit for bit."""
    rb, orc = robots[name], H.oracle64(name)
    x = H.random_configs(name, 300, seed=21)
    J, dist = rb.self_collision_distances_jacobian(dev(x), return_distances=True)
    assert torch.equal(dist, rb.self_collision_distances(dev(x)))
    want_d, want_g = orc.self_dists_grads(x)
    assert J.shape == want_g.shape and np.abs(host(dist) - want_d).max() < 2e-6
    ok = want_d > 1e-4  # touching segments have no direction; near-parallel pairs switch closest points at fp32 noise
    assert np.abs(host(J) - want_g)[ok].max() < 2e-3 and np.median(np.abs(host(J) - want_g)[ok]) < 1e-6
    cub, T = H.PANDA_2CUBES[0]
    lo, hi = H.box_corners([cub], [T])
    Je, de = rb.env_collision_distances_jacobian(dev(x), cub, T, return_distances=True)
    assert torch.equal(de, rb.env_collision_distances(dev(x), cub, T))
    want_d, want_g = orc.env_dists_grads(x, lo[0], hi[0])
    assert Je.shape == want_g.shape and np.abs(host(de) - want_d).max() < 2e-6
    ok = want_d > 1e-4
    assert np.abs(host(Je) - want_g)[ok].max() < 2e-3 and np.median(np.abs(host(Je) - want_g)[ok]) < 1e-6
    assert torch.equal(rb.self_collision_distances_jacobian(dev(x)), J)


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_dense_residual_and_jacobian_reproduce_the_device_step(robots, name):
    """LmResidualFns.get_r_and_J (cppflow/optimization_utils.py:486-731) builds the dense r and J of one trajectory from
    the device's per-row quantities.  (1) r equals the oracle's stacked residual row for row; (2) the reference's own
    dense step  x + solve(J^T J + lambda I, J^T r)  (optimization.py:95-113), evaluated in fp64 on those matrices, lands on
    what cppf_lm_full_step computes without ever forming them; (3) the reference's differencing known answer
    (tests/optimization_utils_test.py:590-637)."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.optimization import OptimizationProblem, OptimizationState, levenberg_marquardt_full
    from cppflow_amd.optimization_utils import LmResidualFns

    rb, ch, orc = robots[name], H.chain(name), H.oracle64(name)
    rng = np.random.RandomState(17)
    T = 20
    obs = H.PANDA_2CUBES
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    cand = H.random_configs(name, 4000, seed=11)
    m = orc.masks(cand, lo, hi, None, None)
    hit = cand[np.flatnonzero((m["self_mask"] | m["env_mask"]) > 0)[0]]
    x = H.f32(np.clip(hit[None, :] + np.cumsum(0.02 * rng.randn(T, rb.ndof), axis=0), ch.lo, ch.hi))
    target = H.f32(orc.fk(x))
    problem = problem_from_arrays(rb, target, [(0.2, 0.3, 0.4, 0.15, 0.15, 0.15), (-0.25, 0.3, 0.75, 0.15, 0.15, 0.15)], device=DEV)
    pm = _full_params()
    pm.virtual_configs = dev(H.f32(x + 0.01 * rng.randn(*x.shape)))
    opt_problem = OptimizationProblem(problem, problem.constraints, dev(x), problem.target_path, 0, 1, None)
    x_new, jac, res = levenberg_marquardt_full(opt_problem, OptimizationState(dev(x), 0, 0.0), pm, return_residual=True)
    J, r = host(jac.get_J()), host(res.get_r())[:, 0]
    assert J.shape == (r.shape[0], T * rb.ndof) and res.self_collisions is not None
    n_fixed = (T - 1) * rb.ndof + 8 * rb.ndof
    assert r.shape[0] > n_fixed, "the case must contain active collision rows"
    _, r_want = orc.lm_full_step(x, target, pm, 1, T, virtual_configs=host(pm.virtual_configs), boxes_lo=lo, boxes_hi=hi,
                                 return_residual=True)  # fmt: skip
    assert r_want.shape == r.shape and np.abs(r - r_want).max() < 2e-7
    A = J.T @ J + pm.lm_lambda * np.eye(T * rb.ndof)
    dense = x + np.linalg.solve(A, J.T @ r).reshape(T, rb.ndof)
    step = np.abs(dense - x).max()
    assert step > 1e-4 and np.abs(host(x_new) - dense).max() < 2e-4 + 2e-3 * step
    # known answer of the reference's differencing test
    xk = dev([[0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7],
              [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07], [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07]])  # fmt: skip
    if name == "panda":
        pk = _full_params(alpha_differencing=1.0, use_virtual_configs=False, use_self_collisions=False, use_env_collisions=False)
        jk, rk = LmResidualFns.get_r_and_J(pk, rb, xk, rb.forward_kinematics(xk))
        expected = [0.09, 0.18, 0.27, 0.36, 0.45, 0.54, 0.63, -0.09, -0.18, -0.27, -0.36, -0.45, -0.54, -0.63] + [0.0] * 7
        np.testing.assert_allclose(host(rk.get_r())[:, 0], expected, atol=1e-6)
        Jk = host(jk.get_J())
        assert Jk.shape == (21, 28) and Jk[0, 0] == 1.0 and Jk[0, 7] == -1.0 and np.count_nonzero(Jk) == 42
    rb.set_obstacles([], [])


def test_env_colliding_links_capsule(robots):
    """env_colliding_links_capsule (cppflow/collision_detection.py:135-145): the links it names are the capsules whose
    oracle distance to an obstacle is negative."""
    from cppflow_amd.collision_detection import env_colliding_links_capsule
    from cppflow_amd.data_type_utils import problem_from_arrays

    rb, orc = robots["panda"], H.oracle64("panda")
    target = H.f32(orc.fk(H.random_configs("panda", 4, seed=1)))
    problem = problem_from_arrays(rb, target, [(0.2, 0.3, 0.4, 0.15, 0.15, 0.15), (-0.25, 0.3, 0.75, 0.15, 0.15, 0.15)], device=DEV)
    lo, hi = H.box_corners([c for c, _ in H.PANDA_2CUBES], [T for _, T in H.PANDA_2CUBES])
    names = list(rb._collision_capsules_by_link.keys())
    assert len(names) == rb.n_capsules
    cand = H.random_configs("panda", 500, seed=2)
    n_hit = 0
    for q in cand[:60]:
        want = set()
        for o in range(2):
            d = orc.env_dists(q[None], lo[o], hi[o])[0]
            want |= {names[i] for i in np.flatnonzero(d < -1e-6)}
            borderline = np.abs(d) < 1e-6
        got = set(env_colliding_links_capsule(problem, dev(q)))
        if not borderline.any():
            assert want <= got and len(got - want) == 0
        n_hit += len(got) > 0
    assert n_hit > 0
    rb.set_obstacles([], [])


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_mjacs_and_dp_search_slow(robots, name):
    """_get_mjacs (cppflow/search.py:100-125) against its torch definition on the same device, and the min-max recurrence
    written with that tensor (search.py:156-159) against cppf_dp_search, which never builds it; dp_search_slow
    (search.py:55-97) is the same recurrence and returns the same path."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.search import _get_mjacs, dp_search, dp_search_slow, q_costs_external

    rb, ch = robots[name], H.chain(name)
    rng = np.random.RandomState(4)
    k, T, d = 23, 17, rb.ndof
    q = dev(H.f32(rng.uniform(ch.lo, ch.hi, size=(k, T, d))))
    got = _get_mjacs(q, rb, 5.0)
    dqs = q[:, 1:, :].unsqueeze(1) - q[:, :-1, :].unsqueeze(0)
    if rb.has_prismatic_joints:
        dqs[:, :, :, rb.prismatic_joint_idxs] *= 5.0
    want = torch.abs(torch.remainder(dqs + torch.pi, 2 * torch.pi) - torch.pi).max(dim=3).values
    assert got.shape == (k, k, T - 1) and float((got - want).abs().max()) < 1e-6
    target = H.f32(H.oracle64(name).fk(H.random_configs(name, T, seed=3)))
    problem = problem_from_arrays(rb, target, [(0.2, 0.3, 0.4, 0.15, 0.15, 0.15)], device=DEV)
    cost, _, _, _ = q_costs_external(rb, q, problem)
    costs = torch.zeros((k, T), device=DEV)
    costs[:, 0] = cost[:, 0]
    for t in range(1, T):
        costs[:, t] = torch.maximum(got[:, :, t - 1], costs[:, t - 1][None, :]).min(dim=1).values + cost[:, t]
    _, _, table = rb.dp_search(q, cost)
    assert torch.equal(table.T.contiguous(), costs)
    fast = dp_search(rb, q, None, None, q_costs=cost)
    slow = dp_search_slow(problem, [q[i] for i in range(k)], verbosity=0)
    assert torch.equal(fast, slow)
    rb.set_obstacles([], [])


def test_add_search_path_mjac(robots):
    """add_search_path_mjac (cppflow/planners.py:50-73) against the same quantities computed with numpy."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.planners import add_search_path_mjac

    for name in ("fetch", "panda"):
        rb, ch = robots[name], H.chain(name)
        rng = np.random.RandomState(2)
        base = rng.uniform(ch.lo, ch.hi)
        q = H.f32(np.clip(base[None] + np.cumsum(0.02 * rng.randn(30, rb.ndof), axis=0), ch.lo + 1e-3, ch.hi - 1e-3))
        problem = problem_from_arrays(rb, H.f32(H.oracle64(name).fk(q)), device=DEV)
        info = {}
        add_search_path_mjac(info, problem, dev(q))
        dq = np.diff(q, axis=0)
        rev = [j for j in range(rb.ndof) if ch.jtype[j] == 0]
        pris = [j for j in range(rb.ndof) if ch.jtype[j] == 1]
        want_deg = np.rad2deg(np.abs(np.remainder(dq[:, rev] + np.pi, 2 * np.pi) - np.pi).max())
        assert info["search_path_mjac-deg"] == pytest.approx(want_deg, rel=1e-5)
        assert info["search_path_mjac-cm"] == pytest.approx(100 * np.abs(dq[:, pris]).max() if pris else 0.0, rel=1e-5)
        margin = np.minimum(np.abs(q - ch.lo).min(0), np.abs(q - ch.hi).min(0))
        if pris:
            assert info["search_path_min_dist_to_jlim_cm"] == pytest.approx(100 * margin[0], rel=1e-4)
            assert info["search_path_min_dist_to_jlim_deg"] == pytest.approx(np.rad2deg(margin[1:].min()), rel=1e-4)
        else:
            assert info["search_path_min_dist_to_jlim_cm"] == -1
            assert info["search_path_min_dist_to_jlim_deg"] == pytest.approx(np.rad2deg(margin.min()), rel=1e-4)


def test_search_avoids_joint_limits(robots):
    """tests/search_test.py:59-78 of the reference (fetch_arm__s__truncated, k = 20 candidates through PlannerSearcher): the
    path dp_search returns stays more than 1 degree away from every joint limit, because candidates inside the 1.5 degree
    padding cost 100 (cppflow/search.py:14-21); and its cost is minimal among the candidates' own paths."""
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.data_types import PlannerSettings
    from cppflow_amd.planners import LmIkSeedProvider, PlannerSearcher

    rb = robots["fetch_arm"]
    z = np.load(os.path.join(GOLDEN, "reference_paths.npz"))
    problem = problem_from_arrays(rb, z["fetch_arm__s__truncated"], device=DEV)
    searcher = PlannerSearcher(PlannerSettings(k=20, tmax_sec=30.0, anytime_mode_enabled=False, verbosity=0), rb,
                               LmIkSeedProvider(seed=3))  # fmt: skip
    plan = searcher.generate_plan(problem).plan
    eps = np.deg2rad(1.0)
    q = plan.q_path
    for j, (l, u) in enumerate(rb.actuated_joints_limits):
        assert not bool((q[:, j] < l + eps).any()) and not bool((q[:, j] > u - eps).any()), j
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)
