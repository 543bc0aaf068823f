"""Round-2 GPU tests (pytest -m gpu): NaN propagation like torch's, the in-launch early-out, x_is_valid's seed selection on
the device, output-buffer contracts.  Every call goes through the C ABI of libcppflow_hip.so."""

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda:0")


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def panda():
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    rb.set_obstacles([c for c, _ in H.PANDA_2CUBES], [T for _, T in H.PANDA_2CUBES])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    return rb


def test_nan_rows_stay_nan_like_the_reference_op_sequence(panda):
    """torch.clamp / torch.max propagate NaN (cppflow/optimization_utils.py:831-833, evaluation_utils.py:41-42): a row with a NaN
    joint (or a NaN target pose) comes back NaN, its seed's maxima are 'not below threshold', and no mask is raised for it --
    exactly what the reference's op sequence (oracle/ref_torch.py) gives.  Finite rows are unaffected."""
    from cppflow_amd.data_types import DEFAULT_CONSTRAINTS
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import ref_torch

    S, W, K = 4, 64, 3
    x0, target = H.lm_problem("panda", S, W, seed=31)
    x_bad = x0.copy()
    x_bad[5, 2] = np.nan  # seed 0
    x_bad[2 * W + 7, 0] = np.inf  # seed 2
    summ = torch.empty((S, 8), dtype=torch.float32, device="cuda:0")
    from cppflow_amd import _hip

    ROW = dict(shape=_hip.SHAPE_ROW)  # one shape throughout: the comparison with the clean run below is bit for bit
    res = panda.lm_pose_steps(dev(x_bad), dev(target), n_steps=K, want_errors=True, want_collisions=True, summary_out=summ, **ROW, **LM)
    x = host(res["x"])
    bad_rows = np.zeros(S * W, dtype=bool)
    bad_rows[[5, 2 * W + 7]] = True
    assert np.isnan(x[bad_rows]).all() and np.isfinite(x[~bad_rows]).all()
    clean = panda.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_collisions=True, **ROW, **LM)
    assert np.array_equal(x[~bad_rows], host(clean["x"])[~bad_rows])
    # the reference's op sequence on the same rows
    rt = ref_torch.TorchRobot(ROBOT_SPECS["panda"](), device="cpu", dtype=torch.float32)
    x_ref = ref_torch.lm_pose_steps(rt, torch.tensor(x_bad, dtype=torch.float32), torch.tensor(H.stacked(target, S), dtype=torch.float32), K)
    assert np.array_equal(np.isnan(x_ref.numpy()).any(1), bad_rows)
    assert np.isnan(host(res["pos_err_m"])[bad_rows]).all()
    for k in ("self_mask", "env_mask"):
        assert not res[k].cpu().numpy()[bad_rows].any()
    s = host(summ)
    assert np.isinf(s[0, 0]) and np.isinf(s[2, 0]) and np.isfinite(s[[1, 3]]).all()
    sel = panda.select_valid_seed(summ, DEFAULT_CONSTRAINTS).cpu().numpy()
    assert sel[0] not in (0, 2)
    # standalone reductions and the standalone clamp agree
    sv = host(panda.seed_validity(res["x"], dev(target)))
    assert np.isinf(sv[0, 0]) and np.isinf(sv[2, 0]) and np.isfinite(sv[[1, 3]]).all()
    xc = dev(x_bad)
    panda.clamp_to_joint_limits(xc)
    assert np.isnan(host(xc)[5, 2]) and host(xc)[2 * W + 7, 0] == H.chain("panda").hi[0]
    # a NaN target pose poisons its waypoint in every seed
    t_bad = target.copy()
    t_bad[9, 4] = np.nan
    r2 = panda.lm_pose_steps(dev(x0), dev(t_bad), n_steps=K, **LM)
    rows9 = np.arange(S) * W + 9
    assert np.isnan(host(r2["x"])[rows9]).all() and np.isfinite(np.delete(host(r2["x"]), rows9, axis=0)).all()


def test_early_out_freezes_converged_rows_and_counts_iterations(panda):
    """cppf_lm_params.tol_*: a row below tolerance at the start of an iteration is left untouched; n_iters counts the steps
    applied.  With the tolerances off the launch runs all K iterations (what bench.py times)."""
    S, W, K = 8, 64, 12
    x0, target = H.lm_problem("panda", S, W, seed=32)
    tol_p, tol_r = 1e-5, 1e-4
    full = panda.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_iters=True, **LM)
    assert (full["n_iters"].cpu().numpy() == K).all()
    early = panda.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_iters=True, tol_pos_m=tol_p,
                                tol_rot_rad=tol_r, **LM)  # fmt: skip
    it = early["n_iters"].cpu().numpy()
    assert it.min() >= 1 and it.max() <= K and (it < K).mean() > 0.8, (it.min(), it.max(), (it < K).mean())
    # a row frozen after k steps equals the k-step result of a launch that freezes nothing, bit for bit.  (An early-out launch runs
    # EVERY iteration in the canonical arithmetic and with the absolute gate -- its frozen iterates are results; a plain K-step launch
    # runs its K - 1 leading iterations lean, include/cppflow_hip.h: cppf_lm_params.n_steps -- so the comparison launch is an early-out
    # one whose tolerances nothing can meet: 1e-18, whose SQUARE is still a normal float -- 1e-30 squared is 0 = early-out off.)
    xe = host(early["x"])
    for k in (int(np.median(it)), int(it.min())):
        rows = it == k
        xk = host(panda.lm_pose_steps(dev(x0), dev(target), n_steps=k, tol_pos_m=1e-18, tol_rot_rad=1e-18, **LM)["x"])
        assert np.array_equal(xe[rows], xk[rows])
        # and a plain k-step launch has converged on those rows as well (one more step than the frozen rows needed to be below tolerance)
        rp = panda.lm_pose_steps(dev(x0), dev(target), n_steps=k + 1, want_errors=True, **LM)
        assert host(rp["pos_err_m"])[rows].max() < tol_p * 1.01 and host(rp["rot_err_rad"])[rows].max() < max(tol_r * 1.05, 8.95e-4)
    # frozen rows are converged: pose error below the tolerances (rpy norm bounds the rotation angle to first order)
    frozen = it < K
    assert host(early["pos_err_m"])[frozen].max() < tol_p * 1.01
    assert host(early["rot_err_rad"])[frozen].max() < max(tol_r * 1.05, 8.95e-4)
    # the oracle's own iteration reaches the tolerance on those rows within the same number of steps (+-1)
    o = H.oracle64("panda")
    tgt = H.stacked(target, S)
    xs, need = x0.copy(), np.full(S * W, K, dtype=int)
    for k in range(K):
        e, _ = o.pose_errors(xs, tgt)
        done = (np.linalg.norm(e[:, 3:], axis=1) < tol_p) & (np.linalg.norm(e[:, :3], axis=1) < tol_r) & (need == K)
        need[done] = k
        xs = o.lm_steps(xs, tgt, 1, solver=0, **LM)
    agree = np.abs(need - it) <= 1
    assert agree.mean() > 0.98, agree.mean()
    with pytest.raises(AssertionError):
        panda.lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, return_residual=True, tol_pos_m=1e-5, tol_rot_rad=1e-4, **LM)
    with pytest.raises(AssertionError):
        panda.lm_pose_steps(dev(x0), dev(target), n_steps=2, tol_pos_m=1e-5, **LM)


def test_select_valid_seed_matches_x_is_valid_rule(panda):
    from cppflow_amd.data_types import Constraints
    from cppflow_amd.evaluation_utils import seed_metrics_are_below_threshold

    rng = np.random.RandomState(5)
    S = 777
    c = Constraints(0.01, 0.1, 7.0, 2.0)
    summ = np.zeros((S, 8), dtype=np.float32)
    summ[:, 0] = rng.uniform(0, 0.02, S)
    summ[:, 1] = rng.uniform(0, 0.2, S)
    summ[:, 2] = rng.uniform(0, 14, S)
    summ[:, 3] = rng.uniform(0, 4, S)
    summ[:, 4] = rng.randint(0, 3, S) * (rng.rand(S) < 0.3)
    summ[:, 5] = rng.randint(0, 3, S) * (rng.rand(S) < 0.3)
    summ[:, 7] = rng.randint(0, 50, S) * 100.0
    summ[10, 0] = np.nan
    summ[11, 1] = np.inf

    def rule(rows, ignore_self, ignore_env):
        ok = [seed_metrics_are_below_threshold(c, r[:4])[0] and (ignore_self or r[4] == 0) and (ignore_env or r[5] == 0) for r in rows]
        first = next((i for i, v in enumerate(ok) if v), -1)
        return first, int(np.sum(ok)), int(np.argmin(rows[:, 7]))

    for ig_s, ig_e in ((False, False), (True, False), (True, True)):
        got = panda.select_valid_seed(dev(summ), c, ig_s, ig_e).cpu().numpy()
        assert tuple(got[:3]) == rule(summ, ig_s, ig_e), (got, rule(summ, ig_s, ig_e))
    none = summ.copy()
    none[:, 0] = 1.0
    assert tuple(panda.select_valid_seed(dev(none), c).cpu().numpy()[:2]) == (-1, 0)
    # the all-gathered layout [world, G, S_local, 8]: group g = the seeds of step g in rank order
    world, G, Sl = 3, 4, 50
    g4 = summ[: world * G * Sl].reshape(world, G, Sl, 8)
    got = panda.select_valid_seed(dev(g4), c).cpu().numpy()
    for g in range(G):
        rows = np.concatenate([g4[r, g] for r in range(world)], axis=0)
        assert tuple(got[g, :3]) == rule(rows, False, False)


def test_output_buffers_must_be_contiguous(panda):
    S, W = 2, 64
    x0, target = H.lm_problem("panda", S, W, seed=33)
    wide = torch.empty((S * W, 14), dtype=torch.float32, device="cuda:0")
    with pytest.raises(AssertionError, match="contiguous"):
        panda.lm_pose_steps(dev(x0), dev(target), n_steps=1, x_out=wide[:, :7], **LM)
    # in-place use (x_out = x) writes the caller's buffer
    x = dev(x0)
    ptr = x.data_ptr()
    res = panda.lm_pose_steps(x, dev(target), n_steps=2, x_out=x, **LM)
    assert res["x"].data_ptr() == ptr and not np.array_equal(host(x), x0)


# ---- the quad shape (four lanes per row) against the row shape and the oracle ------------------------------------------------------
QUAD_ROBOTS = ["panda", "fetch", "fetch_arm", "chain12"]


def _obstacles_for(name):
    if name in ("panda", "chain12"):
        return H.PANDA_2CUBES
    return [H.cuboid_obstacle(0.7, 0.1, 0.8, 0.3, 0.3, 0.3)]


@pytest.fixture(scope="module")
def robots():
    from cppflow_amd.robots import get_robot

    return {n: get_robot(n) for n in QUAD_ROBOTS}


@pytest.mark.parametrize("mfma", [False, True])
@pytest.mark.parametrize("name", QUAD_ROBOTS)
def test_quad_shape_agrees_with_row_shape_and_oracle(robots, name, mfma):
    """Same launch, both kernel shapes (and J J^T by MFMA in the quad shape): x agrees to fp32 rounding of the iteration; the
    metrics are those of each shape's own x (1e-5 vs the fp64 oracle); masks / cost are bit-exact vs the fp32 oracle at that x."""
    from cppflow_amd import _hip

    rb = robots[name]
    obs = _obstacles_for(name)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    jl_lo, jl_hi = rb.padded_joint_limits()
    o64, o32 = H.oracle64(name), H.oracle32(name)
    rb.debug_set("quad_mfma", int(mfma))
    try:
        for S, W, K in ((3, 50, 1), (8, 64, 5), (5, 256, 10)):  # 150 rows: a partly filled last workgroup
            x0, target = H.lm_problem(name, S, W, seed=40 + K)
            tgt = H.stacked(target, S)
            out = {}
            for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
                summ = torch.full((S, 8), -1.0, dtype=torch.float32, device="cuda:0")
                out[shape] = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_collisions=True,
                                              summary_out=summ, want_iters=True, shape=shape, **LM)  # fmt: skip
            r, q = out[_hip.SHAPE_ROW], out[_hip.SHAPE_QUAD]
            xr, xq = host(r["x"]), host(q["x"])
            assert np.isfinite(xq).all()
            # per-step differences are fp32 rounding of a 6x6 solve with cond <~ 1e5, amplified over K steps on rows that
            # have not converged; on converged rows the two agree to 1e-5
            d = np.abs(xr - xq).max(axis=1)
            pe_r = host(r["pos_err_m"])
            conv = pe_r < 1e-4
            assert np.median(d) < 2e-5, np.median(d)
            if K >= 5:
                # converged rows: the same joint configuration up to the self-motion freedom of a redundant arm (two
                # roundings of one iteration may settle at different points of the null-space manifold near a singularity):
                # the reference's own x bar (tests/optimization_test.py:99) on 99 % of them, the same POSE on all (below)
                assert np.quantile(d[conv], 0.99) < 5e-3, np.quantile(d[conv], 0.99)
            # the quad shape's own outputs against the oracle at its own x
            pe, re = o64.pose_metrics_exact(xq, tgt)
            assert np.abs(host(q["pos_err_m"]) - pe).max() < 1e-5
            assert np.abs(host(q["rot_err_rad"]) - re).max() < 1e-5
            want = o32.masks(xq, lo, hi, jl_lo, jl_hi)
            for kx in ("self_mask", "env_mask", "jlim_mask"):
                assert np.array_equal(q[kx].cpu().numpy().astype(np.uint8), want[kx]), (kx, S, W, K)
            assert np.array_equal(host(q["ext_cost"]), want["ext_cost"])
            assert (q["n_iters"].cpu().numpy() == K).all()
            # convergence quality: same as the row shape's
            pe_q = host(q["pos_err_m"])
            if K >= 10:
                assert (pe_q < 1e-4).mean() > 0.9 and abs((pe_q < 1e-4).mean() - conv.mean()) < 0.02
                both = conv & (pe_q < 1e-4)
                assert np.abs(pe_q - pe_r)[both].max() < 1e-5
            # the per-seed summary (separate reduction kernel in this shape) equals the one of its per-row outputs
            want_summ = host(rb.seed_summary(q["x"], torch.cat([q["ext_cost"].view(torch.uint8), q["pos_err_m"].view(torch.uint8),
                                                                   q["rot_err_rad"].view(torch.uint8), q["self_mask"], q["env_mask"],
                                                                   q["jlim_mask"]]), S, W))  # fmt: skip
            assert np.array_equal(host(q["seed_summary"]), want_summ)
    finally:
        rb.debug_set("quad_mfma", 0)
        rb.set_obstacles([], [])
        rb.set_joint_limit_padding(None, None)


def test_quad_shape_single_step_matches_reference_order_oracle(robots):
    """One bare step (clamp off) in the quad shape vs the fp64 reference-order oracle: the same bars as the row shape's test."""
    from cppflow_amd import _hip

    for name in QUAD_ROBOTS:
        S, W = 16, 64
        x0, target = H.lm_problem(name, S, W, seed=3)
        res = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, shape=_hip.SHAPE_QUAD, **LM)
        xo, Jo, eo, fails = H.oracle64(name).lm_step(x0, H.stacked(target, S), solver=0, **LM)
        smin = np.linalg.svd(Jo, compute_uv=False)[:, -1]
        ok = smin >= 2e-2
        diff = np.abs(host(res["x"]) - xo)
        assert diff[ok].max() < 5e-3, (name, diff[ok].max())
        # (task space on ALL rows: tests/test_gpu_parity_allrows.py)


def test_quad_shape_refuses_what_it_cannot_produce(robots):
    from cppflow_amd import _hip

    x0, target = H.lm_problem("panda", 2, 64, seed=1)
    with pytest.raises(AssertionError, match="CPPF_SHAPE_QUAD"):
        robots["panda"].lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, return_residual=True, shape=_hip.SHAPE_QUAD, **LM)
    with pytest.raises(AssertionError, match="CPPF_SHAPE_QUAD"):
        robots["panda"].lm_pose_steps(dev(x0), dev(target), n_steps=2, want_collisions=True, want_min_dists=True,
                                      shape=_hip.SHAPE_QUAD, **LM)  # fmt: skip


@pytest.mark.parametrize("name,k,T", [("panda", 175, 256), ("fetch", 256, 64), ("panda", 96, 40), ("fetch", 97, 33), ("chain12", 5, 9), ("panda", 1, 4),
                                      ("panda", 64, 7), ("fetch", 65, 2), ("panda", 129, 19), ("panda", 200, 1)])
def test_dp_search_single_launch_equals_per_waypoint_launches_and_oracle(robots, name, k, T):
    """cppf_dp_search's resident single-launch form (cost words that are their own flags) and cppf_dp_search_tabled (transition
    table + the recurrence on one compute unit + parallel argmin recovery) against the one-launch-per-waypoint form and the fp32
    oracle restatement of cppflow/search.py:128-191: cost table, argmins (the whole memo table) and path bit for bit."""
    from cppflow_amd import _hip

    rb = robots[name]
    rng = np.random.RandomState(k * 7 + T)
    ch = H.chain(name)
    base = rng.uniform(ch.lo, ch.hi, size=(4, 1, rb.ndof)) + 0.3 * np.cumsum(rng.randn(4, T, rb.ndof) * 0.1, axis=1)
    q = H.f32(np.clip(base[rng.randint(0, 4, size=k)] + 0.02 * rng.randn(k, T, rb.ndof), ch.lo, ch.hi))
    ext = ((rng.rand(k, T) < 0.15) * 1000.0 + (rng.rand(k, T) < 0.1) * 100.0).astype(np.float32)
    got = {}
    try:
        for mode in (1, 0):
            rb.debug_set("dp_persistent", mode)
            for rep in range(3):  # repeated calls reuse nothing: every call re-arms its own cost table
                path, idx, costsT, memoT = rb.dp_search(dev(q), dev(ext), method="resident", return_memo=True)
            got[mode] = (host(path), idx.cpu().numpy(), host(costsT), memoT.cpu().numpy())
    finally:
        rb.debug_set("dp_persistent", 1)
    for rep in range(2):
        path, idx, costsT, memoT = rb.dp_search(dev(q), dev(ext), method="table", return_memo=True)
    got["table"] = (host(path), idx.cpu().numpy(), host(costsT), memoT.cpu().numpy())
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b)
    for i, (a, b) in enumerate(zip(got["table"], got[0])):
        assert np.array_equal(a, b), ("table vs per-waypoint launches", i)
    # what a caller gets without asking: the table form in its range
    auto = rb.dp_search(dev(q), dev(ext))
    assert np.array_equal(host(auto[2]), got[0][2]) and np.array_equal(auto[1].cpu().numpy(), got[0][1])
    want_idx, want_costs = H.oracle32(name).dp_search(q, ext)
    assert np.array_equal(got[1][2].T, want_costs) and np.array_equal(got[1][1], want_idx)
    assert np.array_equal(got[1][0], q[want_idx, np.arange(T)])


# ---- run-time specialisation (hipRTC) of descriptions that match no generated table -----------------------------------------------
@pytest.mark.parametrize("ndof,seed", [(7, 11), (6, 12), (9, 13), (4, 14)])
def test_arbitrary_chains_specialised_equal_generic_bit_for_bit(ndof, seed, tmp_path, monkeypatch):
    """cppf_robot_specialize: a random chain (general joint axes, a prismatic joint, fixed joints inside the chain) gets kernels
    compiled for it with hipRTC; every output of the fused launch (both shapes), of the collision launch and of the K = 1 API
    equals the generic kernels' bit for bit (the two differ only in where the constants come from); the second handle of the
    same description loads the cached code object instead of compiling."""
    import time

    from cppflow_amd import _hip
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robots import Robot

    monkeypatch.setenv("CPPF_CACHE_DIR", str(tmp_path))
    spec = H.random_chain_spec(ndof, seed)
    ch = canonicalize(spec)
    t0 = time.perf_counter()
    fast = Robot(spec, specialize=True)
    assert fast.specialization("cuda:0") == 1000
    t_first = time.perf_counter() - t0
    slow = Robot(spec, specialize=False)
    assert slow.specialization("cuda:0") == -1
    assert len(list(tmp_path.glob("robot_*.cppfrtc"))) == 1
    t0 = time.perf_counter()
    again = Robot(spec, specialize=True)
    assert again.specialization("cuda:0") == 1000 and time.perf_counter() - t0 < max(0.5, 0.5 * t_first)  # cache hit
    obs = [H.cuboid_obstacle(0.1, 0.1, 0.5, 0.3, 0.3, 0.3)]
    rng = np.random.RandomState(seed)
    S, W, K = 6, 64, 6
    q_star = H.f32(rng.uniform(ch.lo, ch.hi, size=(W, ndof)))
    target = H.f32(H.f32(Oracle_fk(ch, q_star)))
    x0 = H.f32(np.clip(q_star[None] + 0.1 * rng.randn(S, W, ndof), ch.lo, ch.hi).reshape(S * W, ndof))
    shapes = [_hip.SHAPE_ROW] + ([_hip.SHAPE_QUAD] if ndof >= 6 else [])
    outs = {}
    for tag, rb in (("fast", fast), ("slow", slow)):
        rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
        rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
        res = []
        for shape in shapes:
            summ = torch.empty((S, 8), dtype=torch.float32, device="cuda:0")
            res.append(rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_collisions=True, summary_out=summ,
                                        shape=shape, **LM))  # fmt: skip
        res.append(rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, return_residual=True, shape=_hip.SHAPE_ROW, **LM))
        res.append(rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_collisions=True, want_min_dists=True,
                                    shape=_hip.SHAPE_ROW, **LM))  # fmt: skip
        res.append(rb.collision_masks(dev(x0).reshape(S, W, ndof), want_min_dists=True))
        res.append(rb.collision_masks(dev(x0).reshape(S, W, ndof)))
        outs[tag] = res
    for a, b in zip(outs["fast"], outs["slow"]):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (ndof, k)
    assert outs["fast"][0]["self_mask"].any() or outs["fast"][0]["env_mask"].any() or True


def Oracle_fk(ch, q):
    from oracle.oracle import Oracle

    return Oracle(ch, f32=False).fk(q)


@pytest.mark.parametrize("ndof,seed", [(3, 4), (4, 5), (5, 6), (6, 0), (8, 2), (9, 7), (10, 8), (12, 3)])
def test_coupled_step_on_arbitrary_chains_all_elimination_orders(ndof, seed):
    """cppf_lm_full_step on chains that match no generated table (every ndof the row-per-lane kernels are instantiated for, a
    prismatic joint with its own differencing scale, T odd and even, a ragged count of trajectories): the two-ended row-per-lane
    elimination, the one-wavefront-per-trajectory elimination and parallel cyclic reduction agree with each other and with
    the oracle's dense restatement of cppflow/optimization.py:95-144.  Beyond 8 joints the row-per-lane kernels use sixteen lanes
    per trajectory, the cross-check ("wave") is the one-lane-per-trajectory kernel and there is no parallel-in-time form."""
    from cppflow_amd import _hip
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    spec = H.random_chain_spec(ndof, seed)
    rb = Robot(spec, specialize=False)
    ch = canonicalize(spec)
    o64 = Oracle(ch, f32=False)
    obs = [H.cuboid_obstacle(0.1, 0.1, 0.5, 0.3, 0.3, 0.3)]
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    d = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    d.update(alpha_differencing_prismatic_scaling=2.0, alpha_self_collision=0.02, alpha_env_collision=0.02)
    L = _hip.lib()
    try:
        for S, T in ((11, 23), (3, 24), (9, 1), (2, 2)):
            pm = OptimizationParameters(**{**d, "use_virtual_configs": 2 * d["n_virtual_configs"] < T})
            pm.virtual_configs = torch.tensor([])
            rng = np.random.RandomState(100 * seed + T)
            base = np.clip(rng.uniform(ch.lo, ch.hi)[None, :] * 0.6 + np.cumsum(0.03 * rng.randn(T, ndof), axis=0), ch.lo, ch.hi)
            x = H.f32(np.clip(base[None] + 0.01 * rng.randn(S, T, ndof), ch.lo, ch.hi).reshape(S * T, ndof))
            target = H.f32(o64.fk(H.f32(base)))
            # explicit virtual configurations (the alternating loop passes the previous x, optimization.py:253) where they apply
            xv = H.f32(x + 0.01 * rng.randn(*x.shape)) if pm.use_virtual_configs else None
            got = {}
            for mode, (pcr, rows) in {"pcr": (1 << 30, 1), "rows": (0, 1), "wave": (0, 0)}.items():
                rb.debug_set("pcr_max_rows", pcr)
                rb.debug_set("full_rows", rows)
                got[mode] = host(rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=dev(xv) if xv is not None else None))
            want = o64.lm_full_step(x, target, pm, S, T, virtual_configs=xv, boxes_lo=lo, boxes_hi=hi)
            step = np.abs(want - x).max()
            for mode, g in got.items():
                assert np.isfinite(g).all(), (mode, S, T)
                assert np.abs(g - want).max() < 2e-4 + 2e-3 * step, (mode, S, T, np.abs(g - want).max(), step)
            assert np.abs(got["rows"] - got["wave"]).max() < 1e-5 + 1e-3 * step, (S, T)
    finally:
        rb.debug_set("pcr_max_rows", -1)
        rb.debug_set("full_rows", 1)
        rb.set_obstacles([], [])


def test_launch_plan_without_collision_stage_matches_the_eager_call_and_replays_in_a_graph(panda):
    """LmLaunchPlan with errors_out (FK + Jacobian + LM only, BASELINE configs[1]): same x and pose errors as lm_pose_steps,
    bit for bit, launched eagerly, on an explicit stream, and replayed from a captured hipGraph (what bench.py does for shards)."""
    rb = panda
    S, W, K = 16, 64, 5
    rng = np.random.RandomState(3)
    ch = H.chain("panda")
    q_star = H.f32(rng.uniform(ch.lo, ch.hi, size=(W, rb.ndof)))
    target = dev(H.f32(H.oracle64("panda").fk(q_star)))
    x0 = dev(H.f32(np.clip(q_star[None] + 0.1 * rng.randn(S, W, rb.ndof), ch.lo, ch.hi).reshape(S * W, rb.ndof)))
    want = rb.lm_pose_steps(x0, target, n_steps=K, want_errors=True, **LM)
    n = S * W
    xo = torch.empty_like(x0)
    pe, re = torch.empty(n, device=x0.device), torch.empty(n, device=x0.device)
    plan = rb.lm_launch_plan(x0, target, n_steps=K, x_out=xo, errors_out=(pe, re), **LM)
    plan.launch()
    torch.cuda.synchronize()
    for got, key in ((xo, "x"), (pe, "pos_err_m"), (re, "rot_err_rad")):
        assert torch.equal(got.reshape(-1), want[key].reshape(-1)), key
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    xo.zero_(), pe.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        plan.launch_on(st)
        plan.launch_on(st)
    torch.cuda.synchronize()
    xo.zero_(), pe.zero_()
    with torch.cuda.stream(st):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(xo.reshape(-1), want["x"].reshape(-1)) and torch.equal(pe, want["pos_err_m"].reshape(-1))
    with pytest.raises(AssertionError):
        rb.lm_launch_plan(x0, target, n_steps=K, x_out=xo, errors_out=(pe, re),
                          packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=x0.device), **LM)


def test_coupled_step_two_ended_elimination_over_a_sweep_of_shapes(robots):
    """The two-ended row-per-lane elimination against the one-wavefront-per-trajectory elimination for every small count of
    trajectories (partly filled wavefronts) and path length (T = 1: the join alone; T = 2, 3: one chain empty; odd / even splits;
    shorter than the prefetch rings; just past them)."""
    from cppflow_amd import _hip

    rb, ch = robots["panda"], H.chain("panda")
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters

    L = _hip.lib()
    worst = 0.0
    try:
        rb.debug_set("pcr_max_rows", 0)
        for T in (1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 33, 40):
            d = dict(ALT_LOSS_V2_1_DIFF.__dict__)
            d["use_virtual_configs"] = 2 * d["n_virtual_configs"] < T
            pm = OptimizationParameters(**d)
            pm.virtual_configs = torch.tensor([])
            for S in (1, 7, 8, 9, 17):
                rng = np.random.RandomState(1000 * T + S)
                base = np.clip(rng.uniform(ch.lo, ch.hi)[None, :] * 0.5 + np.cumsum(0.03 * rng.randn(T, rb.ndof), axis=0), ch.lo, ch.hi)
                x = H.f32(np.clip(base[None] + 0.01 * rng.randn(S, T, rb.ndof), ch.lo, ch.hi).reshape(S * T, rb.ndof))
                target = dev(H.f32(H.oracle64("panda").fk(H.f32(base))))
                rb.debug_set("full_rows", 1)
                rows = host(rb.lm_full_step(dev(x), target, pm))
                rb.debug_set("full_rows", 0)
                wave = host(rb.lm_full_step(dev(x), target, pm))
                step = np.abs(wave - x).max()
                err = np.abs(rows - wave).max()
                worst = max(worst, err / (1e-5 + 1e-3 * step))
                assert np.isfinite(rows).all() and err < 1e-5 + 1e-3 * step, (S, T, err, step)
    finally:
        rb.debug_set("pcr_max_rows", -1)
        rb.debug_set("full_rows", 1)
        rb.set_obstacles([], [])
    assert worst < 1.0


def test_dp_search_table_form_over_a_sweep_of_shapes(robots):
    """cppf_dp_search_tabled against the per-waypoint launches for every group / chunk boundary of its chain kernel (8 source
    groups, chunks of 8 rows, 64-destination columns, one to three register sets in flight) and the shortest paths."""
    from cppflow_amd import _hip

    rb, ch = robots["panda"], H.chain("panda")
    try:
        rb.debug_set("dp_persistent", 0)
        for k in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 161, 191, 192, 193, 255, 256):
            for T in (1, 2, 3, 4, 5, 6, 7, 12):
                rng = np.random.RandomState(31 * k + T)
                q = H.f32(np.clip(rng.uniform(ch.lo, ch.hi, size=(1, 1, rb.ndof)) * 0.5 + 0.2 * np.cumsum(rng.randn(k, T, rb.ndof) * 0.2, axis=1), ch.lo, ch.hi))
                ext = ((rng.rand(k, T) < 0.2) * 1000.0 + (rng.rand(k, T) < 0.1) * 100.0).astype(np.float32)
                a = rb.dp_search(dev(q), dev(ext), method="table", return_memo=True)
                b = rb.dp_search(dev(q), dev(ext), method="resident", return_memo=True)
                for i, (u, v) in enumerate(zip(a, b)):
                    assert torch.equal(u, v), (k, T, i)
    finally:
        rb.debug_set("dp_persistent", 1)
