"""Parity on ALL rows, without a conditioning filter (VERDICT r1 "What's weak" 1-4; pytest -m gpu).

What can be asserted on every row, and why the bars differ by solver precision (DESIGN.md 5.1 has the measured tables):

  * joint space is the wrong place to compare: for ndof >= 7 the reference's normal matrix is rank-deficient and its own fp32
    LU carries 0.02-0.04 rad of null-space noise.  The meaningful quantity is the TASK-space difference of
    the step, |J_s (x_gpu - x_fp64)|, with J_s the scaled Jacobian of the fp64 reference-order oracle.
  * CPPF_SOLVER_F64 (every row in double precision): <= 2e-5 on every row, near-singular ones included, and the pose error
    after ONE step within 1e-5 of the oracle's wherever the oracle's own step is not a jump through a singularity.
  * CPPF_SOLVER_AUTO (the DEFAULT: fp32 with the conditioning-gated double-precision redo, csrc/kernels_chain.h): on every
    robot and both kernel shapes the step is NEVER WORSE THAN THE REFERENCE'S OWN fp32 ARITHMETIC (oracle/lmik_oracle.c, LU with
    partial pivoting in the reference's operation order):  max <= max(max of the reference-order fp32, 1e-4)  and
    99th percentile <= 2 x the reference-order fp32's (round 2's fp32-only solve was 4-100x farther out on
    the 7-11 % near-singular rows of the 7-DoF arms).
  * CPPF_SOLVER_F32 (no gate; kept as an option): sane on the well-conditioned rows.
  * after K steps the comparison is on the pose error: converged rows within 1e-5, non-converged rows never worse than the
    oracle's by more than a stated factor; validity flags at the Constraints thresholds agree with the reference-order fp32
    formula outside that formula's own quantisation band.
"""

import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

ROBOTS = ["panda", "fetch", "fetch_arm", "chain12"]
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"
QS = (0.5, 0.9, 0.99, 1.0)


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def robots():
    from cppflow_amd.robots import get_robot

    return {n: get_robot(n) for n in ROBOTS}


def _task_space(Js, dx):
    return np.abs(np.einsum("nij,nj->ni", Js, dx)).max(axis=1)


def _one_step_case(name, source):
    if source == "golden":
        z = np.load(os.path.join(GOLDEN, f"lm_golden_{name}.npz"))
        x0, target = H.f32(z["x0"]), H.f32(z["target"])
        return x0, target, int(z["S"])
    if source == "special":
        # the corners: starts at / next to a kinematic singularity (the zero pose and the stretched arm), on the joint limits,
        # exactly on the solution (zero residual), and far from it (a target taken from an unrelated configuration)
        ch = H.chain(name)
        rng = np.random.RandomState(11)
        W, d = 64, ch.ndof
        q_star = rng.uniform(ch.lo, ch.hi, size=(W, d))
        target = H.f32(H.oracle64(name).fk(H.f32(q_star)))
        mid = np.clip(np.zeros(d), ch.lo, ch.hi)
        blocks = [
            np.tile(mid, (W, 1)) + 1e-3 * rng.randn(W, d),                       # the zero pose (clamped into the limits)
            np.tile(mid, (W, 1)) + 1e-6 * rng.randn(W, d),                       # ... to within a micro-radian
            np.tile(ch.lo, (W, 1)) + 1e-4 * np.abs(rng.randn(W, d)),             # on the lower limits
            np.tile(ch.hi, (W, 1)) - 1e-4 * np.abs(rng.randn(W, d)),             # on the upper limits
            q_star.copy(),                                                        # exactly on the solution
            q_star + 1e-6 * rng.randn(W, d),                                      # a micro-radian off it
            rng.uniform(ch.lo, ch.hi, size=(W, d)),                              # far: unrelated configurations
            q_star + 0.1 * rng.randn(W, d),                                       # the usual seeds
        ]
        x0 = H.f32(np.clip(np.concatenate(blocks), ch.lo, ch.hi))
        return x0, target, len(blocks)
    S, W = 64, 64
    x0, target = H.lm_problem(name, S, W, seed=3)
    return x0, target, S


@pytest.mark.parametrize("source", ["seeded", "golden", "special"])
@pytest.mark.parametrize("name", ROBOTS)
def test_one_step_task_space_parity_on_all_rows(robots, name, source):
    from cppflow_amd import _hip

    x0, target, S = _one_step_case(name, source)
    tgt = H.stacked(target, S)
    o64, o32 = H.oracle64(name), H.oracle32(name)
    x64, Js, es, fails = o64.lm_step(x0, tgt, solver=0, **LM)  # reference order: primal normal equations, LU, fp64
    assert fails == 0
    x32, _, _, _ = o32.lm_step(x0, tgt, solver=0, **LM)  # the same in the reference's own dtype
    sv = np.linalg.svd(Js, compute_uv=False)
    smin, smax = sv[:, -1], sv[:, 0]
    well = smin >= 2e-2
    pe64, re64 = o64.pose_metrics_exact(x64, tgt)
    ts32 = _task_space(Js, x32 - x64)
    xs = {}
    for solver in (_hip.SOLVER_F64, _hip.SOLVER_AUTO, _hip.SOLVER_F32):
        for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
            r = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, solver=solver, shape=shape, **LM)
            xs[(solver, shape)] = host(r["x"])
            assert np.isfinite(xs[(solver, shape)]).all()
    # ---- fp64 solve: every row, both shapes ----
    step = np.abs(x64 - x0).max(axis=1)
    calm = step < 1.0
    assert calm.mean() > (0.3 if source == "special" else 0.9)
    small = step < 0.1
    # At a singular start with a far target (source "special": the zero pose of Fetch / the 12-joint chain, sigma_min 1e-8 .. 1e-4,
    # |e_s| ~ 5) the step's filter factors sigma^2 / (sigma^2 + lambda) move with the LAST BITS of the fp32 Jacobian itself: a
    # relative change 1e-7 sigma_max / sigma of a singular value near sqrt(lambda) = 1e-3 is 1e-4 .. 1e-3 of |e_s| in task space,
    # whatever solves the system (measured 5e-4 .. 1.3e-3 there with the fp64 solve; the reference-order fp32 arithmetic is off by
    # 0.5 .. 12 on the same rows).  The bar for those rows is that perturbation bound; everywhere else it is 2e-5 as before.
    e_norm = np.abs(es).reshape(len(x0), -1).max(axis=1)
    bar64 = np.full(len(x0), 2e-5)
    if source == "special":
        bar64 = 2e-5 + 2.5e-7 * smax * e_norm / np.maximum(smin, 1e-3)
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        x = xs[(_hip.SOLVER_F64, shape)]
        ts = _task_space(Js, x - x64)
        assert (ts <= bar64).all(), (name, source, shape, ts.max(), np.max(ts / bar64))
        pe, re = o64.pose_metrics_exact(x, tgt)
        # pose error after ONE step: the step itself is nonlinear in x, so a joint-space difference d in a near-null direction of
        # J(x0) moves the pose at x0 + delta by |J(x0 + delta) - J(x0)| d ~ |delta| d: the bar is 1e-5 + 2 |delta| |d|; rows whose
        # oracle step is a jump of radians (near-singular linearisation) are compared in task space above
        dxj = np.abs(x - x64).max(axis=1)
        tol = 1e-5 + 2.0 * step * dxj
        assert (np.abs(pe - pe64) <= tol)[calm].all(), (name, source, np.max((np.abs(pe - pe64) / tol)[calm]))
        assert (np.abs(re - re64) <= tol)[calm].all(), (name, source, np.max((np.abs(re - re64) / tol)[calm]))
        assert np.abs(pe - pe64)[small].max() <= 1.2e-5 and np.abs(re - re64)[small].max() <= 1.2e-5
        # joint space on the well-conditioned rows: the reference's own bar between its two formulations (tests/optimization_test.py:99)
        assert np.abs(x - x64)[well].max() < 5e-3
    # ---- the default solver (fp32, conditioning-gated), both shapes: never worse than the reference's own fp32 arithmetic ----
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        x = xs[(_hip.SOLVER_AUTO, shape)]
        ts = _task_space(Js, x - x64)
        assert ts.max() <= max(ts32.max(), 1e-4), (name, source, shape, ts.max(), ts32.max())
        if source == "seeded":  # 4096 rows: a 99th percentile means something (the golden case has 64)
            assert np.quantile(ts, 0.99) <= 2.0 * np.quantile(ts32, 0.99), (name, shape, np.quantile(ts, QS), np.quantile(ts32, QS))
            assert np.quantile(ts, 0.5) <= 2.0 * max(np.quantile(ts32, 0.5), 2e-7)
            assert np.quantile(ts, 0.9) <= 2.0 * max(np.quantile(ts32, 0.9), 5e-7)
        assert ts[well].max() <= 1e-4, (name, source, shape, ts[well].max())
        if source == "special":  # row by row: the gated solve meets the fp64 solve's bar (x5: the gate's tolerance is 1e-5)
            assert (ts <= 5.0 * bar64).all(), (name, shape, np.max(ts / bar64))
        pe, re = o64.pose_metrics_exact(x, tgt)
        ok = well & calm
        assert np.abs(pe - pe64)[ok].max() <= 2e-4 and np.abs(re - re64)[ok].max() <= 5e-4
        dx, dx32 = np.abs(x - x64).max(axis=1), np.abs(x32 - x64).max(axis=1)
        assert np.median(dx[well]) <= np.median(dx32[well]) + 1e-6  # no null-space noise: closer to fp64 than the reference's fp32 is
    # ---- fp32 without the gate: the well-conditioned rows ----
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        ts = _task_space(Js, xs[(_hip.SOLVER_F32, shape)] - x64)
        if source == "special":  # (the ungated fp32 error is eps cond^2 |e_s|: far targets have |e_s| ~ 5 -- why the gate exists)
            assert (ts[well] <= 1e-4 + 1e-6 * (smax[well] / smin[well]) ** 2 * e_norm[well]).all(), (name, source, shape)
        else:
            assert ts[well].max() <= 1e-4 and np.median(ts) <= 1e-6, (name, source, shape)


@pytest.mark.parametrize("name", ROBOTS)
def test_k_step_pose_error_on_all_rows_including_unconverged(robots, name):
    """K = 10 fused steps.  Rows on which BOTH the oracle and the build converge: pose errors within 1e-5.  The iteration is
    chaotic where it passes a singularity, so a handful of rows converge in one arithmetic and not in the other: their number
    is bounded (<= 0.5 % of the rows) and symmetric (the build is not the one that loses more often).  On the rows the oracle
    does not converge on, the build's pose error is not worse than the oracle's by more than 10x + 1e-4 m / 1e-3 rad on 90 %
    of them, and it is not left far (> 5 cm) from the target more often than the oracle is."""
    from cppflow_amd import _hip

    S, W, K = 32, 64, 10
    x0, target = H.lm_problem(name, S, W, seed=4)
    tgt = H.stacked(target, S)
    o64 = H.oracle64(name)
    x_orc = o64.lm_steps(x0, tgt, K, solver=0, **LM)
    pe_o, re_o = o64.pose_metrics_exact(x_orc, tgt)
    conv = (pe_o < 1e-4) & (re_o < 1.2e-3)
    assert 0.9 < conv.mean() < 1.0 or name in ("chain12",)  # the case does contain rows that do not converge in K steps
    for shape, solver in ((_hip.SHAPE_ROW, _hip.SOLVER_AUTO), (_hip.SHAPE_QUAD, _hip.SOLVER_AUTO), (_hip.SHAPE_ROW, _hip.SOLVER_F32), (_hip.SHAPE_ROW, _hip.SOLVER_F64), (_hip.SHAPE_QUAD, _hip.SOLVER_F64)):
        r = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, shape=shape, solver=solver, **LM)
        pe, re = host(r["pos_err_m"]), host(r["rot_err_rad"])
        pe_at, re_at = o64.pose_metrics_exact(host(r["x"]), tgt)
        assert np.abs(pe - pe_at).max() < 1e-5 and np.abs(re - re_at).max() < 1e-5  # reported errors are those of its own x
        conv_g = (pe < 1e-4) & (re < 1.2e-3)
        both = conv & conv_g
        settled = both & (pe_o < 5e-6)  # at the fp32 floor; rows between 5e-6 and 1e-4 are still contracting (factor below)
        assert settled.sum() > 0.7 * conv.sum()
        assert np.abs(pe - pe_o)[settled].max() < 1e-5 and np.abs(re - re_o)[settled].max() < 1e-5
        assert (pe[both] <= 3.0 * pe_o[both] + 1e-5).all()
        only_oracle, only_build = int((conv & ~conv_g).sum()), int((conv_g & ~conv).sum())
        assert only_oracle <= max(2, 0.005 * conv.size), (name, shape, solver, only_oracle, only_build)
        assert only_oracle <= only_build + max(2, 0.003 * conv.size), (name, shape, solver, only_oracle, only_build)
        rest = ~conv
        if rest.sum() >= 8:
            worse_p = pe[rest] > 10.0 * pe_o[rest] + 1e-4
            worse_r = re[rest] > 10.0 * re_o[rest] + 1e-3
            assert worse_p.mean() <= 0.1 and worse_r.mean() <= 0.1, (name, shape, worse_p.mean(), worse_r.mean())
            # the unconverged rows are bimodal (a few mm from the pose, or half a metre away on another branch): "not worse in
            # the median" is a coin toss on two dozen chaotic rows; what can be held is that the build is not left far from the
            # target more OFTEN than the oracle
            far, far_o = (pe[rest] > 0.05).mean(), (pe_o[rest] > 0.05).mean()
            assert far <= far_o + 0.2, (name, shape, solver, far, far_o)


@pytest.mark.parametrize("name", ROBOTS)
def test_validity_flags_agree_with_reference_order_fp32_formula(robots, name):
    """x_is_valid thresholds (Constraints: 0.01 cm, 0.1 deg -- scripts/evaluate.py:51-56) on the kernel's rotation error
    (atan2 form) against the reference-order fp32 evaluation 2*acos(clamp(q_t . q_c)) (oracle32.pose_metrics): the flags agree
    on every row outside the acos formula's quantisation band around the threshold; rows inside the band are counted."""
    S, W, K = 32, 64, 6  # K = 6: a spread of rotation errors around 0.1 deg
    x0, target = H.lm_problem(name, S, W, seed=9)
    tgt = H.stacked(target, S)
    r = robots[name].lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, **LM)
    x = host(r["x"])
    pe, re = host(r["pos_err_m"]), host(r["rot_err_rad"])
    pe32, re32 = H.oracle32(name).pose_metrics(x, tgt)  # reference order, reference dtype
    thr_p, thr_r = 1e-4, np.deg2rad(0.1)
    flag_p, flag_p32 = pe < thr_p, pe32 < thr_p
    flag_r, flag_r32 = re < thr_r, re32 < thr_r
    # position: plain norm, fp32 rounding only
    band_p = np.abs(pe32 - thr_p) < 2e-7
    assert np.array_equal(flag_p[~band_p], flag_p32[~band_p])
    # rotation: 2*acos(dot) in fp32 moves in steps of ~2*sqrt(2*6e-8 / (1 - dot)) ~ 4e-4 rad at 0.1 deg and is floored at
    # 8.944e-4 by the clamp; outside +-4.5e-4 rad of the threshold the two evaluations must agree
    band_r = np.abs(re32 - thr_r) < 4.5e-4
    assert np.array_equal(flag_r[~band_r], flag_r32[~band_r]), int((flag_r != flag_r32)[~band_r].sum())
    n_band = int(band_r.sum())
    n_disagree = int((flag_r != flag_r32).sum())
    assert n_disagree <= n_band
    # the fp64 evaluation sides with the kernel inside the band (the kernel's formula is the accurate one)
    _, re64 = H.oracle64(name).pose_metrics_exact(x, tgt)
    assert np.array_equal(flag_r, re64 < thr_r) or np.abs(re64 - thr_r)[flag_r != (re64 < thr_r)].max() < 1e-5
    print(f"{name}: {n_band} of {S * W} rows inside the acos quantisation band, {n_disagree} flag disagreements, all inside it")


CONV_FLOOR = {"C2": 0.95, "C3": 0.9, "C4": 0.95}  # measured 0.980 / 0.99 (0.553 in round 3, when a Fetch seed drew a lift height per waypoint: DESIGN.md section 2) / 0.980

CONFIGS = {
    "C2": ("panda", "panda__1cube_first64", 128, []),
    "C3": ("fetch", "fetch__hello_first256", 512, []),
}


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_full_size_configs_sampled_against_the_oracle(robots, cfg):
    """BASELINE.json configs 2-4 at FULL size on the reference's target paths; 4096 sampled rows are re-done by the oracle:
    K-step LM result (pose error of converged rows within 1e-5), one-step task-space parity (fp64 solve <= 2e-5 on every sampled
    row), metrics at the kernel's x (1e-5) and masks / cost at the kernel's x (bit-exact vs the fp32 oracle)."""
    from cppflow_amd import _hip
    from cppflow_amd.data_type_utils import problem_from_arrays
    from cppflow_amd.optimization import run_lm_pose_refinement
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES

    cfgs = dict(CONFIGS, C4=("panda", "panda__2cubes_resampled256", 1024, PANDA_2CUBES_OBSTACLES))
    name, key, S, obs = cfgs[cfg]
    rb = robots[name]
    target = np.load(os.path.join(GOLDEN, "reference_paths.npz"))[key]
    W, K = target.shape[0], 10
    problem = problem_from_arrays(rb, target, obs, device=DEV)
    # the seeds bench.py itself measures on: per seed an IK branch tracking the path, + 0.1 randn, clamped
    import bench

    g = torch.Generator().manual_seed(1)
    x0, target_b, _ = bench.make_inputs_problem(rb, S, W, torch.device(DEV), seed=1)
    assert torch.equal(target_b.cpu(), torch.tensor(target, dtype=torch.float32))
    r = run_lm_pose_refinement(problem, x0, n_steps=K)  # binds the obstacles and the default joint-limit padding
    rows = torch.randperm(S * W, generator=g)[:4096]
    rows_d = rows.to(DEV)
    x0_s = host(x0[rows_d])
    tgt_s = H.f32(target)[(rows % W).numpy()]
    o64, o32 = H.oracle64(name), H.oracle32(name)
    # K fused steps
    x_o = o64.lm_steps(x0_s, tgt_s, K, solver=0, **LM)
    pe_o, re_o = o64.pose_metrics_exact(x_o, tgt_s)
    conv = (pe_o < 1e-4) & (re_o < 1.2e-3)
    print(f"{cfg}: oracle converges on {conv.mean():.3f} of the sampled rows")
    assert conv.mean() > CONV_FLOOR[cfg], conv.mean()
    pe_g, re_g = host(r.pos_err_m.view(-1)[rows_d]), host(r.rot_err_rad.view(-1)[rows_d])
    conv_g = (pe_g < 1e-4) & (re_g < 1.2e-3)
    both = conv & conv_g
    settled = both & (pe_o < 5e-6)  # at the fp32 floor after K steps (the others are still contracting)
    assert settled.sum() > 0.7 * conv.sum()
    assert np.abs(pe_g - pe_o)[settled].max() < 1e-5 and np.abs(re_g - re_o)[settled].max() < 1e-5
    assert (pe_g[both] <= 3.0 * pe_o[both] + 1e-5).all()
    assert (conv & ~conv_g).sum() <= 0.005 * conv.size and abs(conv_g.mean() - conv.mean()) < 0.01
    # metrics and masks at the kernel's own x
    x_g = host(r.x[rows_d])
    pe_at, re_at = o64.pose_metrics_exact(x_g, tgt_s)
    assert np.abs(pe_g - pe_at).max() < 1e-5 and np.abs(re_g - re_at).max() < 1e-5
    lo_b, hi_b = H.box_corners([c.numpy() for c in problem.obstacles_cuboids], [T.numpy() for T in problem.obstacles_Tcuboids])
    jl_lo, jl_hi = rb.padded_joint_limits()
    m = o32.masks(x_g, lo_b if len(obs) else None, hi_b if len(obs) else None, jl_lo, jl_hi)
    assert np.array_equal(r.self_mask.view(-1)[rows_d].cpu().numpy().astype(np.uint8), m["self_mask"])
    assert np.array_equal(r.env_mask.view(-1)[rows_d].cpu().numpy().astype(np.uint8), m["env_mask"])
    assert np.array_equal(r.jlim_mask.view(-1)[rows_d].cpu().numpy().astype(np.uint8), m["jlim_mask"])
    assert np.array_equal(host(r.ext_cost.view(-1)[rows_d]), m["ext_cost"])
    # one step on the sampled rows, task space: fp64 solve <= 2e-5 on every row; the default (conditioning-gated) solve never
    # worse than the reference-order fp32 arithmetic on the same rows
    x64, Js, es, _ = o64.lm_step(x0_s, tgt_s, solver=0, **LM)
    x32_ref, _, _, _ = o32.lm_step(x0_s, tgt_s, solver=0, **LM)
    full64 = rb.lm_pose_steps(x0, problem.target_path, n_steps=1, clamp=False, solver=_hip.SOLVER_F64, **LM)["x"]
    full_auto = rb.lm_pose_steps(x0, problem.target_path, n_steps=1, clamp=False, **LM)["x"]
    ts64 = _task_space(Js, host(full64[rows_d]) - x64)
    ts_auto = _task_space(Js, host(full_auto[rows_d]) - x64)
    ts_ref = _task_space(Js, x32_ref - x64)
    assert ts64.max() <= 2e-5, ts64.max()
    assert ts_auto.max() <= max(ts_ref.max(), 1e-4), (ts_auto.max(), ts_ref.max())
    # (on these well-conditioned planner inputs both sit at the fp32 floor of FK + Jacobian, ~1e-6 in scaled task space: the
    # quantile rule gets a floor of 5e-6 here; the all-rows test above holds it without one on the ill-conditioned random seeds)
    assert np.quantile(ts_auto, 0.99) <= max(2.0 * np.quantile(ts_ref, 0.99), 5e-6), (np.quantile(ts_auto, QS), np.quantile(ts_ref, QS))
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)
