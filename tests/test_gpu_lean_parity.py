"""The K-step iterate of a fused launch against the oracle's K-step iterate OFF the fixed point (pytest -m gpu).

Iterations 0 .. K-2 of a plain K-step launch are "lean" (csrc/kernels_fused.h): in the row shape polynomial sine / cosine (5e-7
absolute), the residual's angle functions behind one shared reciprocal (4e-7) and the 6x6 solve as a 2x2-block L D L^T (three
reciprocals for six reciprocal square roots), and in both shapes a flagged row is re-solved in double precision only when the
estimated step error also exceeds 1e-3 of the residual.  Every other K > 1 test compares CONVERGED rows after K = 10, where any contraction lands on the same point; here
K = 2, 3, 5 -- iterates that are still moving -- are held against `oracle64.lm_steps(x0, K)` (the reference's step,
cppflow/optimization.py:61-92 + the clamp of :259, reference operation order, fp64):

  * scaled task-space difference of the iterate,  ts = |J_s(x_K^oracle) (x_K - x_K^oracle)|_inf,  on the calm rows (every oracle
    step of the K below 1 rad): CPPF_SOLVER_F64 median <= 1e-6 and 99th percentile <= K x 2e-5 (the K = 1 bar of
    test_gpu_parity_allrows.py per step; the maximum is a few near-singular rows whose amplification the row-wise rule prices); the default
    CPPF_SOLVER_AUTO never worse than the reference-order fp32 oracle's own distance from the fp64 one, quantile by quantile (50 / 90 /
    99 %) and at the maximum; and ROW BY ROW  ts <= K x eps x (1 + 2 |step| / sigma_min)  with eps = 2e-5 (F64) / 1e-4 (AUTO, the K = 1
    floor) and sigma_min the smallest singular value of J_s over the row's K linearisation points -- K per-step errors of the K = 1
    size, grown by no more than the row's own amplification (a step's error has a weak-direction part eps / sigma in joint space, which
    the next linearisation turns back into task space through the chain's curvature);
  * pose error after K steps within 1e-5 + 2 K |step| |dx| of the oracle's (the K = 1 rule, the second-order term once per step, with
    the largest of the K steps);
  * the batch entry point is the row shape bit for bit; K launches of ONE canonical step and one launch of K steps (K - 1 of them
    lean) differ by no more than the same bars;
  * the rows on which the relative gate declines a re-solve (modelled on the oracle's iterates with the kernel's own estimate) are
    counted -- a bounded share -- and what declining changes is bounded: gate_rel_ppm = 0 (absolute gate in every iteration) against
    the default moves the iterate by <= K x 1e-3 of the scaled residual it started from, and not at all on the rows never flagged.

The measured distributions behind the bars: scripts/lean_parity_stats.py -> profiles/r5_lean_parity.txt."""

import os

import numpy as np
import pytest
import torch

from scripts.lean_parity_stats import gate_model, oracle_trace, task_space
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROBOTS = ["panda", "fetch", "fetch_arm", "chain12"]
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BAR64_PER_STEP = 2e-5  # the K = 1 bar of the fp64 solve (test_gpu_parity_allrows.py), per step
FLOOR32_PER_STEP = 1e-4  # the K = 1 floor of "never worse than the reference-order fp32"


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(scope="module")
def robots():
    from cppflow_amd.robots import get_robot

    return {n: get_robot(n) for n in ROBOTS}


def _case(robots, name, source):
    if source == "seeded":
        x0, target = H.lm_problem(name, 32, 64, seed=21)
        return x0, H.stacked(target, 32)
    # C4's own inputs: the reference problem's path + per-seed IK branches + 0.1 randn (bench.py), 2048 sampled rows
    from cppflow_amd.problems_synthetic import make_inputs_problem

    target = np.load(os.path.join(GOLDEN, "reference_paths.npz"))["panda__2cubes_resampled256"]
    x0, _, _ = make_inputs_problem(robots["panda"], 1024, 256, torch.device(DEV), seed=1)
    rows = torch.randperm(1024 * 256, generator=torch.Generator().manual_seed(5))[:2048]
    return host(x0[rows.to(DEV)]), H.f32(target)[(rows % 256).numpy()]


CASES = [(n, "seeded") for n in ROBOTS] + [("panda", "C4-sampled")]


def _amplification(Js, steps):
    """1 + 2 |largest step| / (smallest singular value of the scaled Jacobian over the K linearisation points): how far a per-step
    task-space error of size eps can have grown by the end.  A step's error has a null-space / weak-direction part of size eps / sigma in
    JOINT space, which the next linearisation turns back into task space through the chain's curvature, |dJ| |dx| ~ |step| eps / sigma
    (the same first-order argument as the "1e-5 + 2 |step| |dx|" pose-error rule of test_gpu_parity_allrows.py)."""
    smin = np.min([np.linalg.svd(J, compute_uv=False)[:, -1] for J in Js], axis=0)
    return 1.0 + 2.0 * steps / np.maximum(smin, 1e-6), smin


@pytest.mark.parametrize("K", [2, 3, 5])
@pytest.mark.parametrize("name,source", CASES)
def test_k_step_iterate_against_the_oracle_off_the_fixed_point(robots, name, source, K):
    from cppflow_amd import _hip

    bad = []  # every violated bar (all of them are evaluated, then reported together)

    def check(ok, *what):
        if not ok:
            bad.append(what)

    rb = robots[name]
    x0, tgt = _case(robots, name, source)
    n = len(x0)
    o64, o32 = H.oracle64(name), H.oracle32(name)
    xs, Js, es = oracle_trace(o64, x0, tgt, K)
    xo = xs[-1]
    _, JK, _, _ = o64.lm_step(xo, tgt, solver=0, **LM)
    x32 = o32.lm_steps(x0, tgt, K, solver=0, **LM)
    steps = np.max([np.abs(xs[k + 1] - xs[k]).max(axis=1) for k in range(K)], axis=0)
    calm = steps < 1.0
    assert calm.mean() > (0.85 if name != "chain12" else 0.5), calm.mean()
    amp, smin = _amplification(Js + [JK], steps)
    pe_o, re_o = o64.pose_metrics_exact(xo, tgt)
    ts32 = task_space(JK, x32 - xo)
    x0_d, tgt_d = dev(x0), dev(tgt)  # every row with its own target: W = n, the stacked form
    got = {}
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        for solver in (_hip.SOLVER_F64, _hip.SOLVER_AUTO):
            r = rb.lm_pose_steps(x0_d, tgt_d, n_steps=K, want_errors=True, shape=shape, solver=solver, **LM)
            x = got[(shape, solver)] = host(r["x"])
            assert np.isfinite(x).all()
            ts = task_space(JK, x - xo)
            tag = (name, source, K, "row" if shape == _hip.SHAPE_ROW else "quad", "f64" if solver == _hip.SOLVER_F64 else "auto")
            q50, q90, q99 = np.quantile(ts[calm], (0.5, 0.9, 0.99))
            r50, r90, r99 = np.quantile(ts32[calm], (0.5, 0.9, 0.99))
            # never worse than the reference's own fp32 arithmetic, quantile by quantile and at the maximum (floors: the fp32 floor of FK +
            # Jacobian, ~1e-6 in scaled task space per step)
            check(q50 <= max(r50, K * 1e-6) and q90 <= max(r90, K * 5e-6) and q99 <= max(r99, K * BAR64_PER_STEP) and ts[calm].max() <= max(ts32[calm].max(), K * FLOOR32_PER_STEP),
                  tag, "vs reference-order fp32", (q50, q90, q99, ts[calm].max()), (r50, r90, r99, ts32[calm].max()))
            check(q50 <= 1e-6, tag, "median", q50)
            eps = BAR64_PER_STEP if solver == _hip.SOLVER_F64 else FLOOR32_PER_STEP
            if solver == _hip.SOLVER_F64:
                check(q99 <= K * BAR64_PER_STEP, tag, "q99", q99)
            else:
                check(q90 <= K * 5e-6, tag, "q90", q90)
            # row by row: K per-step errors of the K = 1 size, grown by at most the row's own amplification
            ratio = ts / (K * eps * amp)
            check(ratio[calm].max() <= 1.0, tag, "row-wise K eps amp", ratio[calm].max(), int(np.argmax(np.where(calm, ratio, 0))))
            # pose error after K steps (the K = 1 rule with the largest of the K steps)
            pe, re = o64.pose_metrics_exact(x, tgt)
            dxj = np.abs(x - xo).max(axis=1)
            # (K steps: the second-order term of each step adds up)
            tol = (1e-5 if solver == _hip.SOLVER_F64 else 2e-4) + 2.0 * K * steps * dxj
            check((np.abs(pe - pe_o) <= tol)[calm].all(), tag, "pos err", np.max((np.abs(pe - pe_o) / tol)[calm]))
            check((np.abs(re - re_o) <= tol + (0 if solver == _hip.SOLVER_F64 else 3e-4))[calm].all(), tag, "rot err", np.max((np.abs(re - re_o) / tol)[calm]))
            # ... and the launch's own pose-error outputs are those of its own x_K
            check(np.abs(host(r["pos_err_m"]) - pe).max() < 1e-5 and np.abs(host(r["rot_err_rad"]) - re).max() < 1e-5, tag, "own errors")
    row_auto = got[(_hip.SHAPE_ROW, _hip.SOLVER_AUTO)]
    # the batch entry point: the row shape, bit for bit
    xb = torch.empty_like(x0_d)
    rb.lm_batch_plan([dict(x=x0_d, target=tgt_d, x_out=xb)], n_steps=K, **LM).launch()
    torch.cuda.synchronize()
    check(np.array_equal(host(xb), row_auto), "batch entry != row shape")
    # K launches of one canonical step (every iteration canonical, absolute gate) against one launch of K steps (K - 1 lean ones)
    xc = x0_d
    for _ in range(K):
        xc = rb.lm_pose_steps(xc, tgt_d, n_steps=1, shape=_hip.SHAPE_ROW, **LM)["x"]
    tsc = task_space(JK, host(xc) - row_auto)
    # (two iterates of the default solver against each other: each may sit K eps amp from the exact iterate, so the bar is twice that)
    check((tsc / (2 * K * FLOOR32_PER_STEP * amp))[calm].max() <= 1.0, "K canonical launches vs one launch of K: row-wise", (tsc / (2 * K * FLOOR32_PER_STEP * amp))[calm].max())
    check(np.quantile(tsc[calm], 0.5) <= K * 2e-6, "K canonical launches vs one launch of K: median", np.quantile(tsc[calm], 0.5))  # the polynomials' 5e-7, through the chain
    # ---- the relative gate: which rows it declines, and what that changes ----
    flagged, declined = np.zeros(n, bool), np.zeros(n, bool)
    for k in range(K - 1):  # the lean iterations
        f, d = gate_model(Js[k], es[k])
        flagged |= f
        declined |= d
    # a bounded share: a quarter of a per cent on the planner's inputs, up to 15 % on independent random configurations (the worst case)
    check(declined.mean() <= (0.01 if source == "C4-sampled" else 0.2), "declined share", declined.mean())
    rb.debug_set("gate_rel_ppm", 0)
    try:
        xa = host(rb.lm_pose_steps(x0_d, tgt_d, n_steps=K, shape=_hip.SHAPE_ROW, **LM)["x"])
    finally:
        rb.debug_set("gate_rel_ppm", None)
    moved = np.abs(xa - row_auto).max(axis=1) > 0
    e0 = np.linalg.norm(es[0], axis=1)  # the scaled residual the launch started from
    tsa = task_space(JK, xa - row_auto)
    rel = tsa / ((1e-3 * e0 + FLOOR32_PER_STEP) * K * amp)
    check(rel[calm].max() <= 1.0, "relative gate on vs off: row-wise K (1e-3 |e0| + 1e-4) amp", rel[calm].max())
    # a row the model never flags is solved in fp32 either way: switching the relative gate off must not move it (the model's estimate
    # and the kernel's differ in the last bits, so "never flagged" carries a factor-2 margin on the estimate)
    f2 = np.zeros(n, bool)
    for k in range(K - 1):
        f2 |= gate_model(Js[k], es[k], tau=0.5e-5)[0]
    check(moved[~f2 & calm].mean() <= 0.002, "rows never flagged moved by the relative gate", int(moved[~f2 & calm].sum()))
    print(f"{name} {source} K={K}: calm {calm.mean():.3f}, flagged {flagged.mean():.4f}, declined {declined.mean():.4f}, moved by the relative gate {moved.mean():.4f}, "
          f"smin q01 {np.quantile(smin, 0.01):.2e}, amp q99 {np.quantile(amp[calm], 0.99):.1f}")
    assert not bad, "\n".join(str(b) for b in bad)


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_lean_angle_functions_general_path_and_wave_independence(robots, name):
    """The lean iterations evaluate roll / pitch / yaw on a principal-range FAST path when every row of the wavefront is within
    |pitch| <= 30, |roll|, |yaw| <= 45 degrees, and in the general lean form otherwise (csrc/kernels_chain.h: pose_error<true>).  Rows with
    rotation errors of a radian (a start 0.45 rad per joint from the solution) force the general form; interleaved with near rows
    they also put near rows into wavefronts that take it.  (1) a row's K-step result does not depend on which rows share its wavefront
    -- the same rows in another order give the same bits; (2) near AND far rows stay inside the row-wise bar of the test above against
    the fp64 oracle."""
    from cppflow_amd import _hip

    rb = robots[name]
    ch = H.chain(name)
    K, n = 3, 2048
    rng = np.random.RandomState(9)
    q_star = H.f32(rng.uniform(ch.lo, ch.hi, size=(n, ch.ndof)))
    tgt = H.f32(H.oracle64(name).fk(q_star))
    x0 = np.clip(q_star + 0.05 * rng.randn(n, ch.ndof), ch.lo, ch.hi)
    far = rng.rand(n) < 0.25  # a quarter of the rows start 0.45 rad (per joint, 1 sigma) from their solution: rotation errors of a radian
    x0[far] = np.clip(q_star[far] + 0.45 * rng.randn(int(far.sum()), ch.ndof), ch.lo, ch.hi)
    x0 = H.f32(x0)
    o64 = H.oracle64(name)
    e0 = o64.lm_step(x0, tgt, solver=0, **LM)[2].reshape(n, 6) / 0.35  # (the scaled residual's rotation rows -> radians)
    general = (np.abs(e0[:, [0, 2]]).max(axis=1) > np.pi / 4) | (np.abs(e0[:, 1]) > np.pi / 6)  # rows that leave the principal range
    assert general[far].mean() > 0.4 and not general[~far].any()
    x = host(rb.lm_pose_steps(dev(x0), dev(tgt), n_steps=K, shape=_hip.SHAPE_ROW, **LM)["x"])
    # (1) another order of the same rows: near rows first, far rows last -> wavefronts of near rows only take the fast path
    order = np.argsort(far, kind="stable")
    xp = host(rb.lm_pose_steps(dev(x0[order]), dev(tgt[order]), n_steps=K, shape=_hip.SHAPE_ROW, **LM)["x"])
    assert np.array_equal(xp, x[order])
    # (2) against the oracle
    xs, Js, _ = oracle_trace(o64, x0, tgt, K)
    _, JK, _, _ = o64.lm_step(xs[-1], tgt, solver=0, **LM)
    steps = np.max([np.abs(xs[k + 1] - xs[k]).max(axis=1) for k in range(K)], axis=0)
    amp, _ = _amplification(Js + [JK], steps)
    calm = steps < 1.0
    ts = task_space(JK, x - xs[-1])
    assert (ts / (K * FLOOR32_PER_STEP * amp))[calm].max() <= 1.0, (name, (ts / (K * FLOOR32_PER_STEP * amp))[calm].max())
    # ... and that bar was held on rows that DID go through the general form (not only on the near rows of mixed wavefronts)
    assert np.quantile(ts[calm & ~far], 0.5) <= 1e-6 and calm[~far].mean() > 0.95 and (calm & far & general).sum() > 50
