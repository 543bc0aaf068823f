"""Developer fuzz of the fused launch: robots x (S, W) shapes x K x kernel shape x solver x outputs.  For every combination:
(1) x after K steps against the fp64 oracle on the rows whose Jacobian is well conditioned (5e-3, the reference's own tolerance);
(2) pose errors, masks and cost at the launch's OWN x against the oracle (1e-5 / bit-exact); (3) the per-seed summary against the
separate reduction kernel over the same per-row outputs (bit-exact); (4) the two kernel shapes against each other.
Prints every disagreement; exits 1 if any."""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

DEV = "cuda:0"
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


bad = checked = 0


def fail(*a):
    global bad
    bad += 1
    print("DIFF", *a)


for name in ("panda", "fetch", "fetch_arm", "chain12"):
    rb, o64, o32, ch = get_robot(name), H.oracle64(name), H.oracle32(name), H.chain(name)
    d = rb.ndof
    obs = H.PANDA_2CUBES
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    jl_lo, jl_hi = rb.padded_joint_limits()
    for (S, W) in ((1, 1), (1, 63), (3, 64), (2, 65), (5, 128), (3, 200), (2, 256), (1, 300), (7, 37), (130, 64), (40, 256)):
        x0, target = H.lm_problem(name, S, W, seed=S * 1000 + W)
        n = S * W
        for K in (1, 3, 10):
            ref = None
            for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD, _hip.SHAPE_AUTO):
                for solver in (_hip.SOLVER_AUTO, _hip.SOLVER_F32):
                    pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
                    sm = torch.empty((S, 8), dtype=torch.float32, device=DEV)
                    try:
                        r = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, packed_out=pk, summary_out=sm, shape=shape, solver=solver, **LM)
                        torch.cuda.synchronize()
                    except AssertionError as e:
                        if shape == _hip.SHAPE_QUAD:  # the quad shape refuses what it cannot produce (documented)
                            continue
                        fail(name, S, W, K, shape, solver, "EXC", repr(e)[:120]); continue
                    checked += 1
                    tag = (name, "S", S, "W", W, "K", K, "shape", shape, "solver", solver)
                    x = host(r["x"])
                    if not np.isfinite(x).all():
                        fail(*tag, "non-finite x"); continue
                    # (2) per-row outputs at the launch's own x
                    pe, re = o64.pose_metrics_exact(x, H.stacked(target, S))
                    if np.abs(host(r["pos_err_m"]) - pe).max() > 1e-5 or np.abs(host(r["rot_err_rad"]) - np.maximum(re, 8.94427191e-4)).max() > 1e-5:
                        fail(*tag, "pose errors", np.abs(host(r["pos_err_m"]) - pe).max(), np.abs(host(r["rot_err_rad"]) - np.maximum(re, 8.94427191e-4)).max())
                    m = o32.masks(x, lo, hi, jl_lo, jl_hi)
                    for k in ("self_mask", "env_mask", "jlim_mask"):
                        if not np.array_equal(r[k].cpu().numpy().astype(np.uint8), m[k]):
                            fail(*tag, k, int((r[k].cpu().numpy() != m[k]).sum()), "rows differ")
                    if not np.array_equal(host(r["ext_cost"]), m["ext_cost"]):
                        fail(*tag, "ext_cost")
                    # (3) the in-launch / second-kernel summary against the separate reduction
                    sm2 = rb.seed_summary(r["x"], pk, S, W)
                    if not torch.equal(sm, sm2):
                        fail(*tag, "summary", float((sm - sm2).abs().max()))
                    # (1) against the oracle's K steps, well-conditioned rows
                    if solver == _hip.SOLVER_AUTO and shape == _hip.SHAPE_ROW:
                        want = o64.lm_steps(x0, H.stacked(target, S), K)
                        Js = o64.lm_step(want, H.stacked(target, S))[1]
                        ok = (np.linalg.svd(Js, compute_uv=False)[:, min(5, d - 1)] >= 5e-2) & (np.abs(want - x0).max(axis=1) < 0.3)
                        if ok.any() and np.abs(x - want)[ok].max() > 5e-3:
                            fail(*tag, "x vs oracle", np.abs(x - want)[ok].max())
                        ref = x
                    elif ref is not None and solver == _hip.SOLVER_AUTO:
                        close = np.abs(x - ref).max(axis=1) < 1e-3
                        if close.mean() < 0.9:
                            fail(*tag, "shapes disagree on", 1 - close.mean(), "of the rows")
    rb.set_obstacles([], [])
print("comparisons:", checked, " disagreements:", bad)
sys.exit(1 if bad else 0)
