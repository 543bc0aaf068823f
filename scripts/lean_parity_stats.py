#!/usr/bin/env python3
"""The K-step iterate of a fused launch against the oracle's K-step iterate, OFF the fixed point (developer tool behind
tests/test_gpu_lean_parity.py; GPU).  Iterations 0 .. K-2 of a plain launch are "lean" (polynomial sine / cosine, relative gate:
csrc/kernels_fused.h); this prints, per robot / input set / K / launch variant, the quantiles of

    ts  = | J_s(x_K^oracle) (x_K - x_K^oracle) |_inf      (scaled task space, fp64 reference-order oracle)
    dpe = | pos_err(x_K) - pos_err(x_K^oracle) |,  dre likewise

on the calm rows (every oracle step of the K below 1 rad), next to the same for the reference-order fp32 oracle, and how many rows
the relative gate declines to re-solve (modelled on the oracle's iterates) and what that changes (gate_rel_ppm = 0 A/B).

    python scripts/lean_parity_stats.py [--rows-s 64] > profiles/r5_lean_parity.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402

LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
QS = (0.5, 0.9, 0.99, 1.0)
DEV = "cuda:0"


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def task_space(Js, dx):
    return np.abs(np.einsum("nij,nj->ni", Js, dx)).max(axis=1)


def oracle_trace(o, x0, tgt, K):
    """iterates x_0 .. x_K of the reference-order oracle (one clamped step at a time), scaled J and e at every linearisation point"""
    xs, Js, es = [x0], [], []
    for _ in range(K):
        _, J, e, _ = o.lm_step(xs[-1], tgt, solver=0, **LM)
        Js.append(J)
        es.append(e.reshape(len(x0), 6))
        xs.append(o.lm_steps(xs[-1], tgt, 1, solver=0, **LM))
    return xs, Js, es


def gate_model(Js, es, lam=1e-6, a_pos=3.5, a_rot=0.35, tau=1e-5, rel=1e-3):
    """the kernel's a-posteriori estimate on the oracle's (J, e): (flagged by the absolute gate, of those declined by the relative one)"""
    sc = np.array([a_rot] * 3 + [a_pos] * 3)
    J = Js / sc[None, :, None]
    e = es / sc[None]
    A = np.einsum("nik,njk->nij", J, J) + np.diag(lam / sc**2)[None]
    y = np.linalg.solve(A, e[..., None])[..., 0]
    est = 6e-8 * max(a_pos, a_rot) * np.diagonal(A, axis1=1, axis2=2).max(1) * np.abs(y).max(1)
    flagged = est > tau
    declined = flagged & ~(est > rel * np.linalg.norm(es, axis=1))
    return flagged, declined


def cases(args):
    from cppflow_amd.robots import get_robot

    for name in ("panda", "fetch", "fetch_arm", "chain12"):
        x0, target = H.lm_problem(name, args.rows_s, 64, seed=21)
        yield name, "seeded", get_robot(name), x0, target, H.stacked(target, args.rows_s)
    # C4's own inputs (bench.py: the reference problem's path + per-seed IK branches), 4096 sampled rows
    import bench

    rb = get_robot("panda")
    target = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_paths.npz"))["panda__2cubes_resampled256"]
    x0, _, _ = bench.make_inputs_problem(rb, 1024, 256, torch.device(DEV), seed=1)
    rows = torch.randperm(1024 * 256, generator=torch.Generator().manual_seed(5))[:4096]
    yield "panda", "C4-sampled", rb, host(x0[rows.to(DEV)]), None, H.f32(target)[(rows % 256).numpy()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-s", type=int, default=64)
    args = ap.parse_args()
    from cppflow_amd import _hip

    for name, source, rb, x0, target, tgt in cases(args):
        o64, o32 = H.oracle64(name), H.oracle32(name)
        n = len(x0)
        # rows with their own target: launch with W = n (the stacked form)
        tgt_dev, x0_dev = dev(tgt), dev(x0)
        for K in (2, 3, 5):
            xs, Js, es = oracle_trace(o64, x0, tgt, K)
            xo = xs[-1]
            x32 = o32.lm_steps(x0, tgt, K, solver=0, **LM)
            _, JK, _, _ = o64.lm_step(xo, tgt, solver=0, **LM)
            steps = np.max([np.abs(xs[k + 1] - xs[k]).max(axis=1) for k in range(K)], axis=0)
            calm = steps < 1.0
            pe_o, re_o = o64.pose_metrics_exact(xo, tgt)
            fl = np.zeros(n, bool)
            de = np.zeros(n, bool)
            for k in range(K - 1):  # the lean iterations
                f, d = gate_model(Js[k], es[k])
                fl |= f
                de |= d
            ts32 = task_space(JK, x32 - xo)
            print(f"== {name} {source} K={K}: {n} rows, calm {calm.mean():.3f}; lean iterations: flagged {fl.mean():.4f}, declined by the relative gate {de.mean():.4f}")
            print(f"   reference-order fp32 oracle   ts q{QS} = {np.quantile(ts32[calm], QS)}")
            variants = [("row  auto", _hip.SHAPE_ROW, _hip.SOLVER_AUTO), ("row  f64 ", _hip.SHAPE_ROW, _hip.SOLVER_F64),
                        ("quad auto", _hip.SHAPE_QUAD, _hip.SOLVER_AUTO), ("quad f64 ", _hip.SHAPE_QUAD, _hip.SOLVER_F64)]
            got = {}
            for label, shape, solver in variants:
                r = rb.lm_pose_steps(x0_dev, tgt_dev, n_steps=K, want_errors=True, shape=shape, solver=solver, **LM)
                x = host(r["x"])
                got[label] = x
                ts = task_space(JK, x - xo)
                pe, re = o64.pose_metrics_exact(x, tgt)
                dxj = np.abs(x - xo).max(axis=1)
                print(f"   {label}: ts q = {np.quantile(ts[calm], QS)}  all-rows max {ts.max():.2e};  dpe q = {np.quantile(np.abs(pe - pe_o)[calm], QS)}  "
                      f"dre max {np.abs(re - re_o)[calm].max():.2e}; dx max(calm) {dxj[calm].max():.2e}; ts on declined rows max {ts[de & calm].max() if (de & calm).any() else 0:.2e}")
            # batch entry (row shape): bit-identical to the plain launch
            xb = torch.empty_like(x0_dev)
            rb.lm_batch_plan([dict(x=x0_dev, target=tgt_dev, x_out=xb)], n_steps=K, **LM).launch()
            torch.cuda.synchronize()
            print(f"   batch entry == row auto bit for bit: {np.array_equal(host(xb), got['row  auto'])}")
            # the relative gate off (absolute gate in every iteration): what declining costs
            rb.debug_set("gate_rel_ppm", 0)
            xa = host(rb.lm_pose_steps(x0_dev, tgt_dev, n_steps=K, shape=_hip.SHAPE_ROW, **LM)["x"])
            rb.debug_set("gate_rel_ppm", None)
            tsa = task_space(JK, xa - got["row  auto"])
            print(f"   row auto, relative gate on vs off: rows that differ {np.mean(np.abs(xa - got['row  auto']).max(axis=1) > 0):.4f}, ts max {tsa.max():.2e} (calm {tsa[calm].max():.2e})")
            # canonical K steps through K launches of one step (every iteration canonical + absolute gate) vs the fused K-step launch
            xc = x0_dev
            for _ in range(K):
                xc = rb.lm_pose_steps(xc, tgt_dev, n_steps=1, shape=_hip.SHAPE_ROW, **LM)["x"]
            tsc = task_space(JK, host(xc) - got["row  auto"])
            print(f"   K launches of one canonical step vs one launch of K: ts q = {np.quantile(tsc[calm], QS)}")


if __name__ == "__main__":
    main()
