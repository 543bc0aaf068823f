#!/usr/bin/env python3
"""Distribution of the one-step parity quantities over ALL rows (no conditioning filter), per robot and kernel shape (developer
tool behind the bars of tests/test_gpu_round2.py): task-space difference |J_s (x_gpu - x_64)| against the fp64 reference-order
oracle, the same for the reference-order fp32 oracle (the reference's own arithmetic), pose error after the step."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402
from tests import helpers as H  # noqa: E402

LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
Q = (0.5, 0.9, 0.99, 0.999, 1.0)


def q(a):
    return " ".join(f"{v:9.2e}" for v in np.quantile(a, Q))


for name in ("panda", "fetch", "fetch_arm", "chain12"):
    rb = get_robot(name)
    S, W = 64, 64
    x0, target = H.lm_problem(name, S, W, seed=3)
    tgt = H.stacked(target, S)
    o64, o32 = H.oracle64(name), H.oracle32(name)
    x64, J64, e64, _ = o64.lm_step(x0, tgt, solver=0, **LM)
    x32, _, _, fails32 = o32.lm_step(x0, tgt, solver=0, **LM)
    sv = np.linalg.svd(J64, compute_uv=False)
    smin = sv[:, -1]
    pe64, re64 = o64.pose_metrics_exact(x64, tgt)
    print(f"== {name}: {S * W} rows, sigma_min quantiles {q(smin)}  (fp32 reference-order LU failures: {fails32})")
    ts32 = np.abs(np.einsum("nij,nj->ni", J64, x32 - x64)).max(1)
    print(f"   ref32  |Js dx| {q(ts32)}")
    for shape, nm in ((_hip.SHAPE_ROW, "row "), (_hip.SHAPE_QUAD, "quad")):
        res = rb.lm_pose_steps(torch.tensor(x0, dtype=torch.float32, device="cuda:0"), torch.tensor(target, dtype=torch.float32, device="cuda:0"),
                               n_steps=1, clamp=False, want_errors=True, shape=shape, **LM)
        xg = res["x"].cpu().numpy().astype(np.float64)
        ts = np.abs(np.einsum("nij,nj->ni", J64, xg - x64)).max(1)
        dx = np.abs(xg - x64).max(1)
        pe, re = o64.pose_metrics_exact(xg, tgt)
        ok = smin >= 2e-2
        print(f"   {nm}   |Js dx| {q(ts)}   |dx| {q(dx)}")
        print(f"          well-conditioned rows ({ok.mean():.3f}): |Js dx| max {ts[ok].max():.2e}  |dx| max {dx[ok].max():.2e}  |pos err - pos err64| max {np.abs(pe - pe64)[ok].max():.2e}  rot {np.abs(re - re64)[ok].max():.2e}")
        bound = 2e-5 + 3e-6 * (sv[:, 0] ** 2 + 1e-6) / (smin**2 + 1e-6) * np.linalg.norm(e64, axis=1)
        print(f"          all rows: |pos err - pos err64| {q(np.abs(pe - pe64))}   worst ts / bound {np.max(ts / bound):.3f}  worst ts/ts32 {np.max(ts / np.maximum(ts32, 2e-5)):.2f}")
