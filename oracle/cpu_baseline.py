"""The CPU legs of bench.py (`cpu_baseline`, `cpu_baseline_c`): the oracle -- TEST INFRASTRUCTURE, never the product path -- timed on the
GPU box's host cores, on the first seeds of the very x0 / target the GPU run timed, with an agreement check
against the GPU's pose errors riding along.  Only bench.py calls this module."""

import os
import time

import numpy as np
import torch


def cpu_baseline_torch(robot_name, obstacles, x0_host, target_host, gpu_pos_err, gpu_rot_err, K, budget_s=12.0):
    """The reference-equivalent CPU path: oracle/ref_torch.py issues the reference's own torch op sequence
    (cppflow/optimization.py:73-92: in-place row scaling, bmm x2, eye.repeat, torch.linalg.solve, python-loop clamp;
    cppflow/collision_detection.py:27-69: distance tensors -> min -> "< 0") with batched-torch kinematics standing in for the
    un-vendored jrl.  fp32, on the FIRST S_cpu seeds of the very x0 / target the GPU timed, with an
    agreement check riding along: max |pose error_cpu - pose error_gpu| over the rows of the sample both sides converge on."""
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import ref_torch

    W, d = target_host.shape[0], x0_host.shape[1]
    rb = ref_torch.TorchRobot(ROBOT_SPECS[robot_name](), device="cpu", dtype=torch.float32)
    cub, Ts = [torch.tensor(c) for c, _ in obstacles], [torch.tensor(T) for _, T in obstacles]
    eps_r, eps_p = float(np.deg2rad(1.5)), 0.03
    S_have = x0_host.shape[0] // W

    def run(S_cpu):
        x, target = x0_host[: S_cpu * W].clone(), target_host.repeat(S_cpu, 1)  # (the reference's vstack, optimization.py:399-401)
        t0 = time.perf_counter()
        x = ref_torch.lm_pose_steps(rb, x, target, K)
        pe, re = ref_torch.calculate_pose_error_m_rad(rb, x, target)
        ref_torch.q_costs_external(rb, x.reshape(S_cpu, W, d), cub, Ts, eps_r, eps_p)
        return time.perf_counter() - t0, pe, re

    # torch's default (all cores) is far from optimal for these small batched ops on a many-core host: probe a few thread counts
    run(1)
    cores = os.cpu_count() or 1
    best_threads, t_probe = None, None
    for th in sorted({min(cores, c) for c in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(th)
        tt = run(min(32, S_have))[0]
        if t_probe is None or tt < t_probe:
            best_threads, t_probe = th, tt
    torch.set_num_threads(best_threads)
    S_cpu = int(max(1, min(S_have, 32 * budget_s / max(t_probe, 1e-6))))
    t, pe, re = run(S_cpu)
    n = S_cpu * W
    agreement = _agreement(pe.numpy(), re.numpy(), gpu_pos_err[:n].numpy(), gpu_rot_err[:n].numpy())
    threads = torch.get_num_threads()
    return {
        "value": S_cpu * W * K / t, "unit": "LM-IK iterations/s", "cores": threads, "kind": "port",
        "sample": f"the first {S_cpu} seeds x {W} waypoints of the x0 / target the GPU run timed (problem inputs), {K} LM iterations + pose metrics + "
        f"collision masks + cost; torch-CPU restatement of the reference's op sequence (oracle/ref_torch.py; jrl is not vendored so the "
        f"reference itself cannot run), fp32, {threads} torch threads of {os.cpu_count()} host cores, {t:.2f} s",
        "agreement": agreement,
    }


def _agreement(pe_c, re_c, pe_g, re_g):
    """|pose error after K steps, CPU - GPU| on the sample (untimed).  Two row sets: `converged` = below 1e-4 m on both sides (rows still
    contracting by a factor per iteration sit anywhere below that bar on either side, so their difference is bounded by the bar, not by
    the arithmetic), and `settled` = at the fp32 floor on both sides (< 5e-6 m, the set tests/test_gpu_parity_allrows.py compares): THAT
    maximum is the agreement figure BASELINE.md section 3 asks for (1e-5)."""
    pe_c, re_c, pe_g, re_g = (np.asarray(a, dtype=np.float64) for a in (pe_c, re_c, pe_g, re_g))
    conv = (pe_c < 1e-4) & (pe_g < 1e-4)
    settled = (pe_c < 5e-6) & (pe_g < 5e-6)
    dp, dr = np.abs(pe_c - pe_g), np.abs(re_c - re_g)

    def q(a, m):
        return [float(v) for v in np.quantile(a[m], (0.5, 0.99, 1.0))] if m.any() else None

    return {"rows_in_sample": int(pe_c.size), "rows_converged_both": int(conv.sum()), "rows_settled_both": int(settled.sum()),
            "max_abs_pos_err_diff_m": float(dp[settled].max()) if settled.any() else None,
            "max_abs_rot_err_diff_rad": float(dr[settled].max()) if settled.any() else None,
            "converged_rows_pos_err_diff_m_p50_p99_max": q(dp, conv), "converged_rows_rot_err_diff_rad_p50_p99_max": q(dr, conv),
            "converged_only_on_cpu": int(((pe_c < 1e-4) & ~(pe_g < 1e-4)).sum()), "converged_only_on_gpu": int((~(pe_c < 1e-4) & (pe_g < 1e-4)).sum()),
            "what": "max |pose error after K steps, CPU - GPU| over the sample's rows settled at the fp32 floor (< 5e-6 m) on both sides; "
                    "quantiles over the rows converged (< 1e-4 m) on both sides beside it; untimed"}


def cpu_baseline_c(robot_name, obstacles, x0_host, target_host, gpu_pos_err, K, budget_s=6.0):
    """The C restatement (oracle/lmik_oracle.c, canonical fp32 build, LU solve in reference order), OpenMP over rows, same sample rule."""
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import oracle as orc

    orc.build()
    cores = os.cpu_count() or 1
    chain = canonicalize(ROBOT_SPECS[robot_name]())
    o = orc.Oracle(chain, f32=True, threads=cores)
    lo_b = np.array([np.float32(T[:3, 3]) + np.float32(c[:3]) for c, T in obstacles], dtype=np.float64).reshape(-1, 3)
    hi_b = np.array([np.float32(T[:3, 3]) + np.float32(c[3:]) for c, T in obstacles], dtype=np.float64).reshape(-1, 3)
    W = target_host.shape[0]
    x_all, tgt1 = x0_host.numpy().astype(np.float64), target_host.numpy().astype(np.float64)
    S_have = x_all.shape[0] // W

    def run(S_cpu):
        x0, tgt = x_all[: S_cpu * W], np.tile(tgt1, (S_cpu, 1))
        t0 = time.perf_counter()
        x = o.lm_steps(x0, tgt, K, 1e-6, 3.5, 0.35, solver=0)
        pe, _ = o.pose_metrics(x, tgt)
        o.masks(x, lo_b, hi_b, chain.lo, chain.hi)
        return time.perf_counter() - t0, pe

    t_probe = run(min(64, S_have))[0]
    S_cpu = int(max(1, min(S_have, 64 * budget_s / max(t_probe, 1e-6))))
    t, pe = run(S_cpu)
    n = S_cpu * W
    g = gpu_pos_err[:n].numpy().astype(np.float64)
    settled = (pe < 5e-6) & (g < 5e-6)
    return {
        "value": S_cpu * W * K / t, "unit": "LM-IK iterations/s", "cores": cores, "kind": "port",
        "sample": f"the first {S_cpu} seeds x {W} waypoints of the GPU run's x0 / target, {K} LM iterations + pose metrics + collision masks; scalar C "
        f"restatement (oracle/lmik_oracle.c, fp32 canonical build), OpenMP {cores} threads, {t:.2f} s",
        "agreement": {"max_abs_pos_err_diff_m": float(np.abs(pe - g)[settled].max()) if settled.any() else None, "rows_settled_both": int(settled.sum()),
                      "rows_in_sample": n},
    }
