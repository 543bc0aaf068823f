#!/usr/bin/env python3
"""Plausibility of the build's OWN robot models against the reference's problem set.

The reference holds no number that pins a capsule (jrl is un-vendored), but it does hold 18 planning problems it solves
(cppflow/problems/*.yaml: target paths + obstacles for Panda / Fetch / FetchArm).  A robot model under which those problems have no
collision-free solution would be wrong whatever else is right, so for every problem and every waypoint this script solves IK with
the CPU oracle from R random starts inside the joint limits (damped LM, the reference's own step + clamp, oracle/lmik_oracle.c) and
records
  reachable            waypoints with at least one IK solution inside the limits (pose error < 1e-4 m, < 1e-3 rad),
  free                 waypoints with at least one solution that is also free of self- and environment collisions -- what
                       cppflow/planners.py:237, 247 needs to make progress --
  self / env / jlim    the share of ALL IK solutions found that the capsule model flags (each priced 1000 / 1000 / 100 by dp_search,
                       cppflow/search.py:14-15), and which capsule pairs raise the self-collision flag.
The inputs are tests/golden/reference_problems.npz (the reference's DATA through this build's loader; tests/golden/make_golden.py).

    python scripts/problem_plausibility.py [--restarts 32] [--write]      (--write: tests/golden/problem_plausibility.json)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def solve_problem(name, robot, target, obstacles, restarts, seed=0):
    """IK for every waypoint from `restarts` random starts; returns the per-problem record."""
    from cppflow_amd.robot_zoo import ROBOT_SPECS

    ch, o = H.chain(robot), H.oracle64(robot)
    W, d = target.shape[0], ch.ndof
    rng = np.random.RandomState(seed)
    x = H.f32(rng.uniform(ch.lo + 0.05 * (ch.hi - ch.lo), ch.hi - 0.05 * (ch.hi - ch.lo), size=(restarts, W, d))).reshape(restarts * W, d)
    tgt = np.tile(H.f32(target), (restarts, 1))
    x = o.lm_steps(x, tgt, 60, lm_lambda=1e-2, alpha_position=3.5, alpha_rotation=0.35, solver=0)
    x = o.lm_steps(x, tgt, 25, lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35, solver=0)
    pe, re = o.pose_metrics_exact(x, tgt)
    solved = ((pe < 1e-4) & (re < 1e-3)).reshape(restarts, W)
    if len(obstacles):
        lo = np.array([[ob[0] - ob[3] / 2, ob[1] - ob[4] / 2, ob[2] - ob[5] / 2] for ob in obstacles])
        hi = np.array([[ob[0] + ob[3] / 2, ob[1] + ob[4] / 2, ob[2] + ob[5] / 2] for ob in obstacles])
    else:
        lo = hi = None
    pad = np.where(ch.jtype == 1, 0.03, np.deg2rad(1.5))  # search.py:20-21
    m = o.masks(x, lo, hi, ch.lo + pad, ch.hi - pad)
    self_hit, env_hit, jl = (m[k].astype(bool).reshape(restarts, W) for k in ("self_mask", "env_mask", "jlim_mask"))
    free = solved & ~self_hit & ~env_hit
    n_sol = max(int(solved.sum()), 1)
    caps = [c.link for c in ROBOT_SPECS[robot]().capsules]
    dists = o.self_dists(x).reshape(restarts, W, -1)
    pair_share = {}
    for p in range(ch.n_pairs):
        share = float(((dists[:, :, p] < 0) & solved).sum() / n_sol)
        if share >= 0.005:
            pair_share[f"{caps[ch.pairs[p][0]]} -- {caps[ch.pairs[p][1]]}"] = round(share, 4)
    return {
        "robot": robot, "waypoints": int(W), "obstacles": int(len(obstacles)), "restarts": int(restarts),
        "reachable_frac": float(solved.any(axis=0).mean()), "free_frac": float(free.any(axis=0).mean()),
        "solutions_per_waypoint_mean": float(solved.sum(axis=0).mean()),
        "solutions_self_colliding_frac": float((self_hit & solved).sum() / n_sol),
        "solutions_env_colliding_frac": float((env_hit & solved).sum() / n_sol),
        "solutions_jlim_padding_frac": float((jl & solved).sum() / n_sol),
        "free_solutions_per_waypoint_min": int(free.sum(axis=0).min()),
        "self_colliding_pairs_share_of_solutions": dict(sorted(pair_share.items(), key=lambda kv: -kv[1])),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--restarts", type=int, default=32)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from oracle import oracle

    oracle.build()
    z = np.load(os.path.join(GOLDEN, "reference_problems.npz"), allow_pickle=False)
    names = [str(n) for n in z["names"]]
    table = {}
    for n in names:
        if args.only and args.only not in n:
            continue
        rec = solve_problem(n, str(z[n + "__robot"]), z[n + "__target_path"], z[n + "__obstacles"], args.restarts)
        table[n] = rec
        print(f"{n:24s} W {rec['waypoints']:4d}  reachable {rec['reachable_frac']:.3f}  free {rec['free_frac']:.3f}  "
              f"IK solutions flagged: self {rec['solutions_self_colliding_frac']:.3f} env {rec['solutions_env_colliding_frac']:.3f} "
              f"jlim {rec['solutions_jlim_padding_frac']:.3f}  min free/waypoint {rec['free_solutions_per_waypoint_min']}", flush=True)
    if args.write:
        with open(os.path.join(GOLDEN, "problem_plausibility.json"), "w") as f:
            json.dump({"_how": "scripts/problem_plausibility.py --write (CPU oracle; see its docstring)", "problems": table}, f, indent=1)


if __name__ == "__main__":
    main()
