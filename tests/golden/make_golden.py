#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container, where /root/reference is mounted):

  reference_problems.npz  all 18 problems of the reference's problem set (robot, offset target path, cuboid obstacles): data only
  reference_paths.npz   target paths of BASELINE.json configs C1-C4, produced by cppflow_amd.data_type_utils from the
                        reference's own problem yaml / path csv DATA files (no reference source is read or copied)
  lm_golden_<robot>.npz 64 seeded rows per robot: x, target, and what the fp64 oracle (oracle/lmik_oracle.c, reference
                        operation order) returns for them -- pose error, scaled J, one LM step, 10 fused steps + metrics,
                        self / env distances and masks.  The oracle is the ground truth of this build (the reference
                        cannot be imported here: jrl is not vendored), so these files pin it against regressions and are
                        what the GPU tests compare with when run without /root/reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cppflow_amd.data_type_utils import problem_from_filename, resample_path  # noqa: E402
from tests import helpers as H  # noqa: E402

REF = "/root/reference/cppflow"


def reference_paths():
    kw = dict(problems_dir=os.path.join(REF, "problems"), paths_dir=os.path.join(REF, "paths"), device="cpu")
    c1 = problem_from_filename(None, "", filepath_override="/root/reference/tests/fetch_arm__s__truncated.yaml", **kw)
    c2 = problem_from_filename(None, "panda__1cube", **kw)
    c3 = problem_from_filename(None, "fetch__hello", **kw)
    c4 = problem_from_filename(None, "panda__2cubes", **kw)
    np.savez_compressed(
        os.path.join(HERE, "reference_paths.npz"),
        fetch_arm__s__truncated=c1.target_path.numpy(),
        panda__1cube_first64=c2.target_path.numpy()[:64],
        fetch__hello_first256=c3.target_path.numpy()[:256],
        panda__2cubes_resampled256=resample_path(c4.target_path.numpy().astype(np.float64), 256).astype(np.float32),
        panda__2cubes_obstacles=np.array([[0.2, 0.3, 0.4, 0.15, 0.15, 0.15], [-0.25, 0.3, 0.75, 0.15, 0.15, 0.15]]),
        panda__1cube_obstacles=np.array([[0.0, 0.2, 0.7, 0.25, 0.25, 0.25]]),
    )


def reference_problems():
    """every problem of the reference's problem set (cppflow/problems/*.yaml, 18 files): robot, target path with its offset applied by
    this build's loader, cuboid obstacles [O, 6] = (x, y, z, size_x, size_y, size_z) -- DATA only (scripts/problem_plausibility.py and
    tests/test_problem_plausibility.py read it; /root/reference does not exist on the GPU box)"""
    import glob

    import yaml

    kw = dict(problems_dir=os.path.join(REF, "problems"), paths_dir=os.path.join(REF, "paths"), device="cpu")
    out, names = {}, []
    for f in sorted(glob.glob(os.path.join(REF, "problems", "*.yaml"))):
        name = os.path.splitext(os.path.basename(f))[0]
        pr = problem_from_filename(None, name, **kw)
        d = yaml.load(open(f), Loader=yaml.FullLoader)
        obs = [[o["x"], o["y"], o["z"], o["size_x"], o["size_y"], o["size_z"]] for o in pr.obstacles]  # (offset applied by the loader)
        names.append(name)
        out[name + "__robot"] = np.array(d["robot"])
        out[name + "__target_path"] = pr.target_path.numpy().astype(np.float32)
        out[name + "__obstacles"] = np.array(obs, dtype=np.float64).reshape(-1, 6)
    np.savez_compressed(os.path.join(HERE, "reference_problems.npz"), names=np.array(names), **out)


def lm_golden(name):
    S, W, K = 4, 16, 10
    x0, target = H.lm_problem(name, S, W, seed=100)
    tgt = H.stacked(target, S)
    o = H.oracle64(name)
    e, cur = o.pose_errors(x0, tgt)
    x1, J, es, _ = o.lm_step(x0, tgt, solver=0)
    xK = o.lm_steps(x0, tgt, K, solver=0)
    pe, re = o.pose_metrics_exact(xK, tgt)
    obs = H.PANDA_2CUBES
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    ch = H.chain(name)
    q = H.random_configs(name, 64, seed=101)
    m = H.oracle32(name).masks(q, lo, hi, ch.lo, ch.hi)
    np.savez_compressed(
        os.path.join(HERE, f"lm_golden_{name}.npz"), x0=x0, target=target, S=S, W=W, K=K, e=e, fk=cur, J_scaled=J,
        e_scaled=es, x_step1=x1, x_stepK=xK, pos_err_K=pe, rot_err_K=re, q_coll=q, self_dists=o.self_dists(q),
        env_dists_box0=o.env_dists(q, lo[0], hi[0]), self_mask=m["self_mask"], env_mask=m["env_mask"],
        min_self_f32=m["min_self"], min_env_f32=m["min_env"], box_lo=lo, box_hi=hi,
    )  # fmt: skip


if __name__ == "__main__":
    from oracle import oracle

    oracle.build()
    if os.path.isdir(REF):
        reference_paths()
        reference_problems()
    for n in ("panda", "fetch", "fetch_arm", "chain12"):
        lm_golden(n)
    print("golden fixtures written to", HERE)
