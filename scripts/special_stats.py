import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import helpers as H
from tests.test_gpu_parity_allrows import _one_step_case, _task_space, LM, dev, host
from cppflow_amd import _hip
from cppflow_amd.robots import get_robot
np.set_printoptions(precision=2, linewidth=200)
for name in ("panda","fetch","fetch_arm","chain12"):
    x0, target, S = _one_step_case(name, "special")
    tgt = H.stacked(target, S)
    o64, o32 = H.oracle64(name), H.oracle32(name)
    x64, Js, es, fails = o64.lm_step(x0, tgt, solver=0, **LM)
    x32, _, _, _ = o32.lm_step(x0, tgt, solver=0, **LM)
    ts32 = _task_space(Js, x32 - x64)
    sv = np.linalg.svd(Js, compute_uv=False)
    enorm = np.abs(es).reshape(len(x0), -1).max(axis=1)
    rb = get_robot(name)
    print("==", name, "fails", fails)
    for solver, sn in ((_hip.SOLVER_F64,"f64"),(_hip.SOLVER_AUTO,"auto"),(_hip.SOLVER_F32,"f32")):
        x = host(rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, solver=solver, shape=_hip.SHAPE_ROW, **LM)["x"])
        ts = _task_space(Js, x - x64)
        W = 64
        print(sn, "max per block:", np.array([ts[b*W:(b+1)*W].max() for b in range(S)]))
    print("ref32 max per block:", np.array([ts32[b*64:(b+1)*64].max() for b in range(S)]))
    print("smin  min per block:", np.array([sv[b*64:(b+1)*64,-1].min() for b in range(S)]))
    print("|e_s| max per block:", np.array([enorm[b*64:(b+1)*64].max() for b in range(S)]))
