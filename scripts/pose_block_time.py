import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs_problem
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
rb = get_robot("panda")
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
d = dict(ALT_LOSS_V2_1_DIFF.__dict__); d.update(use_pose=True, alpha_position=1.1, alpha_rotation=1.0)
pm = OptimizationParameters(**d)
from cppflow_amd import _hip
for S in (1, 64, 1024):
    x0, target, _ = make_inputs_problem(rb, S, 256, dev, 0)
    fn = lambda: rb.lm_full_step(x0, target, pm)
    outs = {}
    for mode in (0, 1):
        rb.debug_set("rows_pose", mode)
        outs[mode] = fn().clone(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): fn()
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5 * 1e3)
        print(f"panda coupled step WITH the pose block, S={S} T=256, {'row-per-lane kernels' if mode else 'one lane per trajectory'}: {np.median(ts):.1f} us", flush=True)
    rb.debug_set("rows_pose", 0)
    fk0, fk1 = rb.forward_kinematics(outs[0]), rb.forward_kinematics(outs[1])
    print(f"   |x_rows - x_lane| max {float((outs[1]-outs[0]).abs().max()):.3g}  median {float((outs[1]-outs[0]).abs().median()):.3g};  end-effector position difference max {float((fk1[:, :3]-fk0[:, :3]).abs().max()):.3g} m;  step max {float((outs[0]-x0).abs().max()):.3g}", flush=True)
