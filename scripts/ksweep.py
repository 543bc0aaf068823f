import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
bench.torch = torch
from cppflow_amd import _hip
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
rb = get_robot("panda")
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs]); rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
x_all, target, _ = bench.make_inputs_problem(rb, 64, 256, dev, 0)
def med(fn, reps=300):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
warm = rb.lm_launch_plan(x_all, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=torch.empty_like(x_all))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _ in range(50): warm.launch()
    torch.cuda.synchronize()
for S in (8, 32, 64):
    x0 = x_all[: S * 256].contiguous(); n = S * 256
    for shape, nm in ((_hip.SHAPE_ROW, "row "), (_hip.SHAPE_QUAD, "quad")):
        for coll in (0, 1):
            ts = []
            for K in (1, 10, 20):
                pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=dev) if coll else None
                p = rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=K, x_out=torch.empty_like(x0), packed_out=pk, shape=shape)
                ts.append(med(p.launch))
            print(f"{nm} rows {n:6d} coll {coll}: K=1 {ts[0]:6.1f}  K=10 {ts[1]:6.1f}  K=20 {ts[2]:6.1f}   per iteration {(ts[2]-ts[1])/10:5.2f} us")
