"""What shader clock does the MI355X hold while the headline launch runs?  (MI355X_MICROARCH.md, "DVFS give-back".)

A one-wavefront probe kernel (scripts/ubench/clock_probe.hip) samples s_memtime against the constant 100 MHz s_memrealtime on
its own stream while bench.py's Runner keeps two C4 launches in flight on the others; the same probe on an otherwise idle chip
is the comparison.  Prints the median / min / max clock of both.  Developer measurement, not part of the library.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/ubench/clock_probe.hip -o build_var/libclock_probe.so
    python scripts/clock_probe.py [--solver auto|f32] [--streams 2] [--seconds 2.5]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402


def clocks(samples):
    s = samples.cpu().numpy().astype(np.float64).reshape(-1, 2)
    s = s[s[:, 1] > 0]
    return s[:, 0] / s[:, 1] * 100.0  # MHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", default="auto")
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--seeds", type=int, default=1024)
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, "build_var", "libclock_probe.so"))
    lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    device = torch.device("cuda:0")
    robot = get_robot("panda")
    S, W, K = args.seeds, 256, 10
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    obstacles = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    robot.set_obstacles([c for c, _ in obstacles], [T for _, T in obstacles])
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    x0, target, _ = bench.make_inputs_problem(robot, S, W, device, seed=0)
    solver = {"auto": _hip.SOLVER_AUTO, "f32": _hip.SOLVER_F32, "f64": _hip.SOLVER_F64}[args.solver]
    from cppflow_amd.distributed import ShardedRefiner

    run = ShardedRefiner(robot, x0, target, K, n_streams=args.streams, solver=solver)
    probe_stream = torch.cuda.Stream(device=device)
    n_samples, sleeps = 400, 8  # 8 x s_sleep(127) ~ 8 x 8 128 cycles ~ 27 us per sample at 2.4 GHz: ~11 ms per probe launch

    def probe():
        out = torch.zeros(2 * n_samples, dtype=torch.int64, device=device)
        rc = lib.clock_probe_launch(ctypes.c_void_p(probe_stream.cuda_stream), ctypes.c_void_p(out.data_ptr()), n_samples, sleeps)
        assert rc == 0, rc
        return out

    # idle chip
    torch.cuda.synchronize()
    time.sleep(0.5)
    idle = probe()
    torch.cuda.synchronize()
    # under load: keep the launch queue fed for `seconds`, probe in the last third
    t0 = time.perf_counter()
    loaded = []
    n_steps = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(256):
            run.step()
        n_steps += 256
        if time.perf_counter() - t0 > 0.6 * args.seconds and len(loaded) < 3:
            loaded.append(probe())
        # bound the queue depth without draining it
        run.streams[0].synchronize() if n_steps % 2048 == 0 else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"load: {n_steps} steps in {dt:.2f} s = {dt / n_steps * 1e6:.2f} us/step (host-fed, {args.streams} streams, solver {args.solver})")
    ci = clocks(idle)
    print(f"idle chip        : shader clock median {np.median(ci):7.1f} MHz  (min {ci.min():7.1f}, max {ci.max():7.1f}, {len(ci)} samples)")
    for k, o in enumerate(loaded):
        c = clocks(o)
        print(f"under load, probe {k}: shader clock median {np.median(c):7.1f} MHz  (min {c.min():7.1f}, p10 {np.percentile(c, 10):7.1f}, p90 {np.percentile(c, 90):7.1f}, max {c.max():7.1f}, {len(c)} samples)")


if __name__ == "__main__":
    main()
