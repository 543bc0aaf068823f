#!/usr/bin/env python3
"""What run-time specialisation buys (developer tool): the Panda description with ONE chain constant moved by one ulp matches no
generated table; its fused C4 launch is timed with the generic kernels, with the kernels hipRTC compiled for it, and next to the
shipped Panda table build.  Also the compile time and the cache-hit time."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robot_zoo import ROBOT_SPECS  # noqa: E402
from cppflow_amd.robots import Robot, get_robot  # noqa: E402

os.environ["CPPF_CACHE_DIR"] = tempfile.mkdtemp()
dev = torch.device("cuda:0")


def almost_panda():
    spec = ROBOT_SPECS["panda"]()
    import dataclasses

    j = spec.joints[2]
    spec.joints[2] = dataclasses.replace(j, xyz=(j.xyz[0], float(np.nextafter(np.float32(j.xyz[1]), np.float32(0))), j.xyz[2]))
    spec.name = "panda_1ulp"
    return spec


def time_launch(rb, x0, target, S, W):
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    plan = rb.lm_launch_plan(x0, target, 1e-6, 3.5, 0.35, n_steps=10, x_out=torch.empty_like(x0),
                             packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev),
                             summary_out=torch.empty((S, 8), device=dev))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        for _ in range(50):
            plan.launch()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
    for a, b in ev:
        a.record()
        plan.launch()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def main():
    S, W = 1024, 256
    panda = get_robot("panda")
    x0, target, _ = bench.make_inputs_problem(panda, S, W, dev, 0)
    t_table = time_launch(panda, x0, target, S, W)
    t0 = time.perf_counter()
    fast = Robot(almost_panda(), specialize=True)
    spec_id = fast.specialization(dev)
    t_compile = time.perf_counter() - t0
    t0 = time.perf_counter()
    again = Robot(almost_panda(), specialize=True)
    again.specialization(dev)
    t_hit = time.perf_counter() - t0
    slow = Robot(almost_panda(), specialize=False)
    t_fast, t_slow = time_launch(fast, x0, target, S, W), time_launch(slow, x0, target, S, W)
    print(f"fused C4 launch (K=10 + collision + summary, problem inputs, isolated, median of 300):")
    print(f"  shipped Panda table (compiled into the library)        {t_table:7.1f} us")
    print(f"  Panda + 1 ulp, kernels compiled by hipRTC (id {spec_id})   {t_fast:7.1f} us   ({100 * (t_fast / t_table - 1):+.1f} % vs the table build)")
    print(f"  Panda + 1 ulp, generic kernels                         {t_slow:7.1f} us")
    print(f"  handle creation incl. hipRTC compile {t_compile:.2f} s; with the cached code object {t_hit:.3f} s")


if __name__ == "__main__":
    main()
