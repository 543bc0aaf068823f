#!/usr/bin/env python3
"""Developer measurement for the fair-share pacing experiment: needs build_var/lib_pace.so (scripts/make_pace_build.py), run with
CPPFLOW_HIP_LIB=build_var/lib_pace.so.  For each pace (CPPF_PACE, 10 ns ticks per LM iteration; > 0 sleep, < 0 priority only, 0 off):
the isolated C4 launch, one stream back to back, two streams alternating, and when the four wavefront bands of an isolated launch end.
Results must not depend on the pace: x is compared bit for bit with the pace-0 run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

DEV = torch.device("cuda:0")
paces = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "fair5", "fair6", "fair7", "0", "fair5", "-300/4", "fair7", "0"]  # "-P" = two-level schedule priority, "-P/4" four levels (CPPF_FAIR=4), "fairN" = lag-ranked priority through a table, CPPF_FAIR=N
rb = get_robot("panda")
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
S, W, K = 1024, 256, 10
x0, target, _ = bench.make_inputs_problem(rb, S, W, DEV, seed=0)
n = S * W
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
bufs = []
for _ in range(2):
    bufs.append((torch.empty_like(x0), torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV), torch.empty((S, 8), dtype=torch.float32, device=DEV)))
streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]


def launch(i=0, **kw):
    xo, pk, sm = bufs[i]
    return rb.lm_pose_steps(x0, target, n_steps=K, x_out=xo, packed_out=pk, summary_out=sm, **LM, **kw)


def ev():
    return torch.cuda.Event(enable_timing=True)


for _ in range(2000):  # sustained clocks
    launch()
torch.cuda.synchronize()
ref = None
for pace in paces:
    os.environ["CPPF_PACE"] = "0" if pace.startswith("fair") else pace.split("/")[0]
    os.environ["CPPF_FAIR"] = pace[4:] if pace.startswith("fair") else (pace.split("/")[1] if "/" in pace else "0")
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    iso = []
    for _ in range(200):
        a, b = ev(), ev()
        torch.cuda.synchronize()
        a.record()
        launch()
        b.record()
        torch.cuda.synchronize()
        iso.append(a.elapsed_time(b) * 1e3)
    a, b = ev(), ev()
    a.record()
    for _ in range(400):
        launch()
    b.record()
    torch.cuda.synchronize()
    one = a.elapsed_time(b) * 1e3 / 400
    two = []
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for i in range(1000):
            with torch.cuda.stream(streams[i & 1]):
                launch(i & 1)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        b.record()
        torch.cuda.synchronize()
        two.append(a.elapsed_time(b) * 1e3 / 1000)
    r = launch(want_iters=True)
    torch.cuda.synchronize()
    x = bufs[0][0].clone()
    if ref is None:
        ref = x
    same = bool(torch.equal(x, ref))
    v = r["n_iters"].cpu().numpy().astype(np.int64).reshape(-1, 64)
    start, end = v[:, 0] & 0xFFFF, (v[:, 0] >> 16) & 0xFFFF
    t0 = np.sort(start)[0]
    en = ((end - t0) & 0xFFFF) * 0.01
    wg = np.arange(len(en)) // 4
    bands = [float(np.median(en[(wg >= lo) & (wg < lo + 256)])) for lo in (0, 256, 512, 768)]
    print(f"pace {pace:>7s}: isolated {np.median(iso):6.2f} us (min {np.min(iso):6.2f})  one stream {one:6.2f}  two streams {np.median(two):6.2f} (min {np.min(two):6.2f})  "
          f"band ends {' / '.join('%.1f' % b_ for b_ in bands)}  last {en.max():.1f}  x identical: {same}", flush=True)
