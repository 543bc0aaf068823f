#!/usr/bin/env python3
"""Which HIP streams should a ShardedRefiner's buckets run on?  (developer measurement, GPU)

The same 20-step region of a 128-seed shard (8 / 8 / 4-step launches, one-rank RCCL exchange behind each bucket) takes 5.8 ... 8.3 us per
step depending on WHICH two streams carry it (profiles/r4_start_bucket.txt; calibrate_streams() picks the best of 30 pairs).  A
deterministic default would be better than a search: this script times the region for a few stream-creation STRATEGIES, each in a
fresh process (the stream -> hardware-queue mapping depends on creation order within the process):

    python scripts/stream_choice.py            # parent: runs every strategy as a child, prints a table
    python scripts/stream_choice.py <strategy> # child

strategies: first2 (what ShardedRefiner does today), skip2 / skip4 / skip8 (that many streams created and dropped first), hiprio
(priority = -1), hiprio_skip2, start0 / start1 (first2 with every region restarted on bucket 0 / 1)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STRATEGIES = ["first2", "skip2", "skip4", "skip8", "hiprio", "hiprio_skip2", "start0", "start1"]


def child(strategy, seeds=128, steps=20):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + STRATEGIES.index(strategy)), RANK="0", WORLD_SIZE="1")
    import time

    import numpy as np
    import torch
    import torch.distributed as dist

    from cppflow_amd import distributed as D
    from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, make_inputs_problem, obstacle_arrays
    from cppflow_amd.robots import get_robot
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    transport, _ = D.pick_transport(dev)
    rb = get_robot("panda")
    obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    x_all, target, _ = make_inputs_problem(rb, 1024, 256, dev, seed=0)
    x0 = x_all[: seeds * 256].contiguous()
    B, G, _, n_streams = D.launch_plan(x0.shape[0], steps)
    skip = {"skip2": 2, "skip4": 4, "skip8": 8, "hiprio_skip2": 2}.get(strategy, 0)
    dummies = [torch.cuda.Stream(device=dev) for _ in range(skip)]  # noqa: F841 -- held: their queues stay taken
    r = D.ShardedRefiner(rb, x0, target, 10, transport=transport, batch=B, bucket=G, n_streams=n_streams)
    if strategy.startswith("hiprio"):
        cur = torch.cuda.current_stream(dev)
        r.streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(n_streams)]
        for st in r.streams:
            st.wait_stream(cur)
    if strategy in ("start0", "start1"):
        r.start_bucket = int(strategy[-1])
    r.prewarm(60.0)
    r.run_steps(5)
    r.drain()
    ts = []
    for _ in range(41):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run_region(steps)
        r.synchronize()
        ts.append(1e6 * (time.perf_counter() - t0) / steps)
    transport.close()
    dist.destroy_process_group()
    os.dup2(saved, 1)
    ts = np.array(ts)
    print("RESULT " + json.dumps({"strategy": strategy, "us_per_step_median": float(np.median(ts)), "p10": float(np.quantile(ts, 0.1)),
                                  "p90": float(np.quantile(ts, 0.9)), "even_regions": float(np.median(ts[0::2])), "odd_regions": float(np.median(ts[1::2]))}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for rep in range(2):
            for s in STRATEGIES:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
                print(line[-1][7:] if line else f"{s}: FAILED {p.stderr[-300:]}", flush=True)
