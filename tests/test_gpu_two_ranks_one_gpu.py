"""The N > 1 PRODUCT path on the hardware there is (pytest -m gpu): two rank processes share the one GPU, gloo as the process group
(RCCL refuses two ranks on one device; `cppflow_amd.distributed` stages a HIP tensor's all-gather through the host when the group's
backend cannot take device tensors).  Meaningless for timing, exact for semantics: the sharded `Planner` pipeline
(`sharded_candidate_evaluation`: every rank evaluates its slice of the candidates, all-gathers the packed per-row outputs and the
paths, runs `dp_search` over ALL of them) must return on EVERY rank the plan a single process returns over the same candidates, and
a `ShardedRefiner` with the host-staged transport must select the seed a single process selects."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, {root!r})
rank, world = int(sys.argv[1]), 2
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str({port}), RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np, torch, torch.distributed as dist
from cppflow_amd import distributed as D
from cppflow_amd.data_type_utils import problem_from_filename
from cppflow_amd.data_types import Constraints, PlannerSettings
from cppflow_amd.planners import PlannerSearcher
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
z = np.load({npz!r})
qs_all = torch.tensor(z["qs"], dtype=torch.float32, device="cuda:0")
problem = problem_from_filename(None, "panda__line", robot=None, device="cuda:0")
k = qs_all.shape[0]

def provider(problem, k_local):  # this rank's slice of the SAME candidates the single process plans over
    assert k_local == k // world
    return qs_all[rank * k_local : (rank + 1) * k_local].contiguous()

out = {{}}
for lm_steps in (0, 2):
    searcher = PlannerSearcher(PlannerSettings(k=k, tmax_sec=30.0, anytime_mode_enabled=False, verbosity=0), problem.robot, provider, candidate_lm_steps=lm_steps)
    res = searcher.generate_plan(problem)
    out[f"path{{lm_steps}}"] = res.plan.q_path.cpu().numpy().tolist()
# the streaming engine with the host-staged transport: 2 ranks x 4 seeds, launches of 2 steps, buckets of 2
T = problem.n_timesteps
x_local = qs_all[rank * 4 : rank * 4 + 4].reshape(4 * T, -1).contiguous()
loose = Constraints(max_allowed_position_error_cm=1.0, max_allowed_rotation_error_deg=2.0, max_allowed_mjac_deg=400.0, max_allowed_mjac_cm=100.0)
problem.bind_obstacles()
r = D.ShardedRefiner(problem.robot, x_local, problem.target_path, 3, transport=D.HostStagedAllGather(), batch=2, bucket=2, n_streams=2, constraints=loose)
r.run_region(5)
r.synchronize()
path, idx = r.gather_and_search(0)
out["selected"] = [r.selected[b].cpu().numpy().tolist() for b in range(2)]
out["search_idx"] = idx.cpu().numpy().tolist()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_two_ranks_on_one_gpu_plan_like_one_process(tmp_path):
    from cppflow_amd import _hip
    from cppflow_amd.data_type_utils import problem_from_filename
    from cppflow_amd.data_types import Constraints, PlannerSettings
    from cppflow_amd.planners import LmIkSeedProvider, PlannerSearcher

    problem = problem_from_filename(None, "panda__line", robot=None, device=DEV)
    rb, T = problem.robot, problem.n_timesteps
    k = 16
    assert (k // 2 * T) % 4 == 0  # (the packed buffer's alignment rule: no padding needed in this case)
    qs = LmIkSeedProvider(seed=5)(problem, k).contiguous()
    np.savez(tmp_path / "qs.npz", qs=qs.cpu().numpy())
    # the single-process answers over the same candidates
    want = {}
    for lm_steps in (0, 2):
        s = PlannerSearcher(PlannerSettings(k=k, tmax_sec=30.0, anytime_mode_enabled=False, verbosity=0), rb, lambda p, kk: qs.clone(), candidate_lm_steps=lm_steps)
        want[lm_steps] = s.generate_plan(problem).plan.q_path.cpu().numpy()
    loose = Constraints(max_allowed_position_error_cm=1.0, max_allowed_rotation_error_deg=2.0, max_allowed_mjac_deg=400.0, max_allowed_mjac_cm=100.0)
    problem.bind_obstacles()
    x8 = qs[:8].reshape(8 * T, -1).contiguous()
    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * 8 * T, dtype=torch.uint8, device=DEV)
    summ = torch.empty((8, 8), dtype=torch.float32, device=DEV)
    one = rb.lm_pose_steps(x8, problem.target_path, 1e-6, 3.5, 0.35, n_steps=3, packed_out=packed, summary_out=summ, shape=_hip.SHAPE_ROW)
    want_sel = rb.select_valid_seed(summ, loose).cpu().numpy().tolist()
    _, want_idx, _ = rb.dp_search(one["x"].view(8, T, -1), one["ext_cost"].view(8, T))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT, port=port, npz=str(tmp_path / "qs.npz")))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    for r in res:  # every rank: the single process's plan, bit for bit, with and without LM iterations on the candidates
        assert np.array_equal(np.array(r["path0"], dtype=np.float32), want[0]) and np.array_equal(np.array(r["path2"], dtype=np.float32), want[2])
        assert all(row == want_sel for b in r["selected"] for row in b), (r["selected"], want_sel)
        assert r["search_idx"] == want_idx.cpu().numpy().tolist()
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)
