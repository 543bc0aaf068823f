"""cppflow_amd.distributed on the GPU (pytest -m gpu): the per-rank engine `ShardedRefiner` against separate plain launches, the
one-call `sharded_candidate_evaluation` / `Planner` hook, and -- in a child process, the only way a one-GPU box can touch the real
transport -- a ONE-rank RCCL group through `pick_transport` (the C-ABI communicator) driving the same class."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)


def _panda_with_cubes():
    from cppflow_amd.robots import get_robot
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    rb = get_robot("panda")
    rb.set_obstacles([c for c, _ in H.PANDA_2CUBES], [T for _, T in H.PANDA_2CUBES])
    rb.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    return rb


@pytest.mark.parametrize("graphs", [False, True])
def test_refiner_ring_equals_separate_launches_and_selects_like_one_process(graphs):
    """32 seeds x 64 waypoints, launches of 4 steps, buckets of 8, two streams, a local (one-rank) exchange: every ring slot holds bit
    for bit what ONE plain cppf_lm_pose_steps launch writes, every step's selection equals `select_valid_seed` on that launch's
    summary, and `gather_and_search` equals `dp_search` on its outputs."""
    from cppflow_amd import distributed as D
    from cppflow_amd.data_types import Constraints

    rb = _panda_with_cubes()
    S, W, K = 32, 64, 4
    x0, target = H.collision_free_problem("panda", S, W, seed=5, obstacles=H.PANDA_2CUBES, bad_seeds=(0, 1, 2))
    x0, target = torch.tensor(x0, dtype=torch.float32, device=DEV), torch.tensor(target, dtype=torch.float32, device=DEV)
    loose = Constraints(max_allowed_position_error_cm=1.0, max_allowed_rotation_error_deg=2.0, max_allowed_mjac_deg=400.0, max_allowed_mjac_cm=100.0)
    r = D.ShardedRefiner(rb, x0, target, K, transport=D.LocalAllGather(), batch=4, bucket=8, n_streams=2, graphs=graphs, constraints=loose)
    assert (r.B, r.G, r.NBUF) == (4, 8, 16) and (r.graphs is not None) == graphs
    r.prewarm(0.0, count=4)
    r.run_region(22)  # 5 launches of 4 and one of 2: buckets 0, 1, 0 (partly filled, drained)
    r.synchronize()
    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=DEV)
    summ = torch.empty((S, 8), dtype=torch.float32, device=DEV)
    want = rb.lm_pose_steps(x0, target, n_steps=K, packed_out=packed, summary_out=summ, **LM)
    sel = rb.select_valid_seed(summ, loose)
    torch.cuda.synchronize()
    for b in list(range(16)):  # (every slot was written at least once by the pre-warm or the region)
        assert torch.equal(r.x_outs[b], want["x"]) and torch.equal(r.packeds[b], packed) and torch.equal(r.summ_all[b], summ), b
    for bucket in range(2):
        assert torch.equal(r.selected[bucket], sel.view(1, 4).expand(8, 4)), (bucket, r.selected[bucket][:2], sel)
    assert int(sel[0]) >= 3 and 0 < int(sel[1]) <= S - 3  # a non-vacuous selection: the first three seeds are still far from the path
    path, idx = r.gather_and_search(3)
    p2, i2, _ = rb.dp_search(want["x"].view(S, W, 7), want["ext_cost"].view(S, W))
    assert torch.equal(path, p2) and torch.equal(idx, i2)
    assert r.allgather_latency_us(5) > 0
    r.close()
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


def test_one_stream_refiner_paces_full_size_launches_and_changes_nothing():
    """`ShardedRefiner(pace=None)`: with ONE stream (launches in a dependency chain) the engine makes its launch plans with the
    fair-share pacing on (CPPF_TUNE_LM_PACE; csrc/kernels_fused.h: lm_pace), with several it does not; the robot's own switch is as
    it was afterwards, and a paced full-size engine writes bit for bit what an unpaced one writes."""
    import ctypes

    from cppflow_amd import _hip
    from cppflow_amd import distributed as D

    rb = _panda_with_cubes()
    S, W, K = 1024, 256, 10
    x0, target = H.lm_problem("panda", S, W, seed=9)
    x0, target = torch.tensor(x0, dtype=torch.float32, device=DEV), torch.tensor(target, dtype=torch.float32, device=DEV)

    def switch():
        got = ctypes.c_int(-7)
        _hip.check(_hip.lib().cppf_debug_get(rb._handle(torch.device(DEV)), _hip.TUNE_KEYS["lm_pace"], ctypes.byref(got)))
        return got.value

    outs = {}
    for label, kw in (("one stream", dict(n_streams=1)), ("one stream, unpaced", dict(n_streams=1, pace=False)), ("two streams", dict(n_streams=2))):
        r = D.ShardedRefiner(rb, x0, target, K, **kw)
        assert r.pace == (label == "one stream") and switch() == 0, label
        r.run_region(3)
        r.synchronize()
        outs[label] = (r.x_outs[0].clone(), r.packeds[0].clone(), r.summ_all[0].clone())
        r.close()
    for label in ("one stream, unpaced", "two streams"):
        for a, b in zip(outs["one stream"], outs[label]):
            assert torch.equal(a, b), label
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


def test_refiner_without_a_transport_alternates_streams_and_matches():
    from cppflow_amd import distributed as D

    rb = _panda_with_cubes()
    S, W, K = 16, 128, 3
    x0, target = H.lm_problem("panda", S, W, seed=6)
    x0, target = torch.tensor(x0, dtype=torch.float32, device=DEV), torch.tensor(target, dtype=torch.float32, device=DEV)
    r = D.ShardedRefiner(rb, x0, target, K, batch=1, n_streams=2)
    assert r.selected is None and r.NBUF == 4
    r.run_region(7)
    r.synchronize()
    from cppflow_amd import _hip

    # (the batch entry point is the row shape; a plain launch of 2 048 rows would pick the four-lanes-per-row shape by itself)
    want = rb.lm_pose_steps(x0, target, n_steps=K, want_errors=True, want_collisions=True, shape=_hip.SHAPE_ROW, **LM)
    torch.cuda.synchronize()
    for b in range(4):
        assert torch.equal(r.x_outs[b], want["x"])
        assert torch.equal(D.unpack_rows(r.packeds[b], S * W)[0], want["ext_cost"])
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


def test_sharded_candidate_evaluation_and_the_planner_hook():
    """One rank (no process group): the one-call form equals the direct launches, with and without LM iterations on the candidates, and
    a Planner with `candidate_lm_steps` plans through it."""
    from cppflow_amd import distributed as D
    from cppflow_amd.collision_detection import qpaths_batched_collisions
    from cppflow_amd.data_type_utils import problem_from_filename
    from cppflow_amd.data_types import PlannerSettings
    from cppflow_amd.planners import CppFlowPlanner, LmIkSeedProvider

    problem = problem_from_filename(None, "panda__line", robot=None, device=DEV)
    rb = problem.robot
    qs = LmIkSeedProvider(seed=3)(problem, 24)
    q_all, sm, em = D.sharded_candidate_evaluation(problem, qs, 0)
    sm2, em2 = qpaths_batched_collisions(problem, qs.contiguous())
    assert torch.equal(q_all, qs) and torch.equal(sm, sm2) and torch.equal(em, em2)
    q3, sm3, em3 = D.sharded_candidate_evaluation(problem, qs, 3)
    k, T, d = qs.shape
    r = rb.lm_pose_steps(qs.view(k * T, d), problem.target_path, n_steps=3, want_collisions=True, **LM)
    assert torch.equal(q3.view(k * T, d), r["x"]) and torch.equal(sm3.view(-1), r["self_mask"].view(torch.bool)) and torch.equal(em3.view(-1), r["env_mask"].view(torch.bool))
    planner = CppFlowPlanner(PlannerSettings(k=48, tmax_sec=30.0, anytime_mode_enabled=False, verbosity=0), rb, LmIkSeedProvider(seed=1), candidate_lm_steps=2)
    plan = planner.generate_plan(problem).plan
    assert plan.is_valid, str(plan)
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)


CHILD = r"""
import os, sys, json
sys.path.insert(0, {root!r})
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch, torch.distributed as dist
from cppflow_amd import distributed as D
from cppflow_amd.data_types import Constraints
from cppflow_amd.robots import get_robot
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str({port}), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
transport, rec = D.pick_transport(dev)
rb = get_robot("panda")
S, W, K = 128, 256, 5
z = np.load({npz!r})
x0, target = torch.tensor(z["x0"], dtype=torch.float32, device=dev), torch.tensor(z["target"], dtype=torch.float32, device=dev)
B, G, _, n_streams = D.launch_plan(S * W, 20)
loose = Constraints(max_allowed_position_error_cm=1.0, max_allowed_rotation_error_deg=2.0, max_allowed_mjac_deg=400.0, max_allowed_mjac_cm=100.0)
r = D.ShardedRefiner(rb, x0, target, K, transport=transport, batch=B, bucket=G, n_streams=n_streams, constraints=loose)
r.run_region(20)
r.synchronize()
summ = torch.empty((S, 8), dtype=torch.float32, device=dev)
pk = torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=dev)
want = rb.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=K, packed_out=pk, summary_out=summ)
sel = rb.select_valid_seed(summ, loose)
cal = r.calibrate_streams(20)
r.run_region(20)
r.synchronize()
path, idx = r.gather_and_search(0)
p2, i2, _ = rb.dp_search(want["x"].view(S, W, 7), want["ext_cost"].view(S, W))
out = dict(transport=rec["transport"], world_seen=rec["world_seen"], plan=[B, G, n_streams],
           selected_ok=bool(all(torch.equal(r.selected[b], sel.view(1, 4).expand(G, 4)) for b in range(n_streams))),
           x_ok=bool(torch.equal(r.x_outs[0], want["x"])), search_ok=bool(torch.equal(path, p2) and torch.equal(idx, i2)),
           latency_us=r.allgather_latency_us(50), candidates=cal["candidates"], n_valid=int(sel[1]))
transport.close()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_one_rank_rccl_group_drives_the_package_class(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    x0, target = H.collision_free_problem("panda", 128, 256, seed=0, bad_seeds=(0,))  # (no obstacles: self-collision-free waypoints)
    np.savez(tmp_path / "in.npz", x0=x0, target=target)
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT, port=port, npz=str(tmp_path / "in.npz")))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    import json

    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    d = json.loads(line[len("RESULT "):])
    assert d["world_seen"] == 1 and "C ABI" in d["transport"], d  # RCCL through cppf_comm_init_rank / cppf_allgather_bytes
    assert d["plan"] == [8, 8, 2] and d["selected_ok"] and d["x_ok"] and d["search_ok"], d
    assert d["latency_us"] > 0.5 and d["candidates"] == 30 and 0 < d["n_valid"] <= 127, d
