"""Round-5 additions on the GPU (pytest -m gpu): the resident dp_search's fit check and the C-ABI lifetime rule through the
Python wrapper (the C client exercises it from C, tests/c_client/abi_client.c)."""

import gc

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_dp_search_takes_the_per_waypoint_launches_when_the_resident_grid_cannot_fit():
    """cppf_dp_search holds the resident launch's grid against occupancy x compute units before launching (include/cppflow_hip.h): with
    the device pretended down to 8 compute units (cppf_debug_set CPPF_TUNE_CU_COUNT) CPPF_DP_AUTO must run one launch per waypoint --
    same result, no spinning -- and a forced CPPF_DP_RESIDENT must be refused with CPPF_ERR_UNSUPPORTED."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    k, T, d = 300, 24, 7
    g = torch.Generator().manual_seed(3)
    q = (torch.rand((k, T, d), generator=g) * 2 - 1).to(DEV).contiguous()
    cost = (torch.rand((k, T), generator=g) < 0.1).float().mul(1000.0).to(DEV).contiguous()
    want_path, want_idx, _ = rb.dp_search(q, cost, method="launches")
    res_path, res_idx, _ = rb.dp_search(q, cost, method="resident")
    assert torch.equal(res_idx, want_idx) and torch.equal(res_path, want_path)
    rb.debug_set("cu_count", 8)  # 75 workgroups of 1 024 lanes cannot be resident on 8 compute units
    try:
        rb.debug_set("dp_spin_log2", 0)  # (were the resident kernel launched all the same, its waits would expire at once: best_idx = -1)
        p, i, _ = rb.dp_search(q, cost, method="resident")  # CPPF_DP_AUTO underneath
        assert torch.equal(i, want_idx) and torch.equal(p, want_path)
        qT, cT, mT = torch.empty((T, k, d), device=DEV), torch.empty((T, k), device=DEV), torch.empty((T, k), dtype=torch.int32, device=DEV)
        bp, bi = torch.empty((T, d), device=DEV), torch.empty(T, dtype=torch.int32, device=DEV)
        rc = _hip.lib().cppf_dp_search(rb._handle(torch.device(DEV)), q.data_ptr(), cost.data_ptr(), k, T, 5.0, qT.data_ptr(), cT.data_ptr(),
                                       mT.data_ptr(), bp.data_ptr(), bi.data_ptr(), _hip.DP_RESIDENT, None)
        assert rc == _hip.CPPF_ERR_UNSUPPORTED and b"cannot hold the resident launch" in _hip.lib().cppf_last_error()
    finally:
        rb.debug_set("cu_count", None)
        rb.debug_set("dp_spin_log2", None)
    torch.cuda.synchronize()


def test_a_batch_outlives_its_robot_handle_without_a_fault():
    """`Robot.__del__` destroys the handle while an `LmBatchPlan` of it is still alive (whatever order the interpreter picks at
    shutdown): the batch's launch is then refused (AssertionError: CPPF_ERR_INVALID), nothing faults, and deleting the plan releases both."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    x0, target = H.lm_problem("panda", 4, 64, seed=1)
    x0, target = torch.tensor(x0, dtype=torch.float32, device=DEV), torch.tensor(target, dtype=torch.float32, device=DEV)
    xo = torch.empty_like(x0)
    plan = rb.lm_batch_plan([dict(x=x0, target=target, x_out=xo)], 1e-6, 3.5, 0.35, n_steps=3)
    plan.launch()
    torch.cuda.synchronize()
    want = xo.clone()
    handle = rb._handle(torch.device(DEV))
    plan._keep[0] = None  # the plan normally keeps its Robot alive: drop that reference, as interpreter shutdown may
    _hip.lib().cppf_robot_destroy(handle)  # what Robot.__del__ does
    rb._handles = {}
    del rb
    gc.collect()
    with pytest.raises(AssertionError):
        plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(xo, want)  # nothing was launched
    del plan  # cppf_lm_batch_destroy: releases the batch and the deferred robot
    gc.collect()
    rb2 = get_robot("panda")  # the library is intact
    assert torch.equal(rb2.lm_pose_steps(x0, target, 1e-6, 3.5, 0.35, n_steps=3, shape=_hip.SHAPE_ROW)["x"], want)


@pytest.mark.parametrize("name,S,W", [("panda", 1024, 256), ("fetch", 1024, 256), ("panda", 96, 64)])
def test_fair_share_pacing_changes_no_result(name, S, W):
    """CPPF_TUNE_LM_PACE (include/cppflow_hip_debug.h; csrc/kernels_fused.h: lm_pace): wavefront priorities by progress against a clock
    schedule, for a full-size launch in a dependency chain.  It is scheduling only: x, the packed per-row outputs and the per-seed
    summary of a full-size launch (the kind that is paced) and of a small one (never paced) are bit for bit those of the unpaced
    launch, through the plain entry point and through the batch entry point, with the built-in estimate and with explicit schedules."""
    import numpy as np

    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    x0, target = H.lm_problem(name, S, W, seed=5)
    x0d, td = torch.tensor(x0, dtype=torch.float32, device=DEV), torch.tensor(target, dtype=torch.float32, device=DEV)
    n = S * W
    LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)

    def run(batch):
        xo = torch.empty_like(x0d)
        pk = torch.zeros(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
        sm = torch.zeros((S, 8), dtype=torch.float32, device=DEV)
        if batch:
            plan = rb.lm_batch_plan([dict(x=x0d, target=td, x_out=xo, packed_out=pk, summary_out=sm)], n_steps=10, **LM)
            plan.launch()
        else:
            rb.lm_pose_steps(x0d, td, n_steps=10, x_out=xo, packed_out=pk, summary_out=sm, shape=_hip.SHAPE_ROW, **LM)
        torch.cuda.synchronize()
        return xo, pk, sm

    try:
        want = run(False)
        for pace in (-1, 150, 300, 2000):
            rb.debug_set("lm_pace", pace)
            for batch in (False, True):
                got = run(batch)
                for a, b, what in zip(got, want, ("x", "packed", "summary")):
                    assert torch.equal(a, b), (name, S, W, pace, batch, what)
    finally:
        rb.debug_set("lm_pace")
        rb.set_obstacles([], [])
        rb.set_joint_limit_padding(None, None)
