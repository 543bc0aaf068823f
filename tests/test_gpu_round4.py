"""Round-4 GPU tests (pytest -m gpu): several independent problems in one fused launch (cppf_lm_batch_*), the all-rows double-precision
mode inside a clamped launch, what the leading iterations of a fused K-step launch may and may not change, dp_search in one resident
launch beyond 256 candidates."""

import ctypes

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _robot(kind):
    """a shipped table, the generic kernels, or a run-time-specialised description"""
    from cppflow_amd.robots import Robot, get_robot

    if kind in ("panda", "fetch", "chain12"):
        return get_robot(kind), kind
    if kind == "panda_generic":
        rb = get_robot("panda")
        rb.debug_set("force_generic", 1)
        return rb, "panda"
    spec = H.random_chain_spec(7, seed=21)
    return Robot(spec, specialize=(kind == "random7_rtc")), spec


def _problem(chain, S, W, seed):
    rng = np.random.RandomState(seed)
    from oracle.oracle import Oracle

    o = Oracle(chain, f32=False)
    q_star = rng.uniform(chain.lo, chain.hi, size=(W, chain.ndof))
    target = H.f32(o.fk(H.f32(q_star)))
    x0 = np.clip(np.tile(q_star[None], (S, 1, 1)) + 0.1 * rng.randn(S, W, chain.ndof), chain.lo, chain.hi).reshape(S * W, chain.ndof)
    return H.f32(x0), target


@pytest.mark.parametrize("kind", ["panda", "fetch", "chain12", "panda_generic", "random7_rtc", "random7_generic"])
def test_batched_launch_equals_separate_launches_bit_for_bit(kind):
    """cppf_lm_batch_create / _launch: up to CPPF_MAX_BATCH independent problems (own x, target, S, W, outputs) laid end to end in ONE
    grid of the fused kernel.  Every workgroup belongs to exactly one problem, so each problem's outputs -- x, pose errors, the three
    masks, the cost, the per-seed summary (in the launch for W = 64 / 128 / 256, by the reduction launch behind it otherwise) -- must
    be bit for bit those of cppf_lm_pose_steps on it alone, whatever its neighbours in the grid are: ragged row counts (a last
    workgroup that is partly empty in the middle of the grid), W not a power of two, one-row problems, problems without a collision
    stage next to problems with one, on the shipped tables, the generic kernels and a run-time-specialised description."""
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robot_zoo import ROBOT_SPECS

    rb, spec = _robot(kind)
    chain = H.chain(spec) if isinstance(spec, str) else canonicalize(spec)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    shapes = [(3, 64), (1, 1), (2, 128), (5, 100), (1, 256), (7, 37), (4, 64), (1, 300), (2, 256), (16, 64), (1, 63), (9, 128)]
    K = 4
    try:
        for n_items in (1, 2, 5, len(shapes)):
            items, singles = [], []
            for i, (S, W) in enumerate(shapes[:n_items]):
                x0, target = _problem(chain, S, W, seed=100 * i + n_items)
                n = S * W
                x, tg = dev(x0), dev(target)
                want_coll = i % 4 != 3  # every fourth problem asks for the pose errors only
                it = dict(x=x, target=tg, x_out=torch.full_like(x, float("nan")))
                if want_coll:
                    it["packed_out"] = torch.full((rb.PACKED_BYTES_PER_ROW * n,), 0xAB, dtype=torch.uint8, device=DEV)
                    it["summary_out"] = torch.full((S, 8), float("nan"), dtype=torch.float32, device=DEV)
                    ref = rb.lm_pose_steps(x, tg, n_steps=K, packed_out=torch.empty_like(it["packed_out"]),
                                           summary_out=torch.empty_like(it["summary_out"]), shape=_hip.SHAPE_ROW, **LM)
                else:
                    it["errors_out"] = (torch.full((n,), float("nan"), device=DEV), torch.full((n,), float("nan"), device=DEV))
                    ref = rb.lm_pose_steps(x, tg, n_steps=K, want_errors=True, shape=_hip.SHAPE_ROW, **LM)
                items.append(it)
                singles.append(ref)
            plan = rb.lm_batch_plan(items, n_steps=K, **LM)
            st = torch.cuda.Stream(device=DEV)
            st.wait_stream(torch.cuda.current_stream())
            plan.launch_on(st)
            st.synchronize()
            for i, (out, ref) in enumerate(zip(plan.outputs, singles)):
                for k, v in out.items():
                    r = ref["seed_summary"] if k == "seed_summary" else ref[k]
                    assert torch.equal(v, r, ), (kind, n_items, i, shapes[i], k)
            # a second launch on torch's current stream over the same buffers: the same bits again (nothing is accumulated)
            before = [{k: v.clone() for k, v in out.items()} for out in plan.outputs]
            plan.launch()
            torch.cuda.synchronize()
            for out, b in zip(plan.outputs, before):
                for k, v in out.items():
                    assert torch.equal(v, b[k])
            del plan
    finally:
        rb.set_obstacles([], [])
        if kind == "panda_generic":
            rb.debug_set("force_generic")


def test_batch_contract_violations_are_refused():
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    x0, target = H.lm_problem("panda", 2, 64, seed=1)
    x, tg = dev(x0), dev(target)
    ok = dict(x=x, target=tg, x_out=torch.empty_like(x))
    with pytest.raises(AssertionError):
        rb.lm_batch_plan([], n_steps=2, **LM)
    with pytest.raises(AssertionError):
        rb.lm_batch_plan([ok] * (_hip.MAX_BATCH + 1), n_steps=2, **LM)
    with pytest.raises(AssertionError):  # clamp = 0 is only defined for a single step, as for cppf_lm_pose_steps
        rb.lm_batch_plan([ok], n_steps=2, clamp=False, **LM)
    # the C ABI refuses the quad shape and the outputs a batch cannot produce
    arr = (_hip.LmBatchItem * 1)()
    arr[0].x_in, arr[0].target, arr[0].S, arr[0].W = x.data_ptr(), tg.data_ptr(), 2, 64
    arr[0].out.x_out = ok["x_out"].data_ptr()
    h = ctypes.c_void_p()
    prm = _hip.LmParams(1e-6, 3.5, 0.35, 2, 1, 0.0, 0.0, _hip.SHAPE_QUAD, _hip.SOLVER_AUTO)
    with pytest.raises(AssertionError, match="row shape"):
        _hip.check(_hip.lib().cppf_lm_batch_create(rb._handle(torch.device(DEV)), 1, arr, ctypes.byref(prm), ctypes.byref(h)))
    prm.shape = _hip.SHAPE_ROW
    J = torch.empty((128, 6, 7), device=DEV)
    arr[0].out.J_out = J.data_ptr()
    with pytest.raises(AssertionError, match="not available in a batched launch"):
        _hip.check(_hip.lib().cppf_lm_batch_create(rb._handle(torch.device(DEV)), 1, arr, ctypes.byref(prm), ctypes.byref(h)))
    arr[0].out.J_out = None
    arr[0].S = 0
    with pytest.raises(AssertionError):
        _hip.check(_hip.lib().cppf_lm_batch_create(rb._handle(torch.device(DEV)), 1, arr, ctypes.byref(prm), ctypes.byref(h)))
    assert not h.value


def test_batched_launch_is_hip_graph_capturable_and_sees_obstacle_updates():
    """what bench.py does with it for strong-scaling shards: the launch captured into a hipGraph and replayed; and the robot's
    obstacles are read at LAUNCH time, like cppf_lm_pose_steps reads them (a batch created before set_obstacles sees the new set)."""
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    items = []
    for i in range(4):
        x0, target = H.lm_problem("panda", 8, 64, seed=40 + i)
        x = dev(x0)
        items.append(dict(x=x, target=dev(target), x_out=torch.empty_like(x),
                          packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * x.shape[0], dtype=torch.uint8, device=DEV),
                          summary_out=torch.empty((8, 8), dtype=torch.float32, device=DEV)))
    plan = rb.lm_batch_plan(items, n_steps=3, **LM)
    plan.launch()
    torch.cuda.synchronize()
    free = [out["env_mask"].clone() for out in plan.outputs]
    assert all(int(m.sum()) == 0 for m in free)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    try:
        st = torch.cuda.Stream(device=DEV)
        st.wait_stream(torch.cuda.current_stream())
        plan.launch_on(st)
        st.synchronize()
        eager = [{k: v.clone() for k, v in out.items()} for out in plan.outputs]
        for it, e in zip(items, eager):
            from cppflow_amd import _hip

            ref = rb.lm_pose_steps(it["x"], it["target"], n_steps=3, want_collisions=True, shape=_hip.SHAPE_ROW, **LM)  # (the batch's kernel)
            assert torch.equal(ref["env_mask"], e["env_mask"])
        assert sum(int(e["env_mask"].sum()) for e in eager) > 0
        for out in plan.outputs:
            for v in out.values():
                v.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            plan.launch_on(st)
        with torch.cuda.stream(st):
            g.replay()
        st.synchronize()
        for out, e in zip(plan.outputs, eager):
            for k, v in out.items():
                assert torch.equal(v, e[k]), k
    finally:
        rb.set_obstacles([], [])


@pytest.mark.parametrize("name", ["panda", "fetch_arm"])
def test_f64_mode_re_solves_rows_that_leave_the_joint_limits(name):
    """inside a clamped launch the conditioning gate used to drop every flagged row whose fp32 step left the joint limits --
    in CPPF_SOLVER_F64 too, which promises the exactly solved step on EVERY row.  One clamped step (K = 1, clamp = 1) from starts on
    and near the joint limits, where most steps are cut by the clamp: CPPF_SOLVER_F64 must equal clamp(exactly solved step) in task
    space on every row, in both kernel shapes; the rows where the clamp binds are where the old code kept the fp32 step.
    CPPF_SOLVER_AUTO keeps its documented exception (a flagged row that leaves the limits keeps the fp32 step), identically in both
    shapes."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb, ch, o64 = get_robot(name), H.chain(name), H.oracle64(name)
    rng = np.random.RandomState(7)
    W, d = 64, ch.ndof
    q_star = rng.uniform(ch.lo, ch.hi, size=(W, d))
    target = H.f32(o64.fk(H.f32(q_star)))
    blocks = []
    for b in range(32):  # a random subset of the joints sits on a limit, the others anywhere: near-singular AND against the clamp
        q = rng.uniform(ch.lo, ch.hi, size=(W, d))
        on = rng.rand(W, d) < 0.4
        side = rng.rand(W, d) < 0.5
        q = np.where(on, np.where(side, ch.lo + 1e-4 * rng.rand(W, d), ch.hi - 1e-4 * rng.rand(W, d)), q)
        blocks.append(q)
    x0 = H.f32(np.clip(np.concatenate(blocks), ch.lo, ch.hi))
    S = len(blocks)
    tgt = H.stacked(target, S)
    x64, Js, es, fails = o64.lm_step(x0, tgt, solver=0, **LM)
    assert fails == 0
    want = o64.clamp(x64.copy())
    binds = (want != x64).any(axis=1)
    assert binds.mean() > 0.3, binds.mean()
    sv = np.linalg.svd(Js, compute_uv=False)
    smin, smax = sv[:, -1], sv[:, 0]
    e_norm = np.abs(es).reshape(len(x0), -1).max(axis=1)
    # the bar of tests/test_gpu_parity_allrows.py for the fp64 solve: 2e-5, widened where the fp32 JACOBIAN's last bits move the step
    bar = 2e-5 + 2.5e-7 * smax * e_norm / np.maximum(smin, 1e-3)
    got = {}
    for solver in (_hip.SOLVER_F64, _hip.SOLVER_AUTO, _hip.SOLVER_F32):
        for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
            got[(solver, shape)] = host(rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=True, solver=solver, shape=shape, **LM)["x"])
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        x = got[(_hip.SOLVER_F64, shape)]
        # a joint the exact step clamps is clamped here too, to the same limit, bit for bit (up to steps that land within rounding of it)
        same_side = (x == want) | (np.abs(x - want) < 1e-4)
        ts = np.abs(np.einsum("nij,nj->ni", Js, x - want)).max(axis=1)
        ok = same_side.all(axis=1)
        assert ok.mean() > 0.97, (name, shape, ok.mean())
        assert (ts[ok] <= bar[ok]).all(), (name, shape, np.max(ts[ok] / bar[ok]), int(np.argmax(ts[ok] / bar[ok])))
        assert (ts[ok & binds] <= bar[ok & binds]).all()
    # the gate is at work on these rows: the ungated fp32 solve is farther out than the bar on some of the rows where the clamp binds
    ts32 = np.abs(np.einsum("nij,nj->ni", Js, got[(_hip.SOLVER_F32, _hip.SHAPE_ROW)] - want)).max(axis=1)
    assert (ts32[binds] > bar[binds]).sum() >= 3, (name, int((ts32[binds] > bar[binds]).sum()))
    # AUTO: every row is the fp32 step or the double-precision one, and a flagged row that leaves the limits is the fp32 one
    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        xa, x32, xd = got[(_hip.SOLVER_AUTO, shape)], got[(_hip.SOLVER_F32, shape)], got[(_hip.SOLVER_F64, shape)]
        is32, is64 = (xa == x32).all(axis=1), (xa == xd).all(axis=1)
        assert (is32 | is64).all(), (name, shape, int((~(is32 | is64)).sum()))
        # a row the fp32 step takes outside the limits (some joint of the fp32 result sits exactly on a limit it was not on before)
        # keeps the fp32 step
        cut32 = ((x32 == ch.lo) | (x32 == ch.hi)).any(axis=1) & ~((x0 == ch.lo) | (x0 == ch.hi)).any(axis=1)
        assert is32[cut32].mean() > 0.999, (name, shape, is32[cut32].mean())


def test_leading_iterations_do_not_move_the_fixed_point():
    """cppf_lm_params.n_steps: the K - 1 leading iterations of a plain fused launch run a leaner instantiation of the iteration (no
    early-out tests, no J / e outputs; and a cheaper sine / cosine valid inside the joint limits -- CPPF_LEAD_SINCOS, kernels_fused.h: 5e-7 absolute), the LAST
    iteration -- the one that produces x_out -- the general one.  So, whichever build: (1) a K = 1 launch is
    canonical (bit for bit the early-out launch that freezes nothing); (2) after K = 10 steps the converged rows sit at the same pose
    error as an all-general launch (an early-out launch whose tolerances nothing meets) and as the fp64 oracle's own iteration, to
    1e-5 -- the metric's bar -- and to 2e-6 against the all-general launch; (3) masks and cost are evaluated at x_out in the canonical
    arithmetic either way: bit-exact against the fp32 oracle at the launch's own x."""
    from cppflow_amd.robots import get_robot

    for name in ("panda", "fetch", "chain12"):
        rb, o64, o32 = get_robot(name), H.oracle64(name), H.oracle32(name)
        S, W, K = 16, 64, 10
        x0, target = H.lm_problem(name, S, W, seed=17)
        tgt = H.stacked(target, S)
        one = rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, **LM)["x"]
        one_c = rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, tol_pos_m=1e-18, tol_rot_rad=1e-18, **LM)["x"]
        assert torch.equal(one, one_c), name
        plain = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, want_collisions=True, **LM)
        canon = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, tol_pos_m=1e-18, tol_rot_rad=1e-18, **LM)
        xo = o64.lm_steps(x0, tgt, K, solver=0, **LM)
        pe_o, re_o = o64.pose_metrics_exact(xo, tgt)
        pe_p, re_p, pe_c, re_c = host(plain["pos_err_m"]), host(plain["rot_err_rad"]), host(canon["pos_err_m"]), host(canon["rot_err_rad"])
        conv = (pe_o < 1e-5) & (re_o < 1e-3) & (pe_c < 1e-5) & (re_c < 1e-3)  # (the rotation metric is floored at 8.94e-4, data_types.py:408-411)
        assert conv.mean() > 0.6, (name, conv.mean())
        assert np.abs(pe_p - pe_c)[conv].max() < 2e-6 and np.abs(re_p - re_c)[conv].max() < 5e-6, (name, np.abs(pe_p - pe_c)[conv].max())
        assert np.abs(pe_p - pe_o)[conv].max() < 1e-5 and np.abs(re_p - np.maximum(re_o, 8.94427191e-4))[conv].max() < 1e-5
        # the launch's own outputs at its own x: exact pose errors within 1e-5, masks bit-exact
        x = host(plain["x"])
        pe_x, re_x = o64.pose_metrics_exact(x, tgt)
        assert np.abs(pe_p - pe_x).max() < 1e-5 and np.abs(re_p - np.maximum(re_x, 8.94427191e-4)).max() < 1e-5
        m = o32.masks(x, None, None, None, None)
        assert np.array_equal(plain["self_mask"].cpu().numpy(), m["self_mask"])


def test_rows_outside_the_joint_limits_and_their_wavefront_neighbours():
    """The inner iterations of a fused K-step launch evaluate sine / cosine with polynomials that are valid inside the joint limits
    (CPPF_LEAD_SINCOS, kernels_fused.h).  The launch's own input need not be: the reference steps from wherever the seed is and clamps
    afterwards (cppflow/optimization.py:258-259).  So (1) rows that start OUTSIDE their limits -- by a little, by several turns -- must
    end where the fp64 oracle's iteration ends (pose error within 1e-5 on the rows it converges), and (2) the rows that share a
    wavefront with them must come out bit for bit as in a launch where those neighbours are ordinary rows: the choice of arithmetic
    is made per iteration, never per wavefront."""
    from cppflow_amd.robots import get_robot

    for name in ("panda", "fetch"):
        rb, ch, o64 = get_robot(name), H.chain(name), H.oracle64(name)
        S, W, K = 4, 64, 10
        x0, target = H.lm_problem(name, S, W, seed=5)
        tgt = H.stacked(target, S)
        rng = np.random.RandomState(3)
        rev = np.array([j for j in range(ch.ndof) if ch.jtype[j] == 0])
        x_out_of = x0.copy()
        rows = rng.choice(S * W, size=24, replace=False)
        for n, r in enumerate(rows):
            j = rev[n % len(rev)]
            x_out_of[r, j] = (ch.hi[j] + 0.3) if n % 3 == 0 else ((ch.lo[j] - 0.7) if n % 3 == 1 else x0[r, j] + 2 * np.pi * (1 + n % 2))
        x_out_of = H.f32(x_out_of)
        a = rb.lm_pose_steps(dev(x_out_of), dev(target), n_steps=K, want_errors=True, **LM)
        b = rb.lm_pose_steps(dev(x0), dev(target), n_steps=K, want_errors=True, **LM)
        others = np.setdiff1d(np.arange(S * W), rows)
        assert np.array_equal(host(a["x"])[others], host(b["x"])[others]), name
        xo = o64.lm_steps(x_out_of, tgt, K, solver=0, **LM)
        pe_o, re_o = o64.pose_metrics_exact(xo, tgt)
        pe, re = host(a["pos_err_m"]), host(a["rot_err_rad"])
        conv = (pe_o < 1e-5) & (re_o < 1e-3)
        assert conv[rows].sum() >= 5, (name, conv[rows].sum())  # (rows that start turns away from their limits mostly do not get there in K steps)
        sel = rows[conv[rows]]
        assert np.abs(pe[sel] - pe_o[sel]).max() < 1e-5 and np.abs(re[sel] - np.maximum(re_o[sel], 8.94427191e-4)).max() < 1e-5, name
        lo, hi = H.f32(ch.lo), H.f32(ch.hi)
        assert (host(a["x"]) >= lo).all() and (host(a["x"]) <= hi).all()


@pytest.mark.parametrize("name,k,T", [("panda", 257, 40), ("panda", 300, 256), ("panda", 512, 33), ("panda", 513, 33), ("panda", 1024, 64),
                                      ("panda", 1000, 17), ("fetch", 300, 64), ("chain12", 300, 48), ("chain12", 1024, 12)])
def test_resident_dp_search_beyond_256_candidates(name, k, T):
    """cppf_dp_search in ONE resident launch up to k = 1024 (four destinations per workgroup, at most 256 workgroups, one or two
    sources per lane; dp_persistent4_kernel on 1 024 lanes, dp_resident_kernel on 512 for 513+ candidates of chains of 10+ joints and
    as the A/B): the reference's rerun searches 300 candidates (cppflow/planners.py:47, 253-258), eight
    ranks gather 1024.  Cost table, argmins (the whole memo table) and path bit for bit against one launch per waypoint and against
    the fp32 oracle restatement of cppflow/search.py:128-191, with ties, +inf columns and candidates that share configurations."""
    from cppflow_amd.robots import get_robot

    rb, ch = get_robot(name), H.chain(name)
    rng = np.random.RandomState(k * 7 + T)
    base = rng.uniform(ch.lo, ch.hi, size=(4, 1, rb.ndof)) + 0.3 * np.cumsum(rng.randn(4, T, rb.ndof) * 0.1, axis=1)
    q = H.f32(np.clip(base[rng.randint(0, 4, size=k)] + 0.02 * rng.randn(k, T, rb.ndof), ch.lo, ch.hi))
    q[k // 2] = q[k // 3]  # two identical candidates: ties everywhere, the smaller index must win
    ext = ((rng.rand(k, T) < 0.15) * 1000.0 + (rng.rand(k, T) < 0.1) * 100.0).astype(np.float32)
    if T > 4:
        ext[: k // 2, T // 2] = np.inf
    got = {}
    for method in ("resident", "launches"):
        for rep in range(2):  # repeated calls reuse nothing: every call re-arms its own cost table
            path, idx, costsT, memoT, ran = rb.dp_search(dev(q), dev(ext), method=method, return_memo=True, return_method=True)
        assert ran == method
        got[method] = (host(path), idx.cpu().numpy(), host(costsT), memoT.cpu().numpy())
    assert got["resident"][1][0] >= 0
    for i, (a, b) in enumerate(zip(got["resident"], got["launches"])):
        assert np.array_equal(a, b), ("resident vs per-waypoint launches", i)
    auto = rb.dp_search(dev(q), dev(ext), return_method=True)
    assert auto[-1] == "resident" and np.array_equal(auto[1].cpu().numpy(), got["launches"][1])
    # the other resident form of this size (CPPF_TUNE_DP_PERSISTENT = 2: dp_resident_kernel, 512 lanes x four destinations, instead of
    # the 1 024-lane form of dp_persistent4_kernel that is the default up to 9 joints)
    rb.debug_set("dp_persistent", 2)
    try:
        path2, idx2, costs2, memo2, ran2 = rb.dp_search(dev(q), dev(ext), method="resident", return_memo=True, return_method=True)
    finally:
        rb.debug_set("dp_persistent")
    assert ran2 == "resident"
    for i, (a, b) in enumerate(zip((host(path2), idx2.cpu().numpy(), host(costs2), memo2.cpu().numpy()), got["launches"])):
        assert np.array_equal(a, b), ("resident (512-lane form) vs per-waypoint launches", i)
    want_idx, want_costs = H.oracle32(name).dp_search(q, ext)
    assert np.array_equal(got["resident"][2].T, want_costs) and np.array_equal(got["resident"][1], want_idx)
    assert np.array_equal(got["resident"][0], q[want_idx, np.arange(T)])


@pytest.mark.parametrize("k", [48, 175, 300, 1024])
def test_resident_dp_search_timeout_is_reported_and_the_caller_falls_back(k):
    """the bounded waits of the resident dp_search had no test and no short-circuit.  With the spin budget shrunk to nothing
    (CPPF_TUNE_DP_SPIN_LOG2 = 0: a wait that does not find its word at the first read expires) the launch must (1) return -- promptly:
    once one wait has expired every later one gives up as soon as it sees the flag -- (2) report best_idx = -1 and a NaN path instead
    of a wrong one, and (3) cppflow_amd.search.dp_search must fall back to one launch per waypoint FOR THAT CALL and return the right
    path, leaving the handle's own switches as they were (another thread's searches on the same robot are not affected)."""
    import time

    from cppflow_amd import _hip
    from cppflow_amd import search as search_mod
    from cppflow_amd.robots import get_robot

    rb, ch = get_robot("panda"), H.chain("panda")
    T = 64
    rng = np.random.RandomState(k)
    q = H.f32(np.clip(rng.uniform(ch.lo, ch.hi, size=(1, 1, 7)) * 0.5 + 0.2 * np.cumsum(rng.randn(k, T, 7) * 0.1, axis=1), ch.lo, ch.hi))
    ext = ((rng.rand(k, T) < 0.15) * 1000.0).astype(np.float32)
    want_idx, _ = H.oracle32("panda").dp_search(q, ext)
    rb.debug_set("dp_spin_log2", 0)
    try:
        t0 = time.perf_counter()
        path, idx, _, ran = rb.dp_search(dev(q), dev(ext), method="resident", return_method=True)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0
        assert ran == "resident"
        if int(idx[0].item()) >= 0:  # (every word happened to be there at the first read: nothing expired, the result must be right)
            assert np.array_equal(idx.cpu().numpy(), want_idx)
        else:
            assert (idx.cpu().numpy() == -1).all() and torch.isnan(path).all()
        best = search_mod.dp_search(rb, dev(q), None, None, q_costs=dev(ext))
        assert np.array_equal(host(best), q[want_idx, np.arange(T)])
        got = ctypes.c_int(-7)
        h = rb._handle(torch.device(DEV))
        _hip.check(_hip.lib().cppf_debug_get(h, _hip.TUNE_KEYS["dp_persistent"], ctypes.byref(got)))
        assert got.value == 1  # the fall-back did not touch the handle
    finally:
        rb.debug_set("dp_spin_log2")
    path, idx, _ = rb.dp_search(dev(q), dev(ext), method="resident")
    assert np.array_equal(idx.cpu().numpy(), want_idx)
