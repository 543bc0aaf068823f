"""Round-3 GPU tests (pytest -m gpu): the proof behind the collision stage's 3-instruction reciprocal, the conditioning gate of
the damped solve, per-handle tuning switches, the coupled step against the banded oracle at production path lengths."""

import ctypes
import threading

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


def test_fast_reciprocal_is_correctly_rounded():
    """csrc/lmik_device.h: rcp_rn(x) = v_rcp_f32 + one Newton step, DEFINED as 0 below 2^-100.  The CPU oracle spells the same
    function `|x| >= 2^-100 ? 1 / x : 0` (oracle/lmik_oracle.c), so the collision masks are bit-exact between the two only if the
    three instructions ARE the correctly rounded reciprocal.  All 2^32 bit patterns, on the GPU itself, against hipcc's IEEE
    division: no mismatch for any x with |x| < 2^126 (biased exponent <= 252: zero, denormals -- both sides 0 -- and every
    normal number up to there); beyond, where 1 / x is denormal, the two may differ (nothing geometric lives there)."""
    import os

    from cppflow_amd import _hip, build as hip_build

    _hip.lib()  # (torch's HIP runtime first, as for the product library)
    # the sweep kernel is a TEST translation unit (tests/native/rcp_sweep.hip, built by __graft_entry__.build()): it includes the
    # product's lmik_device.h, so the function under test is the product's rcp_rn, but no test kernel ships in libcppflow_hip.so
    assert os.path.exists(hip_build.TEST_OUT), "tests/native/libcppf_testkernels.so is missing: run __graft_entry__.build()"
    tk = ctypes.CDLL(hip_build.TEST_OUT)
    tk.cppf_test_rcp_sweep.restype = ctypes.c_int
    tk.cppf_test_rcp_sweep.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    mism = torch.zeros(256, dtype=torch.int64, device=DEV)
    chunk = 1 << 30
    for first in range(0, 1 << 32, chunk):
        assert tk.cppf_test_rcp_sweep(0, first, chunk, mism.data_ptr(), None) == 0
    torch.cuda.synchronize()
    m = mism.cpu().numpy()
    assert m[:253].sum() == 0, {int(e): int(v) for e, v in enumerate(m) if v}
    print("rcp_rn == RN(1/x) on every fp32 bit pattern with biased exponent <= 252; mismatches beyond:",
          {int(e): int(v) for e, v in enumerate(m) if v})


def test_tuning_switches_belong_to_one_handle_and_hold_across_threads():
    """include/cppflow_hip_debug.h: cppf_debug_set acts on ONE robot handle (no global mutable state besides the
    communicator).  Two Robot objects of the same description, driven from two threads with different settings, each keep their
    own: the one forced onto the generic kernels reports specialization >= 0 but launches DynRobot code (bit-identical results,
    so the observable is the switch itself), and a third handle created meanwhile sees the defaults."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import Robot
    from cppflow_amd.robot_zoo import ROBOT_SPECS

    a, b = Robot(ROBOT_SPECS["panda"]()), Robot(ROBOT_SPECS["panda"]())
    x0, target = H.lm_problem("panda", 4, 64, seed=5)
    ref = a.lm_pose_steps(dev(x0), dev(target), n_steps=3, want_errors=True, want_collisions=True, shape=_hip.SHAPE_ROW, **LM)
    errors = []

    def work(rb, generic, quad_rows, out):
        try:
            torch.cuda.set_device(0)
            rb.debug_set("force_generic", generic)
            rb.debug_set("quad_max_rows", quad_rows)
            for _ in range(20):
                r = rb.lm_pose_steps(dev(x0), dev(target), n_steps=3, want_errors=True, want_collisions=True, **LM)
                got = ctypes.c_int(-7)
                h = rb._handle(torch.device(DEV))
                _hip.check(_hip.lib().cppf_debug_get(h, _hip.TUNE_KEYS["force_generic"], ctypes.byref(got)))
                assert got.value == generic
                _hip.check(_hip.lib().cppf_debug_get(h, _hip.TUNE_KEYS["quad_max_rows"], ctypes.byref(got)))
                assert got.value == quad_rows
            torch.cuda.synchronize()
            out.append(r)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ra, rb_ = [], []
    ta = threading.Thread(target=work, args=(a, 1, 0, ra))
    tb = threading.Thread(target=work, args=(b, 0, 1 << 20, rb_))
    ta.start(), tb.start()
    ta.join(), tb.join()
    assert not errors, errors
    # a: CPPF_SHAPE_AUTO with quad_max_rows = 0 is the row shape, on the generic kernels: bit-identical to the specialised row
    # kernels (ref).  b: AUTO with quad_max_rows = 2^20 is the quad shape: the same x to fp32 rounding of the iteration
    for k in ("self_mask", "env_mask", "jlim_mask"):
        assert torch.equal(ra[0][k], ref[k])
    assert torch.equal(ra[0]["x"], ref["x"])
    assert ((rb_[0]["x"] - ref["x"]).abs().amax(dim=1) < 1e-3).float().mean() > 0.9  # (a few chaotic rows part ways in 3 steps)
    c = Robot(ROBOT_SPECS["panda"]())
    got = ctypes.c_int(-7)
    _hip.check(_hip.lib().cppf_debug_get(c._handle(torch.device(DEV)), _hip.TUNE_KEYS["force_generic"], ctypes.byref(got)))
    assert got.value == 0
    with pytest.raises(AssertionError):
        _hip.check(_hip.lib().cppf_debug_set(c._handle(torch.device(DEV)), 99, 1))
    a.debug_set("force_generic"), a.debug_set("quad_max_rows")


@pytest.mark.parametrize("name", ["panda", "fetch_arm"])
def test_gate_flags_what_it_should_and_every_mode_lands_where_it_says(name):
    """The three solver modes on one step of 4096 random rows of a 7-DoF arm (7-11 % of them near-singular):
    CPPF_SOLVER_F64 is the exactly solved step of the fp32 Jacobian; CPPF_SOLVER_AUTO equals the pure fp32 solve bit for bit on
    the rows the gate leaves alone and the double-precision one on the rows it flags (so every row is one or the other), flags
    between 3 % and 25 % of these rows, and tightening / loosening cppf_lm_params.solver_gate moves that fraction the right way."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    x0, target = H.lm_problem(name, 64, 64, seed=3)

    def step(solver, gate=0.0, shape=_hip.SHAPE_ROW):
        n, d = x0.shape
        x = dev(x0)
        out = _hip.LmOutputs()
        xo = torch.empty_like(x)
        out.x_out = xo.data_ptr()
        prm = _hip.LmParams(1e-6, 3.5, 0.35, 1, 0, 0.0, 0.0, shape, solver, gate)
        _hip.check(_hip.lib().cppf_lm_pose_steps(rb._handle(torch.device(DEV)), x.data_ptr(), dev(target).data_ptr(), 64, 64,
                                                 ctypes.byref(prm), ctypes.byref(out), None))
        torch.cuda.synchronize()
        return host(xo)

    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        x32, x64, xa = step(_hip.SOLVER_F32, shape=shape), step(_hip.SOLVER_F64, shape=shape), step(_hip.SOLVER_AUTO, shape=shape)
        is32, is64 = (xa == x32).all(axis=1), (xa == x64).all(axis=1)
        assert (is32 | is64).all(), (name, shape, int((~(is32 | is64)).sum()))
        flagged = ~is32
        assert 0.03 < flagged.mean() < 0.25, (name, shape, flagged.mean())
        tight, loose = step(_hip.SOLVER_AUTO, 1e-6, shape), step(_hip.SOLVER_AUTO, 1e-3, shape)
        f_tight, f_loose = (~(tight == x32).all(axis=1)).mean(), (~(loose == x32).all(axis=1)).mean()
        assert f_loose < flagged.mean() < f_tight, (name, shape, f_loose, flagged.mean(), f_tight)
    with pytest.raises(AssertionError, match="solver_gate"):
        step(_hip.SOLVER_AUTO, -1.0)


@pytest.mark.parametrize("name", ["panda", "fetch_arm"])
def test_a_row_does_not_depend_on_which_rows_share_its_wavefront(name):
    """The conditioning gate's double-precision re-solve is done by the whole wavefront (tasks dealt to the ACTIVE lanes, one slot per
    flagged row, further rounds when more rows are flagged than there are slots): a row's result must not depend on how many lanes
    are active, on which slot it got or on how many rounds its wavefront took.  K = 3 clamped steps on random rows (about a tenth of
    them near-singular), the same rows launched as 4096, as every prefix length that leaves a partly empty last wavefront, and in
    reverse order: bit for bit the same per row, in the default (gated) mode and with every row in double precision."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    x0, target = H.lm_problem(name, 64, 64, seed=5)
    tg = np.tile(target, (64, 1))  # one target per row: the launch is one "seed" of n waypoints
    # three regimes: the default (round 4: the lean iterations re-solve a row only when its step error is also large RELATIVE to the
    # residual -- few rows), the absolute gate in every iteration (cppf_debug_set gate_rel_ppm = 0: a tenth of the rows, more than
    # eight in some wavefronts, i.e. several rounds), and every row in double precision
    for solver, ppm in ((_hip.SOLVER_AUTO, None), (_hip.SOLVER_AUTO, 0), (_hip.SOLVER_F64, None)):
        rb.debug_set("gate_rel_ppm", ppm)  # (None: the default)
        full = rb.lm_pose_steps(dev(x0), dev(tg), n_steps=3, want_errors=True, shape=_hip.SHAPE_ROW, solver=solver, **LM)
        xf, pf = full["x"].cpu(), full["pos_err_m"].cpu()
        flagged = (rb.lm_pose_steps(dev(x0), dev(tg), n_steps=3, shape=_hip.SHAPE_ROW, solver=_hip.SOLVER_F32, **LM)["x"].cpu() != xf).any(dim=1)
        if ppm == 0 or solver == _hip.SOLVER_F64:
            assert flagged.float().mean() > 0.01, (name, solver, ppm, flagged.float().mean())  # the gate is at work on these rows
        for n in (1, 2, 9, 63, 65, 127, 200, 1000, 4059, 4095):
            r = rb.lm_pose_steps(dev(x0[:n]), dev(tg[:n]), n_steps=3, want_errors=True, shape=_hip.SHAPE_ROW, solver=solver, **LM)
            assert torch.equal(r["x"].cpu(), xf[:n]) and torch.equal(r["pos_err_m"].cpu(), pf[:n]), (name, solver, n)
        rev = rb.lm_pose_steps(dev(x0[::-1].copy()), dev(tg[::-1].copy()), n_steps=3, shape=_hip.SHAPE_ROW, solver=solver, **LM)
        assert torch.equal(rev["x"].cpu().flip(0), xf), (name, solver, "reversed")
        # and a permutation that scatters the flagged rows over other wavefronts
        perm = np.random.RandomState(1).permutation(x0.shape[0])
        pr = rb.lm_pose_steps(dev(x0[perm]), dev(tg[perm]), n_steps=3, shape=_hip.SHAPE_ROW, solver=solver, **LM)
        assert torch.equal(pr["x"].cpu(), xf[torch.as_tensor(perm)]), (name, solver, "permuted")
    rb.debug_set("gate_rel_ppm")


def _coupled_case(name, S, T, seed):
    """S trajectories tracking one smooth path that passes through a colliding configuration (so that collision rows are active
    in the block system), perturbed per seed like the seeds of one planning problem."""
    o, ch = H.oracle64(name), H.chain(name)
    rng = np.random.RandomState(seed)
    lo, hi = H.box_corners([c for c, _ in H.PANDA_2CUBES], [T_ for _, T_ in H.PANDA_2CUBES])
    cand = H.random_configs(name, 2000, seed=11)
    m = o.masks(cand, lo, hi, None, None)
    hit = cand[np.flatnonzero((m["self_mask"] | m["env_mask"]) > 0)[0]]
    base = np.clip(hit[None, :] + np.cumsum(0.01 * rng.randn(T, ch.ndof), axis=0), ch.lo, ch.hi)
    x = H.f32(np.clip(base[None] + 0.003 * rng.randn(S, T, ch.ndof), ch.lo, ch.hi).reshape(S * T, ch.ndof))
    target = H.f32(o.fk(H.f32(base)) + np.concatenate([0.002 * rng.randn(T, 3), np.zeros((T, 4))], axis=1))
    return x, target, lo, hi


@pytest.mark.parametrize("name,T", [("panda", 256), ("panda", 300), ("panda", 512), ("fetch", 256)])
def test_coupled_step_meets_the_banded_oracle_at_production_path_lengths(name, T):
    """every elimination order of cppf_lm_full_step -- parallel cyclic reduction over the waypoints, the
    two-ended row-per-lane DPP kernels (the fastest and most intricate code of kernels_coupled.h, the default at 1024 x 256), one
    wavefront per trajectory -- against the fp64 banded oracle AT their operating sizes: T in {256, 300, 512}, S in {1, 8, 512,
    1024}.  The oracle is evaluated on a spread of 12 trajectories of each batch (trajectories are independent; all S are
    checked against each other across the orders).  Collision rows active; |dx| < 2e-4 + 2e-3 |step|."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robots import get_robot

    rb, o = get_robot(name), H.oracle64(name)
    d = rb.ndof
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    pm = OptimizationParameters(**{**ALT_LOSS_V2_1_DIFF.__dict__, "alpha_self_collision": 0.02, "alpha_env_collision": 0.02})
    pm.virtual_configs = torch.tensor([])
    try:
        for S in (1, 8, 512, 1024):
            x, target, lo, hi = _coupled_case(name, S, T, seed=1000 * T + S)
            pick = np.unique(np.linspace(0, S - 1, 12).astype(int))
            rows = (pick[:, None] * T + np.arange(T)[None]).reshape(-1)
            want = o.lm_full_step(x[rows], target, pm, len(pick), T, boxes_lo=lo, boxes_hi=hi, banded=True)
            step = np.abs(want - x[rows]).max()
            assert step > 1e-5
            got = {}
            for mode, (pcr, full_rows) in {"pcr": (1 << 30, 1), "rows": (0, 1), "wave": (0, 0)}.items():
                if mode == "pcr" and T > 512:
                    continue
                rb.debug_set("pcr_max_rows", pcr)
                rb.debug_set("full_rows", full_rows)
                got[mode] = host(rb.lm_full_step(dev(x), dev(target), pm))
            for mode, g in got.items():
                assert np.isfinite(g).all(), (mode, S, T)
                err = np.abs(g[rows] - want).max()
                assert err < 2e-4 + 2e-3 * step, (name, mode, S, T, err, step)
            full_step = np.abs(got["wave"] - x).max()
            for mode in got:
                assert np.abs(got[mode] - got["wave"]).max() < 1e-5 + 1e-3 * full_step, (name, mode, S, T)
            # the default dispatch (no switch set) is one of them
            rb.debug_set("pcr_max_rows"), rb.debug_set("full_rows")
            dflt = host(rb.lm_full_step(dev(x), dev(target), pm))
            assert min(np.abs(dflt - g).max() for g in got.values()) == 0.0
        # the pose block on (rank-deficient blocks: parity in task space, like the short-path test of tests/test_gpu_api.py)
        S = 8
        x, target, lo, hi = _coupled_case(name, S, T, seed=77 + T)
        pmp = OptimizationParameters(**{**pm.__dict__, "use_pose": True, "alpha_position": 1.1, "alpha_rotation": 1.0})
        pmp.virtual_configs = torch.tensor([])
        want = o.lm_full_step(x, target, pmp, S, T, boxes_lo=lo, boxes_hi=hi, banded=True)
        g = host(rb.lm_full_step(dev(x), dev(target), pmp))
        Js = o.lm_step(x, H.stacked(target, S), lm_lambda=pmp.lm_lambda, alpha_position=pmp.alpha_position, alpha_rotation=pmp.alpha_rotation)[1]
        ok = np.linalg.svd(Js, compute_uv=False)[:, -1] >= 2e-2
        assert ok.mean() > 0.5
        assert np.abs(np.einsum("nij,nj->ni", Js, g - want))[ok].max() < 2e-3
    finally:
        rb.debug_set("pcr_max_rows"), rb.debug_set("full_rows")
        rb.set_obstacles([], [])


@pytest.mark.parametrize("name", ["panda", "fetch", "fetch_arm"])
def test_gpu_forward_kinematics_equals_hand_derived_values(name):
    """tests/helpers.py:FK_PINS (poses worked out on paper from the public URDF constants) through cppf_forward_kinematics, and
    the Jacobian at those configurations against central differences of those same pinned kinematics (oracle fp64)."""
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    q, pose = H.fk_pin_arrays(name)
    got = host(rb.forward_kinematics(dev(q)))
    assert H.pose_close(got, pose, 2e-6, 2e-6), (got, pose)
    J = host(rb.jacobian(dev(q)))
    assert np.abs(J - H.oracle64(name).jacobian(H.f32(q))).max() < 1e-5
    if name == "panda":  # the one FK datum of the reference tree (tests/planners_test.py:299-309), at the reference's own atol
        got = host(rb.forward_kinematics(dev([H.REFERENCE_PANDA_Q0])))
        assert H.pose_close(got, np.array([H.REFERENCE_PANDA_POSE]), 1e-3, 1e-3), got
    # and the zero-pose Jacobian against the values derived on paper (tests/helpers.py:J_PINS)
    Jz = H.J_PINS[name]
    assert np.abs(host(rb.jacobian(dev(np.zeros((1, Jz.shape[1])))))[0] - Jz).max() < 2e-6


# ---- the "satisfied" row options of LmResidualFns.get_r_and_J ------------------------------------------------


def _opt_params(**kw):
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters

    d = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    d.update(kw)
    return OptimizationParameters(**d)


def test_reference_row_option_known_answers_through_get_r_and_J():
    """The reference's known answers for its row operations (tests/optimization_utils_test.py:405-456 pose scale-down; :122-219
    differencing scale-down; :222-308 the same with shift_invalid_to_threshold; :458-588 the filter), reproduced THROUGH
    LmResidualFns.get_r_and_J on the GPU robot: the configurations are built so that the raw residuals are the vectors those tests
    start from, and the Constraints so that get_r_and_J forms their thresholds (cppflow/optimization_utils.py:515-520, 562-567)."""
    from cppflow_amd.data_types import Constraints
    from cppflow_amd.optimization_utils import LmResidualFns
    from cppflow_amd.robots import get_robot

    fetch = get_robot("fetch")
    z = [0.0] * 7
    # --- pose scale-down: thresholds 0.125 m / 1e-8 rad, scale 0.3; rows 1-3 of the reference vector (row 4 is a random draw)
    poses = fetch.forward_kinematics(dev([[0.05] + z, [0.2] + z, [0.1] + z]))
    qs = dev([[0.15] + z, [0.05] + z, [0.1] + z])
    pm = _opt_params(use_pose=True, alpha_position=1.0, alpha_rotation=1.0, use_differencing=False, use_virtual_configs=False,
                     use_self_collisions=False, use_env_collisions=False, pose_do_scale_down_satisfied=True,
                     pose_ignore_satisfied_threshold_scale=1.0, pose_ignore_satisfied_scale_down=0.3)  # fmt: skip
    cons = Constraints(max_allowed_position_error_cm=12.5, max_allowed_rotation_error_deg=1e-8, max_allowed_mjac_deg=7.0, max_allowed_mjac_cm=2.0)
    J, r = LmResidualFns.get_r_and_J(pm, fetch, qs, poses, constraints=cons)
    want = torch.tensor([[0, 0, 0, 0, 0, -0.03], [0, 0, 0, 0, 0, 0.15], [0, 0, 0, 0, 0, 0]], dtype=torch.float32).reshape(18, 1)
    torch.testing.assert_close(r.pose.cpu(), want, atol=2e-6, rtol=0)
    Jfk = fetch.jacobian(qs).cpu()
    assert abs(float(J.pose[5, 0].cpu()) - 0.3 * float(Jfk[0, 5, 0])) < 1e-6 and abs(float(Jfk[0, 5, 0]) - 1.0) < 1e-6  # the scaled row of J
    assert abs(float(J.pose[11, 8].cpu()) - float(Jfk[1, 5, 0])) < 1e-6  # 0.15 m > threshold: untouched
    # --- differencing: the 3 x 8 joint changes of the reference vector as a 4-configuration Fetch path
    d0 = [0.5, 0.1, 1.6, 0.1, 0.1, 0.1, 0.1, 0.1]
    d1 = [-0.4, 1.7, -1.7, 0.1, 0.1, 0.1, 0.1, 0.1]
    d2 = [0.2, 0.01, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
    x = np.cumsum(np.array([[0.0] * 8, d0, d1, d2]), axis=0)
    # the reference's differencing residual is angular_changes(x) = x[t+1] - x[t] (wrapped), its rows ordered (t, joint)
    cons = Constraints(max_allowed_position_error_cm=0.01, max_allowed_rotation_error_deg=0.1,
                       max_allowed_mjac_deg=float(np.rad2deg(1.5)) + 1.0, max_allowed_mjac_cm=26.0)
    base = dict(use_pose=False, use_virtual_configs=False, use_self_collisions=False, use_env_collisions=False, alpha_differencing=1.0,
                alpha_differencing_prismatic_scaling=1.0, differencing_ignore_satisfied_margin_deg=1.0,
                differencing_ignore_satisfied_margin_cm=1.0)  # thresholds: 1.5 rad, 0.25 m
    raw = np.array(d0 + d1 + d2)
    full = np.zeros(24, dtype=bool)
    full[[0, 2, 8, 9, 10]] = True  # invalid_row_idxs_expected of the reference tests
    for shift in (False, True):
        pm = _opt_params(**base, differencing_do_scale_satisfied=True, differencing_scale_down_satisfied_scale=0.5,
                         differencing_scale_down_satisfied_shift_invalid_to_threshold=shift)
        J, r = LmResidualFns.get_r_and_J(pm, fetch, dev(x), fetch.forward_kinematics(dev(x)), constraints=cons)
        want = np.where(full, raw, raw / 2)
        if shift:
            thr = np.where(np.arange(24) % 8 == 0, 0.25, 1.5)
            want = np.where(full, raw - np.sign(raw) * thr, want)
        assert np.abs(host(r.differencing).reshape(-1) - want).max() < 2e-6, (shift, host(r.differencing).reshape(-1), want)
        assert np.array_equal(r.differencing_invalid_row_idxs.cpu().numpy(), full)
        Jd = host(J.differencing)
        assert np.allclose(np.abs(Jd).max(axis=1), np.where(full, 1.0, 0.5)) and np.allclose(np.abs(Jd).sum(axis=1), np.where(full, 2.0, 1.0))
    # the filter (filter_rows_from_r_J_differencing, shift_to_threshold = True): only the five rows beyond their threshold stay
    pm = _opt_params(**base, differencing_do_ignore_satisfied=True)
    J, r = LmResidualFns.get_r_and_J(pm, fetch, dev(x), fetch.forward_kinematics(dev(x)), constraints=cons)
    thr = np.where(np.arange(24) % 8 == 0, 0.25, 1.5)
    assert np.abs(host(r.differencing).reshape(-1) - (raw - np.sign(raw) * thr)[full]).max() < 2e-6
    assert J.differencing.shape == (5, 32)


@pytest.mark.parametrize("option", ["pose_scale", "diff_scale", "diff_scale_shift", "diff_filter", "all"])
@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_device_coupled_step_with_each_satisfied_option_equals_the_oracle_dense_step(name, option):
    """cppf_lm_full_step with each "satisfied" option on == the oracle's dense restatement of the reference's step with that
    option (oracle/lmik_oracle.c: full_rows applies the options in the reference's order), the banded oracle (same rows), and the
    reference's dense formulation evaluated on the mirror's get_r_and_J matrices (J^T J + lambda I, solve).  Thresholds are chosen
    inside the spread of the case's residuals so that rows fall on both sides; collision rows are active."""
    from cppflow_amd.data_types import Constraints
    from cppflow_amd.optimization_utils import LmResidualFns
    from cppflow_amd.robots import get_robot

    rb, o = get_robot(name), H.oracle64(name)
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
    S, T = 3, 20
    x, target, lo, hi = _coupled_case(name, S, T, seed=31)
    rng = np.random.RandomState(3)
    x = H.f32(x + 0.02 * rng.randn(*x.shape))  # joint changes of ~0.03 rad between waypoints, pose errors of centimetres
    kw = dict(alpha_self_collision=0.02, alpha_env_collision=0.02, use_pose=True, alpha_position=1.1, alpha_rotation=1.0)
    if option in ("pose_scale", "all"):
        kw.update(pose_do_scale_down_satisfied=True, pose_ignore_satisfied_threshold_scale=2.0, pose_ignore_satisfied_scale_down=0.25)
    if option in ("diff_scale", "diff_scale_shift", "all"):
        kw.update(differencing_do_scale_satisfied=True, differencing_scale_down_satisfied_scale=0.4,
                  differencing_scale_down_satisfied_shift_invalid_to_threshold=option != "diff_scale",
                  differencing_ignore_satisfied_margin_deg=0.5, differencing_ignore_satisfied_margin_cm=0.5)
    if option == "diff_filter":
        kw.update(differencing_do_ignore_satisfied=True, differencing_ignore_satisfied_margin_deg=0.5, differencing_ignore_satisfied_margin_cm=0.5)
    if option not in ("pose_scale", "all"):
        kw.update(use_pose=False)  # (without the pose block the blocks are well conditioned: parity in joint space)
    pm = _opt_params(**kw)
    pm.virtual_configs = torch.tensor([])
    # thresholds in the middle of the data: 3 cm / 0.04 "rad" (sic: the reference passes degrees) for the pose rows,
    # 1.5 deg / 1 cm per step for the joint changes
    cons = Constraints(max_allowed_position_error_cm=1.5, max_allowed_rotation_error_deg=0.02, max_allowed_mjac_deg=2.0, max_allowed_mjac_cm=1.5)
    got = host(rb.lm_full_step(dev(x), dev(target), pm, constraints=cons))
    want = o.lm_full_step(x, target, pm, S, T, boxes_lo=lo, boxes_hi=hi, constraints=cons)
    band = o.lm_full_step(x, target, pm, S, T, boxes_lo=lo, boxes_hi=hi, constraints=cons, banded=True)
    plain = o.lm_full_step(x, target, _opt_params(**{k: v for k, v in kw.items() if "satisfied" not in k}), S, T, boxes_lo=lo, boxes_hi=hi)
    step = np.abs(want - x).max()
    assert np.abs(band - want).max() < 1e-7 * max(1.0, step / 1e-3)
    assert np.abs(plain - want).max() > 10 * (2e-4 + 2e-3 * step) or pm.use_pose, "the option must matter in this case"
    # the reference's own formulation on the mirror's dense matrices, trajectory 0 (fp64 on the GPU robot's fp32 kinematics)
    x0 = dev(x[:T])
    Tc, cub = [torch.tensor(T_) for _, T_ in obs], [torch.tensor(c) for c, _ in obs]
    pm.virtual_configs = x0  # what the loop sets (optimization.py:253) and what the device step takes for virtual_configs = NULL
    Jm, rm = LmResidualFns.get_r_and_J(pm, rb, x0, dev(target), Tcuboids=Tc, cuboids=cub, constraints=cons)
    Jd, rd = Jm.get_J().double().cpu().numpy(), rm.get_r().double().cpu().numpy()
    dense = x[:T] + np.linalg.solve(Jd.T @ Jd + pm.lm_lambda * np.eye(Jd.shape[1]), Jd.T @ rd).reshape(T, -1)
    if pm.use_pose:  # rank-deficient blocks: task space (see tests/test_gpu_api.py)
        Js = o.lm_step(x, H.stacked(target, S), lm_lambda=pm.lm_lambda, alpha_position=pm.alpha_position, alpha_rotation=pm.alpha_rotation)[1]
        ok = np.linalg.svd(Js, compute_uv=False)[:, -1] >= 2e-2
        assert ok.mean() > 0.5
        assert np.abs(np.einsum("nij,nj->ni", Js, got - want))[ok].max() < 2e-3
        assert np.abs(np.einsum("nij,nj->ni", Js[:T], dense - want[:T]))[ok[:T]].max() < 2e-3
        kept = rm.pose_invalid_row_idxs.float().mean().item()  # pose rows left at full weight: the case must have both kinds
        assert 0.1 < kept < 0.9, kept
        assert np.abs(plain - want).max() > 1e-3, "the option must matter in this case"
    else:
        assert np.abs(got - want).max() < 2e-4 + 2e-3 * step, (np.abs(got - want).max(), step)
        assert np.abs(dense - want[:T]).max() < 2e-4 + 2e-3 * step
    rb.set_obstacles([], [])


def test_finite_difference_jacobian_of_the_stacked_residual_agrees_with_the_analytic_one():
    """get_jacobian_finite_differencing (cppflow/optimization_utils.py:771-799, the reference's debugging aid): forward differences
    of the stacked residual of LmResidualFns.get_r_and_J against its analytic Jacobian (dr/dx = -J: r is  desired - current, J the
    Jacobian of `current`, which is why the step is x + solve(J^T J + lambda I, J^T r)), pose + differencing + virtual-config rows.  The differencing / virtual rows are linear (exact up to rounding / eps); the pose rows carry the O(eps) curvature term."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.optimization_utils import LmResidualFns, get_jacobian_finite_differencing

    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    rb.set_obstacles([], [])
    d = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    d.update(use_pose=True, alpha_position=3.5, alpha_rotation=0.35, use_self_collisions=False, use_env_collisions=False,
             alpha_differencing=1.0, n_virtual_configs=2)
    pm = OptimizationParameters(**d)
    rng = np.random.RandomState(5)
    T = 6
    x = H.f32(np.clip(H.random_configs("panda", 1, seed=3) + np.cumsum(0.03 * rng.randn(T, rb.ndof), axis=0), H.chain("panda").lo, H.chain("panda").hi))
    xd = dev(x)
    target = rb.forward_kinematics(dev(x + 0.002 * rng.randn(*x.shape)))
    pm.virtual_configs = xd.clone()
    jac, res = LmResidualFns.get_r_and_J(pm, rb, xd, target)
    J = jac.get_J().cpu().numpy()
    eps = 1e-3
    J_fd = get_jacobian_finite_differencing(rb, pm, xd, target, eps=eps).cpu().numpy()
    assert J_fd.shape == J.shape and res.get_r().shape[0] == J.shape[0]
    n_pose = 6 * T
    # linear rows: exact up to fp32 rounding of the difference quotient (residuals ~1, eps 1e-3 -> 1e-4)
    # (sign: the residual is  desired - current  and J the Jacobian of `current`, so dr/dx = -J throughout)
    assert np.abs(J_fd[n_pose:] + J[n_pose:]).max() < 5e-4
    # pose rows: curvature |d2r/dq2| * eps / 2 with lever arms <= ~1 m and alpha <= 3.5; the rotation rows use the geometric Jacobian
    # for d(roll, pitch, yaw of R_target R^T)/dq as the reference does, exact only at zero rotation error: + alpha_rot |J| |e_rot|
    e_rot = np.abs(res.pose.cpu().numpy().reshape(T, 6)[:, :3]).max() / 0.35
    assert 1e-4 < e_rot < 2e-2
    assert np.abs(J_fd[:n_pose] + J[:n_pose]).max() < 3.5 * 2.0 * eps + 0.35 * 2.0 * e_rot + 5e-4
    assert np.abs(J[:n_pose]).max() > 0.3  # (the comparison is not vacuous)


@pytest.mark.parametrize("specialize", [False, True])
def test_capsule_distance_known_answers_on_the_device(specialize):
    """tests/helpers.py:SEGMENT_KATS / BOX_KATS (closed-form segment-segment and segment-box distances, zero-length capsules =
    spheres included) through cppf_self_collision_distances / cppf_env_collision_distances and the masks of the fused launch, with
    the generic kernels and with the run-time-specialised ones."""
    from cppflow_amd.robots import Robot

    q0 = dev(np.zeros((64, 3)))
    for c0, c1, want in H.SEGMENT_KATS:
        rb = Robot(H.two_capsule_spec(c0, c1), specialize=specialize)
        got = host(rb.self_collision_distances(q0))
        assert got.shape == (64, 1) and np.abs(got - want).max() < 1e-6, (c0, c1, got[0], want)
        m = rb.collision_masks(q0.reshape(1, 64, 3), want_min_dists=True)
        assert np.abs(host(m["min_self"]) - want).max() < 1e-6 and not m["self_mask"].any()
    # distance gradients with spheres in play (sphere / sphere, sphere / segment, segment / sphere; off the joint axis so that the
    # distance moves): the device's analytic d(distance)/dq against the oracle's (which equals central differences to 2e-10)
    from oracle.oracle import Oracle

    p = (0.5, 0.25, 0.125)
    grad_cases = [(((1.5, 0, 0.5), (1.5, 0, 0.5)), (p, p)), (((1.5, -1, 0.5), (1.5, 1, 0.5)), (p, p)),
                  (((1.5, 0, 0.5), (1.5, 0, 0.5)), ((0.25, 0, 0), (0.75, 0.5, 0.25)))]
    qs = np.concatenate([np.zeros((1, 3)), H.f32(0.3 * np.random.RandomState(2).randn(63, 3))])
    for c0, c1 in grad_cases:
        rb = Robot(H.two_capsule_spec(c0, c1), specialize=specialize)
        o = Oracle(H.two_capsule_chain(c0, c1), f32=False)
        Jd, dist = rb.self_collision_distances_jacobian(dev(qs), return_distances=True)
        want_d, want_g = o.self_dists_grads(qs)
        assert np.isfinite(host(Jd)).all() and np.abs(host(dist) - want_d).max() < 1e-6
        assert np.abs(want_g).max() > 0.3 and np.abs(host(Jd) - want_g).max() < 1e-5, np.abs(host(Jd) - want_g).max()
    unit = H.cuboid_obstacle(0.5, 0.5, 0.5, 1.0, 1.0, 1.0)
    for c1, want in H.BOX_KATS:
        rb = Robot(H.two_capsule_spec(((0, 0, 5), (0, 0, 5)), c1), specialize=specialize)
        got = host(rb.env_collision_distances(q0, unit[0], unit[1]))
        assert got.shape == (64, 2) and np.abs(got[:, 1] - want).max() < 1e-6, (c1, got[0], want)
        rb.set_obstacles([unit[0]], [unit[1]])
        m = rb.collision_masks(q0.reshape(1, 64, 3), want_min_dists=True)
        assert np.abs(host(m["min_env"]) - want).max() < 1e-6, (c1, host(m["min_env"])[0], want)


def test_launches_past_four_gigabytes_of_rows_address_every_row():
    """Maximum sizes: 2^27 + 2^24 + 2^23 rows of a 7-joint robot are 4.46 GB of `x` -- byte offsets beyond 32 bits in every per-row array.
    The input is one 65 536-row block tiled, so every block of the result must equal the result of that block alone, bit for bit
    (fused launch with the collision stage and the per-seed summary, FK, and the standalone collision launch)."""
    from cppflow_amd.robots import get_robot

    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2**30:
        pytest.skip("needs 24 GB of free device memory")
    rb = get_robot("panda")
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    W, blk = 256, 65536
    reps = (2**27 + 2**24 + 2**23) // blk  # 2432 blocks = 159 383 552 rows, x = 4.46 GB
    n = reps * blk
    q_star = H.random_configs("panda", W, seed=4)
    target = dev(H.oracle64("panda").fk(q_star))
    ch = H.chain("panda")
    x_blk = dev(np.clip(np.tile(q_star, (blk // W, 1)) + 0.1 * np.random.RandomState(0).randn(blk, 7), ch.lo, ch.hi))
    small = rb.lm_pose_steps(x_blk, target, n_steps=2, want_errors=True, **LM,
                             packed_out=torch.empty(rb.PACKED_BYTES_PER_ROW * blk, dtype=torch.uint8, device=DEV),
                             summary_out=torch.empty((blk // W, 8), device=DEV))
    x_big = x_blk.repeat(reps, 1)
    assert x_big.shape == (n, 7) and x_big.numel() * 4 > 2**32
    pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
    sm = torch.empty((n // W, 8), device=DEV)
    big = rb.lm_pose_steps(x_big, target, n_steps=2, want_errors=True, **LM, packed_out=pk, summary_out=sm)
    torch.cuda.synchronize()
    xs, xb = small["x"], big["x"].view(reps, blk, 7)
    assert torch.equal(xb, xs[None].expand_as(xb))
    for k in ("pos_err_m", "rot_err_rad", "ext_cost", "self_mask", "env_mask", "jlim_mask"):
        assert torch.equal(big[k].view(reps, -1), small[k].view(1, -1).expand(reps, -1)), k
    assert torch.equal(sm.view(reps, -1), small["seed_summary"].view(1, -1).expand(reps, -1))
    del big, pk, sm
    poses = rb.forward_kinematics(x_big)
    assert torch.equal(poses.view(reps, blk, 7), rb.forward_kinematics(x_blk)[None].expand(reps, blk, 7))
    del poses
    m_big = rb.collision_masks(x_big.view(n // W, W, 7))
    m_small = rb.collision_masks(x_blk.view(blk // W, W, 7))
    for k in ("self_mask", "env_mask", "jlim_mask", "ext_cost"):
        assert torch.equal(m_big[k].reshape(reps, -1), m_small[k].reshape(1, -1).expand(reps, -1)), k
    rb.set_obstacles([], [])


@pytest.mark.parametrize("inputs", ["problem", "random"])
def test_fused_launch_is_deterministic_under_concurrency(inputs):
    """The same fused launch issued 96 times over four streams (gate-heavy random inputs included: the double-precision re-solves go
    through per-wavefront LDS slots) must write the same bits every time -- x, per-row outputs and the per-seed summary."""
    import bench
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    S, W, K = 256, 256, 10
    if inputs == "problem":
        x0, target, _ = bench.make_inputs_problem(rb, S, W, torch.device(DEV), seed=0)
    else:
        x0, target = bench.make_inputs(rb, S, W, torch.device(DEV), seed=0)
    n = S * W
    streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
    outs = []
    torch.cuda.synchronize()
    for i in range(96):
        pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
        sm = torch.empty((S, 8), dtype=torch.float32, device=DEV)
        xo = torch.empty_like(x0)
        with torch.cuda.stream(streams[i % 4]):
            rb.lm_pose_steps(x0, target, n_steps=K, x_out=xo, packed_out=pk, summary_out=sm, **LM)
        outs.append((xo, pk, sm))
    torch.cuda.synchronize()
    x_ref, pk_ref, sm_ref = outs[0]
    assert bool(torch.isfinite(x_ref).all())
    for xo, pk, sm in outs[1:]:
        assert torch.equal(xo, x_ref) and torch.equal(pk, pk_ref) and torch.equal(sm, sm_ref)
    rb.set_obstacles([], [])


def test_resident_dp_search_hand_offs_hold_under_uneven_load():
    """MI355X_MICROARCH.md: "test every hand-off under UNEVEN load".  The resident dp_search launch (workgroups handing their cost
    rows on through memory) repeated 40 times while two other streams keep the chip busy with full-size fused launches: every
    repetition must return the oracle's path, cost table and argmins bit for bit, and never report a hand-off time-out."""
    import bench
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    k, T = 175, 96
    rng = np.random.RandomState(4)
    ch = H.chain("panda")
    q = H.f32(np.clip(rng.uniform(ch.lo, ch.hi, size=(k, 1, 7)) + np.cumsum(0.05 * rng.randn(k, T, 7), axis=1), ch.lo, ch.hi))
    ext = H.f32(rng.choice([0.0, 100.0, 1000.0], size=(k, T), p=[0.8, 0.1, 0.1]))
    want_idx, want_costs = H.oracle32("panda").dp_search(q, ext)
    qd, ed = dev(q), dev(ext)
    S, W = 1024, 256
    x0, target, _ = bench.make_inputs_problem(rb, S, W, torch.device(DEV), seed=0)
    load_streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    bufs = [(torch.empty_like(x0), torch.empty(rb.PACKED_BYTES_PER_ROW * S * W, dtype=torch.uint8, device=DEV)) for _ in range(2)]
    dp_stream = torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    results = []
    for rep in range(40):
        for s_, (xo, pk) in zip(load_streams, bufs):  # uneven: the load comes and goes between repetitions
            if rep % 3 != 2:
                with torch.cuda.stream(s_):
                    rb.lm_pose_steps(x0, target, n_steps=10, x_out=xo, packed_out=pk, **LM)
        with torch.cuda.stream(dp_stream):
            results.append(rb.dp_search(qd, ed, method="resident"))
    torch.cuda.synchronize()
    for path, idx, costsT in results:
        gi = idx.cpu().numpy()
        assert (gi >= 0).all(), "hand-off time-out"
        assert np.array_equal(gi, want_idx)
        assert np.array_equal(costsT.cpu().numpy().T.astype(np.float64), want_costs)
        assert np.array_equal(path.cpu().numpy().astype(np.float64), q[want_idx, np.arange(T)])
    rb.set_obstacles([], [])


def test_kernels_stay_inside_their_output_buffers():
    """Ragged sizes (rows not a multiple of the workgroup, W not a multiple of the wavefront) with every output carved out of ONE
    sentinel-filled arena with 1 KB gaps: after the launches the gaps still hold the sentinel (no write past the end or in front of a
    buffer), for the fused launch in both kernel shapes, the standalone collision launch, FK / Jacobian, and dp_search."""
    from cppflow_amd import _hip
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    S, W, d = 5, 77, 7
    n = S * W
    x0, target = H.lm_problem("panda", S, W, seed=21)
    SENT = 0xA5
    arena = torch.full((64 << 20,), SENT, dtype=torch.uint8, device=DEV)
    cursor = [4096]
    spans = []

    def carve(nbytes, dtype, shape):
        start = (cursor[0] + 255) // 256 * 256
        t = arena[start : start + nbytes].view(dtype).view(shape)
        spans.append((start, start + nbytes))
        cursor[0] = start + nbytes + 1024
        return t

    def gaps_intact():
        torch.cuda.synchronize()
        mask = torch.ones(cursor[0] + 4096, dtype=torch.bool, device=DEV)
        for a, b in spans:
            mask[a:b] = False
        return bool((arena[: cursor[0] + 4096][mask] == SENT).all())

    for shape in (_hip.SHAPE_ROW, _hip.SHAPE_QUAD):
        xo = carve(n * d * 4, torch.float32, (n, d))
        pk = carve(rb.PACKED_BYTES_PER_ROW * n, torch.uint8, (rb.PACKED_BYTES_PER_ROW * n,))
        sm = carve(S * 8 * 4, torch.float32, (S, 8))
        rb.lm_pose_steps(dev(x0), dev(target), n_steps=3, x_out=xo, packed_out=pk, summary_out=sm, shape=shape, **LM)
        assert gaps_intact(), ("fused", shape)
        assert bool(torch.isfinite(xo).all()) and not bool((pk.view(torch.uint8)[-n:] == SENT).all())
    r = rb.lm_pose_steps(dev(x0), dev(target), n_steps=1, clamp=False, return_residual=True, want_iters=True,
                         x_out=carve(n * d * 4, torch.float32, (n, d)), **LM)
    assert gaps_intact() and r["J"].shape == (n, 6, d)
    q3 = dev(x0).reshape(S, W, d)
    rb.collision_masks(q3, want_min_dists=True)
    poses = rb.forward_kinematics(dev(x0))
    J = rb.jacobian(dev(x0))
    assert gaps_intact() and poses.shape == (n, 7) and J.shape == (n, 6, d)
    for k, T in ((5, 77), (130, 9), (300, 5)):
        qq = dev(H.random_configs("panda", k * T, seed=k).reshape(k, T, d))
        rb.dp_search(qq, torch.zeros((k, T), device=DEV))
        assert gaps_intact(), ("dp", k, T)
    rb.set_obstacles([], [])


class _SentinelArena:
    """Routes torch.empty / zeros / empty_like of CUDA tensors through one sentinel-filled arena with 1 KB gaps, so that a write outside
    ANY buffer the Python wrappers allocate for a kernel shows up as a damaged gap."""

    SENT = 0xA5

    def __init__(self, mbytes=256, fill=None):
        self.SENT = self.SENT if fill is None else fill
        self.arena = torch.full((mbytes << 20,), self.SENT, dtype=torch.uint8, device=DEV)
        self.cursor, self.spans = 4096, []
        self._orig = {}

    def alloc(self, shape, dtype):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
        start = (self.cursor + 255) // 256 * 256
        assert start + nbytes + 8192 < self.arena.numel(), "arena too small"
        self.spans.append((start, start + nbytes))
        self.cursor = start + nbytes + 1024
        return self.arena[start : start + nbytes].view(dtype).view(shape)

    def __enter__(self):
        o_empty, o_zeros, o_like = torch.empty, torch.zeros, torch.empty_like
        self._orig = dict(empty=o_empty, zeros=o_zeros, empty_like=o_like)

        def on_cuda(kw):
            d = kw.get("device", None)
            return d is not None and "cuda" in str(d)

        def empty(*size, **kw):
            if on_cuda(kw) and not kw.get("pin_memory", False):
                shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
                return self.alloc(shape, kw.get("dtype", torch.float32))
            return o_empty(*size, **kw)

        def zeros(*size, **kw):
            if on_cuda(kw):
                shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
                t = self.alloc(shape, kw.get("dtype", torch.float32))
                t.zero_()
                return t
            return o_zeros(*size, **kw)

        def empty_like(t, **kw):
            if t.is_cuda and not kw:
                return self.alloc(t.shape, t.dtype)
            return o_like(t, **kw)

        torch.empty, torch.zeros, torch.empty_like = empty, zeros, empty_like
        return self

    def __exit__(self, *a):
        torch.empty, torch.zeros, torch.empty_like = self._orig["empty"], self._orig["zeros"], self._orig["empty_like"]

    def intact(self):
        torch.cuda.synchronize()
        end = self.cursor + 4096
        mask = torch.ones(end, dtype=torch.bool, device=DEV)
        for a, b in self.spans:
            mask[a:b] = False
        return bool((self.arena[:end][mask] == self.SENT).all())


@pytest.mark.parametrize("name", ["panda", "fetch", "chain12"])
def test_every_entry_point_stays_inside_the_buffers_its_wrapper_allocates(name):
    """Every device entry point of the Python mirror at ragged sizes, with the wrappers' own output allocations routed through a
    sentinel arena (`_SentinelArena`): FK, Jacobian, pose errors / metrics, clamp, distance matrices and their Jacobians, masks, the
    fused launch (default outputs, residuals, iteration counts, minimum distances), seed validity / summary / selection, plan
    metrics, the coupled step in its three elimination orders, mjacs and dp_search in its three forms."""
    from cppflow_amd.data_types import DEFAULT_CONSTRAINTS
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    d = rb.ndof
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    S, W = 3, 37
    n = S * W
    x0, target = H.lm_problem(name, S, W, seed=31)
    x, tg = dev(x0), dev(target)
    arena = _SentinelArena()
    checks = []
    with arena:
        def step(label, fn):
            fn()
            checks.append((label, arena.intact()))

        step("fk", lambda: rb.forward_kinematics(x))
        step("jacobian", lambda: rb.jacobian(x))
        step("pose_error_metrics", lambda: rb.pose_error_metrics(x, tg))
        step("self distances", lambda: rb.self_collision_distances(x))
        step("env distances", lambda: rb.env_collision_distances(x, obs[0][0], obs[0][1]))
        step("self distance jacobian", lambda: rb.self_collision_distances_jacobian(x, return_distances=True))
        step("env distance jacobian", lambda: rb.env_collision_distances_jacobian(x, obs[0][0], obs[0][1], return_distances=True))
        step("masks", lambda: rb.collision_masks(x.reshape(S, W, d), want_min_dists=True))
        step("fused default", lambda: rb.lm_pose_steps(x, tg, n_steps=3, want_errors=True, want_collisions=True, want_min_dists=True, **LM))
        step("fused residual", lambda: rb.lm_pose_steps(x, tg, n_steps=1, clamp=False, return_residual=True, want_iters=True, **LM))
        step("fused early-out", lambda: rb.lm_pose_steps(x, tg, n_steps=6, tol_pos_m=1e-4, tol_rot_rad=1e-3, want_iters=True, want_errors=True, **LM))
        step("seed validity", lambda: rb.seed_validity(x, tg))
        step("plan metrics", lambda: rb.plan_metrics(x, tg))
        step("mjacs", lambda: rb.mjacs(x.reshape(S, W, d)))
        for method in ("table", "resident"):
            step("dp " + method, lambda: rb.dp_search(x.reshape(S, W, d), torch.zeros((S, W), device=DEV), method=method))
        rb.debug_set("dp_persistent", 0)
        step("dp per waypoint", lambda: rb.dp_search(x.reshape(S, W, d), torch.zeros((S, W), device=DEV), method="resident"))
        rb.debug_set("dp_persistent", None)
        kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
        kw.update(n_virtual_configs=2)
        pm = OptimizationParameters(**kw)
        pm.virtual_configs = x.clone()
        for order, sets in (("default", {}), ("sequential", {"pcr_max_rows": 0}), ("per wave", {"pcr_max_rows": 0, "full_rows": 0})):
            for k_, v_ in sets.items():
                rb.debug_set(k_, v_)
            try:
                step("coupled " + order, lambda: rb.lm_full_step(x, tg, pm, virtual_configs=pm.virtual_configs))
            finally:
                for k_ in sets:
                    rb.debug_set(k_, None)
        sm = torch.empty((S, 8), dtype=torch.float32, device=DEV)
        pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
        step("fused packed + summary", lambda: rb.lm_pose_steps(x, tg, n_steps=2, packed_out=pk, summary_out=sm, **LM))
        step("select", lambda: rb.select_valid_seed(sm, DEFAULT_CONSTRAINTS))
    bad = [label for label, ok in checks if not ok]
    assert not bad, bad
    assert len(arena.spans) > 40
    rb.set_obstacles([], [])


@pytest.mark.parametrize("name", ["panda", "fetch"])
def test_results_do_not_depend_on_what_the_output_and_work_buffers_held_before(name):
    """No entry point may read memory it has not written: the same calls with every wrapper-allocated buffer pre-filled with 0xA5 and
    with 0x00 return the same bits (work arrays of dp_search and of the coupled step included), and no input tensor is modified."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robots import get_robot

    rb = get_robot(name)
    d = rb.ndof
    obs = H.PANDA_2CUBES
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    S, W = 4, 53
    x0, target = H.lm_problem(name, S, W, seed=41)
    x, tg = dev(x0), dev(target)
    x_keep, tg_keep = x.clone(), tg.clone()
    kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    kw.update(n_virtual_configs=2)
    pm = OptimizationParameters(**kw)
    pm.virtual_configs = x.clone()
    ext = dev(np.random.RandomState(1).choice([0.0, 100.0, 1000.0], size=(S, W)))

    def run_all():
        out = {}
        r = rb.lm_pose_steps(x, tg, n_steps=4, want_errors=True, want_collisions=True, want_min_dists=True, want_iters=True, **LM)
        out.update({"lm_" + k: v for k, v in r.items()})
        r = rb.lm_pose_steps(x, tg, n_steps=1, clamp=False, return_residual=True, **LM)
        out.update({"res_" + k: v for k, v in r.items()})
        m = rb.collision_masks(x.reshape(S, W, d), want_min_dists=True)
        out.update({"mask_" + k: v for k, v in m.items()})
        out["seed_validity"] = rb.seed_validity(x, tg)
        out["plan_metrics"] = rb.plan_metrics(x, tg)
        out["coupled"] = rb.lm_full_step(x, tg, pm, virtual_configs=pm.virtual_configs)
        rb.debug_set("pcr_max_rows", 0)
        out["coupled_sequential"] = rb.lm_full_step(x, tg, pm, virtual_configs=pm.virtual_configs)
        rb.debug_set("pcr_max_rows", None)
        for method in ("table", "resident"):
            p_, i_, c_, m_ = rb.dp_search(x.reshape(S, W, d), ext, method=method, return_memo=True)
            out["dp_%s_path" % method], out["dp_%s_idx" % method], out["dp_%s_costs" % method] = p_, i_, c_
            out["dp_%s_memo" % method] = m_[1:]  # (row 0 of the memo table is defined as zeros, never an argmin)
        out["mjacs"] = rb.mjacs(x.reshape(S, W, d))
        out["self_dj"] = rb.self_collision_distances_jacobian(x)
        out["env_dj"] = rb.env_collision_distances_jacobian(x, obs[0][0], obs[0][1])
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in out.items()}

    results = []
    for fill in (0xA5, 0x00):
        arena = _SentinelArena(fill=fill)
        with arena:
            results.append(run_all())
        assert arena.intact()
    a, b = results
    assert a.keys() == b.keys()
    differing = [k for k in a if not torch.equal(a[k].view(torch.uint8) if a[k].dtype != torch.bool else a[k], b[k].view(torch.uint8) if b[k].dtype != torch.bool else b[k])]
    assert not differing, differing
    assert torch.equal(x, x_keep) and torch.equal(tg, tg_keep) and torch.equal(pm.virtual_configs, x_keep)
    rb.set_obstacles([], [])


@pytest.mark.parametrize("S,T", [(1, 256), (8, 256), (3, 300), (2, 512), (9, 64), (1, 2), (600, 64)])
def test_coupled_step_and_dp_search_workspaces_hold_at_boundary_shapes(S, T):
    """The sentinel arena again, at the shapes where the coupled step and dp_search switch kernels (parallel-in-time up to 256
    waypoints in LDS / beyond in the workspace, rows kernels beyond 512 trajectories; table / resident / per-waypoint dp_search)."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robots import get_robot

    rb = get_robot("panda")
    rb.set_obstacles([c for c, _ in H.PANDA_2CUBES], [T_ for _, T_ in H.PANDA_2CUBES])
    ch = H.chain("panda")
    rng = np.random.RandomState(S * 1000 + T)
    base = np.clip(H.random_configs("panda", 1, seed=T) + np.cumsum(0.02 * rng.randn(T, 7), axis=0), ch.lo, ch.hi)
    x = dev(np.clip(base[None] + 0.01 * rng.randn(S, T, 7), ch.lo, ch.hi).reshape(S * T, 7))
    tg = dev(H.oracle64("panda").fk(H.f32(base)))
    kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    nvc = 4 if T > 8 else 0
    kw.update(use_virtual_configs=bool(nvc), n_virtual_configs=nvc if nvc else None)
    pm = OptimizationParameters(**kw)
    pm.virtual_configs = x.clone() if nvc else torch.tensor([])
    arena = _SentinelArena(mbytes=512)
    bad = []
    with arena:
        for order, sets in (("default", {}), ("sequential", {"pcr_max_rows": 0}), ("per wave", {"pcr_max_rows": 0, "full_rows": 0}),
                            ("pcr in workspace", {"pcr_lds": 0})):
            for k_, v_ in sets.items():
                rb.debug_set(k_, v_)
            try:
                out = rb.lm_full_step(x, tg, pm, virtual_configs=pm.virtual_configs)
                assert bool(torch.isfinite(out).all()), order
            finally:
                for k_ in sets:
                    rb.debug_set(k_, None)
            if not arena.intact():
                bad.append("coupled " + order)
        q3 = x.reshape(S, T, 7)
        for method in ("table", "resident") if S <= 256 and T >= 2 else ("resident",):
            rb.dp_search(q3, torch.zeros((S, T), device=DEV), method=method)
            if not arena.intact():
                bad.append("dp " + method)
    assert not bad, bad
    rb.set_obstacles([], [])


@pytest.mark.parametrize("ndof,specialize", [(7, False), (7, True), (5, False), (12, False), (12, True)])
def test_generic_and_run_time_specialised_kernels_stay_inside_their_buffers(ndof, specialize):
    """The sentinel arena for robots that are NOT in the generated tables: the generic kernels (chain constants in kernel arguments,
    capsules staged in LDS) and the run-time-specialised ones, ragged sizes, prismatic joints included."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    spec = H.random_chain_spec(ndof, seed=5)
    rb = Robot(spec, specialize=specialize)
    ch = canonicalize(spec)
    o64 = Oracle(ch, f32=False)
    rng = np.random.RandomState(3)
    S, W = 3, 41
    q_star = H.f32(rng.uniform(ch.lo, ch.hi, size=(W, ndof)))
    tg = dev(o64.fk(q_star))
    x = dev(np.clip(q_star[None] + 0.05 * rng.randn(S, W, ndof), ch.lo, ch.hi).reshape(S * W, ndof))
    obs = [H.cuboid_obstacle(0.1, 0.1, 0.5, 0.3, 0.3, 0.3)]
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    kw.update(n_virtual_configs=2)
    pm = OptimizationParameters(**kw)
    pm.virtual_configs = x.clone()
    arena = _SentinelArena()
    bad = []
    with arena:
        calls = [
            ("fused", lambda: rb.lm_pose_steps(x, tg, n_steps=3, want_errors=True, want_collisions=True, want_min_dists=True, **LM)),
            ("fused residual", lambda: rb.lm_pose_steps(x, tg, n_steps=1, clamp=False, return_residual=True, **LM)),
            ("masks", lambda: rb.collision_masks(x.reshape(S, W, ndof), want_min_dists=True)),
            ("self distances", lambda: rb.self_collision_distances(x)),
            ("env distance jacobian", lambda: rb.env_collision_distances_jacobian(x, obs[0][0], obs[0][1], return_distances=True)),
            ("plan metrics", lambda: rb.plan_metrics(x, tg)),
            ("coupled", lambda: rb.lm_full_step(x, tg, pm, virtual_configs=pm.virtual_configs)),
            ("dp", lambda: rb.dp_search(x.reshape(S, W, ndof), torch.zeros((S, W), device=DEV))),
        ]
        if ndof >= 6:
            from cppflow_amd import _hip

            calls.append(("fused quad", lambda: rb.lm_pose_steps(x, tg, n_steps=3, want_errors=True, want_collisions=True, shape=_hip.SHAPE_QUAD, **LM)))
        for label, fn in calls:
            fn()
            if not arena.intact():
                bad.append(label)
    assert not bad, bad


@pytest.mark.parametrize("specialize", [False, True])
def test_robot_at_the_capsule_and_pair_maxima(specialize):
    """CPPF_MAX_CAPSULES (24) capsules and CPPF_MAX_PAIRS (128) checked pairs on an 8-joint chain with a prismatic joint: the
    collision stage, standalone and fused, stays bit-exact with the canonical-fp32 oracle in the generic kernels (capsules staged in
    LDS) and in the run-time-specialised ones (every pair test unrolled)."""
    from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec, canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    rng = np.random.RandomState(2)
    d = 8
    joints = [JointSpec(f"j{i}", f"l{i}", tuple(rng.uniform(-0.1, 0.2, 3)), tuple(rng.uniform(-1, 1, 3)), (0, 0, 1),
                        "revolute" if i != 3 else "prismatic", (-2.0, 2.0) if i != 3 else (-0.1, 0.3)) for i in range(d)]  # fmt: skip
    joints.append(JointSpec("tool", "tool", (0, 0, 0.1), (0, 0, 0), jtype="fixed"))
    caps = [CapsuleSpec(f"l{i}", tuple(rng.uniform(-0.05, 0.05, 3)), tuple(rng.uniform(-0.1, 0.1, 3)), 0.02 + 0.01 * r)
            for i in range(d) for r in range(3)]  # fmt: skip
    pairs = [(a, b) for a in range(24) for b in range(a + 1, 24) if b // 3 - a // 3 >= 2][:128]
    spec = RobotSpec("maxcaps", "24 capsules, 128 pairs", "base", joints, caps, collision_pairs=pairs)
    ch = canonicalize(spec)
    assert ch.n_capsules == 24 and ch.n_pairs == 128
    o32 = Oracle(ch, f32=True)
    S, W = 4, 100
    q = H.f32(rng.uniform(ch.lo, ch.hi, size=(S * W, d)))
    obs = [H.cuboid_obstacle(0.2, 0.1, 0.3, 0.3, 0.3, 0.3), H.cuboid_obstacle(-0.2, 0.1, 0.5, 0.2, 0.2, 0.2)]
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    rb = Robot(spec, specialize=specialize)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    jl_lo, jl_hi = rb.padded_joint_limits()
    want = o32.masks(q, lo, hi, jl_lo, jl_hi)
    assert want["self_mask"].sum() > 50 and want["env_mask"].sum() > 50
    keys = ("self_mask", "env_mask", "jlim_mask", "ext_cost", "min_self", "min_env")
    got = rb.collision_masks(dev(q).reshape(S, W, d), want_min_dists=True)
    for k in keys:
        assert np.array_equal(got[k].cpu().numpy().reshape(-1).astype(want[k].dtype), want[k]), k
    tg = dev(Oracle(ch, f32=False).fk(q[:W]))
    r = rb.lm_pose_steps(dev(q), tg, n_steps=3, want_errors=True, want_collisions=True, want_min_dists=True, **LM)
    w1 = o32.masks(host(r["x"]), lo, hi, jl_lo, jl_hi)
    for k in keys:
        assert np.array_equal(r[k].cpu().numpy().astype(w1[k].dtype), w1[k]), ("fused", k)


def test_generic_kernels_refuse_what_they_cannot_stage():
    """12 joints x 24 capsules: the generic fused kernels would need 144 KB of capsule staging beside the conditioning gate's slots --
    more than a compute unit's 160 KB of LDS.  The launch must be refused with a message (it used to take the process down), and
    the same robot specialised at run time runs and meets the oracle."""
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec, canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    rng = np.random.RandomState(5)
    d = 12
    joints = [JointSpec(f"j{i}", f"l{i}", tuple(rng.uniform(-0.1, 0.2, 3)), tuple(rng.uniform(-1, 1, 3)), (0, 0, 1), "revolute", (-2.0, 2.0))
              for i in range(d)]  # fmt: skip
    joints.append(JointSpec("tool", "tool", (0, 0, 0.1), (0, 0, 0), jtype="fixed"))
    caps = [CapsuleSpec(f"l{i}", tuple(rng.uniform(-0.05, 0.05, 3)), tuple(rng.uniform(-0.1, 0.1, 3)), 0.03) for i in range(d) for _ in range(2)]
    pairs = [(a, b) for a in range(24) for b in range(a + 1, 24) if b // 2 - a // 2 >= 2][:128]
    spec = RobotSpec("wide", "12 joints, 24 capsules", "base", joints, caps, collision_pairs=pairs)
    ch = canonicalize(spec)
    o32 = Oracle(ch, f32=True)
    q = H.f32(rng.uniform(ch.lo, ch.hi, size=(300, d)))
    tg = dev(Oracle(ch, f32=False).fk(q[:100]))
    rb = Robot(spec, specialize=False)
    with pytest.raises(RuntimeError, match="specialize"):
        rb.lm_pose_steps(dev(q), tg, n_steps=2, want_errors=True, want_collisions=True, shape=_hip.SHAPE_ROW, **LM)
    r0 = rb.lm_pose_steps(dev(q), tg, n_steps=2, want_errors=True, want_collisions=False, shape=_hip.SHAPE_ROW, **LM)  # nothing staged
    rs = Robot(spec, specialize=True)
    rs.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    r = rs.lm_pose_steps(dev(q), tg, n_steps=2, want_errors=True, want_collisions=True, shape=_hip.SHAPE_ROW, **LM)
    assert torch.equal(r["x"], r0["x"])
    jl_lo, jl_hi = rs.padded_joint_limits()
    want = o32.masks(host(r["x"]), None, None, jl_lo, jl_hi)
    for k in ("self_mask", "jlim_mask", "ext_cost"):
        assert np.array_equal(r[k].cpu().numpy().astype(want[k].dtype), want[k]), k


def test_coupled_step_at_the_capsule_pair_and_obstacle_maxima():
    """24 capsules, 128 pairs and 8 cuboids at once: the screening bit sets of the coupled step's block kernel wrap (more candidates
    than bits), which may cost work but never a collision row -- the step equals the oracle's dense restatement."""
    from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters
    from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec, canonicalize
    from cppflow_amd.robots import Robot
    from oracle.oracle import Oracle

    rng = np.random.RandomState(2)
    d = 8
    joints = [JointSpec(f"j{i}", f"l{i}", tuple(rng.uniform(-0.1, 0.2, 3)), tuple(rng.uniform(-1, 1, 3)), (0, 0, 1),
                        "revolute" if i != 3 else "prismatic", (-2.0, 2.0) if i != 3 else (-0.1, 0.3)) for i in range(d)]  # fmt: skip
    joints.append(JointSpec("tool", "tool", (0, 0, 0.1), (0, 0, 0), jtype="fixed"))
    caps = [CapsuleSpec(f"l{i}", tuple(rng.uniform(-0.05, 0.05, 3)), tuple(rng.uniform(-0.1, 0.1, 3)), 0.02 + 0.01 * r)
            for i in range(d) for r in range(3)]  # fmt: skip
    pairs = [(a, b) for a in range(24) for b in range(a + 1, 24) if b // 3 - a // 3 >= 2][:128]
    spec = RobotSpec("maxcaps", "24 capsules, 128 pairs", "base", joints, caps, collision_pairs=pairs)
    ch = canonicalize(spec)
    o64 = Oracle(ch, f32=False)
    obs = [H.cuboid_obstacle(*rng.uniform(-0.4, 0.4, 3), *rng.uniform(0.05, 0.25, 3)) for _ in range(8)]
    lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs])
    S, T = 3, 14
    base = np.clip(rng.uniform(ch.lo, ch.hi, size=(1, d)) * 0.5 + np.cumsum(0.03 * rng.randn(T, d), axis=0), ch.lo, ch.hi)
    x = H.f32(np.clip(base[None] + 0.01 * rng.randn(S, T, d), ch.lo, ch.hi).reshape(S * T, d))
    target = H.f32(o64.fk(H.f32(base)))
    kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
    kw.update(n_virtual_configs=2)
    pm = OptimizationParameters(**kw)
    xv = H.f32(x + 0.01 * rng.randn(*x.shape))
    pm.virtual_configs = dev(xv)
    want, r = o64.lm_full_step(x, target, pm, S, T, virtual_configs=xv, boxes_lo=lo, boxes_hi=hi, return_residual=True)
    n_fixed = (T - 1) * d + 4 * d
    assert r.shape[0] > n_fixed + 20, "the case must carry many active collision rows"
    step = np.abs(want - x).max()
    for specialize in (False, True):
        rb = Robot(spec, specialize=specialize)
        rb.set_obstacles([c for c, _ in obs], [T_ for _, T_ in obs])
        for sets in ({}, {"pcr_max_rows": 0}, {"pcr_max_rows": 0, "full_rows": 0}):
            for k_, v_ in sets.items():
                rb.debug_set(k_, v_)
            got = host(rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=pm.virtual_configs))
            assert np.abs(got - want).max() < 2e-4 + 2e-3 * step, (specialize, sets, np.abs(got - want).max(), step)
