"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute through the HIP library)."""

import math
import os

import numpy as np
import pytest
import torch

from cppflow_amd import search
from cppflow_amd.collision_detection import get_only_non_colliding_qpaths
from cppflow_amd.data_type_utils import load_path_csv, offset_target_path, problem_from_filename, resample_path
from cppflow_amd.data_types import DEFAULT_CONSTRAINTS, Constraints, PlannerSettings, Problem
from cppflow_amd.distributed import PACKED_BYTES_PER_ROW, seed_shard, unpack_rows
from cppflow_amd.evaluation_utils import errors_are_below_threshold, seed_metrics_are_below_threshold
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, OptimizationParameters
from cppflow_amd.robots import Robot, get_robot

REFERENCE = "/root/reference/cppflow"


def test_pose_preset_values():
    """cppflow/lm_hyper_parameters.py:119-151."""
    p = ALT_LOSS_V2_1_POSE
    assert (p.lm_lambda, p.alpha_position, p.alpha_rotation) == (1e-6, 3.5, 0.35)
    assert p.use_pose and not p.use_differencing and not p.use_self_collisions and not p.use_env_collisions
    d = ALT_LOSS_V2_1_DIFF
    assert d.lm_lambda == 1e-6 and d.alpha_differencing == 0.00375 and d.n_virtual_configs == 4 and d.use_differencing
    with pytest.raises(AssertionError):
        OptimizationParameters(**{**p.__dict__, "use_self_collisions": True, "alpha_self_collision": 0.0})


def test_constraints_and_settings():
    c = Constraints(0.01, 0.1, 7.0, 2.0)  # scripts/evaluate.py:51-56
    assert c == DEFAULT_CONSTRAINTS and c.max_allowed_position_error_m == pytest.approx(1e-4)
    with pytest.raises(AssertionError):
        PlannerSettings(k=10, tmax_sec=1.0, anytime_mode_enabled=False, latent_distribution="cauchy")
    assert seed_metrics_are_below_threshold(c, [0.005, 0.05, 3.0, 0.0]) == (True, (True, True, True, True))
    assert seed_metrics_are_below_threshold(c, [0.02, 0.05, 8.0, 0.0])[1] == (False, True, False, True)
    ok, flags = errors_are_below_threshold(0.01, 0.1, 7.0, 2.0, torch.tensor([0.001]), torch.tensor([0.2]),
                                           torch.tensor([[1.0, -3.0]]), torch.zeros((1, 0)))  # fmt: skip
    assert not ok and flags == (True, False, True, True)


def test_robot_duck_type_surface():
    """Attributes the reference's hot path reads off jrl.Robot."""
    for name, d, npris in (("panda", 7, 0), ("fetch", 8, 1), ("fetch_arm", 7, 0), ("chain12", 12, 0)):
        rb = get_robot(name)
        assert isinstance(rb, Robot) and rb.ndof == d and rb.name == name
        assert len(rb.actuated_joints_limits) == d and all(l < u for l, u in rb.actuated_joints_limits)
        assert len(rb.prismatic_joint_idxs) == npris and rb.has_prismatic_joints == (npris > 0)
        assert sorted(rb.revolute_joint_idxs + rb.prismatic_joint_idxs) == list(range(d))
        x = torch.arange(2 * d, dtype=torch.float32).reshape(2, d)
        rev, pris = rb.split_configs_to_revolute_and_prismatic(x)
        assert rev.shape == (2, d - npris) and pris.shape == (2, npris)
        q = rb.sample_joint_angles(50)
        lo = np.array([l for l, _ in rb.actuated_joints_limits])
        hi = np.array([u for _, u in rb.actuated_joints_limits])
        assert q.shape == (50, d) and (q >= lo).all() and (q <= hi).all()
    with pytest.raises(ValueError):
        get_robot("pr2")


def test_capsule_link_map_is_ordered_like_the_distance_columns():
    """`robot._collision_capsules_by_link` (jrl attribute read at cppflow/collision_detection.py:137): one key per capsule, in
    the column order of env_collision_distances."""
    for name in ("panda", "fetch", "fetch_arm"):
        rb = get_robot(name)
        keys = list(rb._collision_capsules_by_link.keys())
        assert len(keys) == rb.n_capsules == len(rb.collision_capsule_names)
        assert [k.split("#")[0] for k in keys] == rb.collision_capsule_names
        assert all(v.shape == (7,) and float(v[6]) > 0 for v in rb._collision_capsules_by_link.values())


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors instead of computing anywhere else."""
    rb = get_robot("panda")
    x = torch.zeros((4, 7))
    for call in (lambda: rb.forward_kinematics(x), lambda: rb.jacobian(x), lambda: rb.self_collision_distances(x),
                 lambda: rb.clamp_to_joint_limits(x), lambda: rb.collision_masks(x.reshape(1, 4, 7)),
                 lambda: rb.lm_pose_steps(x, torch.zeros((4, 7)), 1e-6, 3.5, 0.35)):  # fmt: skip
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_padded_joint_limits_follow_search_py_arithmetic():
    """cppflow/search.py:46-51: fp32 limit tensors, in-place += / -= of the python-float paddings."""
    rb = get_robot("fetch")
    rb.set_joint_limit_padding(search.DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, search.DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    lo, hi = rb.padded_joint_limits()
    l_lim = torch.tensor([l for l, _ in rb.actuated_joints_limits], dtype=torch.float32)
    u_lim = torch.tensor([u for _, u in rb.actuated_joints_limits], dtype=torch.float32)
    l_lim[rb.prismatic_joint_idxs] += search.DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC
    l_lim[rb.revolute_joint_idxs] += search.DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE
    u_lim[rb.prismatic_joint_idxs] -= search.DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC
    u_lim[rb.revolute_joint_idxs] -= search.DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE
    assert np.array_equal(lo, l_lim.numpy()) and np.array_equal(hi, u_lim.numpy())
    assert search.DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE == pytest.approx(math.radians(1.5))
    assert (search.K_JLIM_COST, search.K_COLLISION_COST) == (100, 1000)


def test_problem_loader_formats():
    p = problem_from_filename(None, "panda__line", device="cpu")
    assert isinstance(p, Problem) and p.robot.name == "panda" and p.n_timesteps == 32 and p.name == "line"
    assert p.target_path.shape == (32, 7) and p.target_path.dtype == torch.float32
    np.testing.assert_allclose(p.target_path[0, :3].numpy(), [-0.1, 0.45, 0.5], atol=1e-6)
    # obstacle encoding of cppflow/data_type_utils.py:109-124, including element [3,3] left at 0
    assert len(p.obstacles_cuboids) == 1
    np.testing.assert_allclose(p.obstacles_cuboids[0].numpy(), [-0.06, -0.06, -0.06, 0.06, 0.06, 0.06], atol=1e-7)
    T = p.obstacles_Tcuboids[0].numpy()
    np.testing.assert_allclose(T[:3, 3], [0.35, 0.3, 0.45], atol=1e-7)
    assert np.array_equal(T[:3, :3], np.eye(3)) and T[3, 3] == 0.0
    # frame offset: torso_lift_link at q = 0 (cppflow/data_type_utils.py:65-73)
    f = problem_from_filename(None, "fetch__line", device="cpu")
    np.testing.assert_allclose(f.target_path[0, :3].numpy(), [-0.1 + 0.8 - 0.086875, 0.2, 0.25 + 0.37743], atol=1e-6)
    with pytest.raises(AssertionError):
        problem_from_filename(None, "panda__line.yaml")
    with pytest.raises(ValueError):
        bad = p.target_path.clone()
        bad[0, 3:] *= 1.5
        Problem(p.constraints, bad, None, p.robot, "x", "x")


def test_problem_listing():
    """ALL_PROBLEM_FILENAMES / get_problem_dict / get_all_problems (cppflow/data_type_utils.py:24-52, 222-241) over the
    shipped problems directory and, when it is mounted, over the reference's own (every one of its problem files parses)."""
    from cppflow_amd import data_type_utils as U

    assert U.ALL_PROBLEM_FILENAMES == ["fetch__line", "panda__line"] and set(U.ALL_OBS_PROBLEM_FILENAMES) <= set(U.ALL_PROBLEM_FILENAMES)
    problems = U.get_all_problems(device="cpu")
    assert [p.full_name for p in problems] == U.ALL_PROBLEM_FILENAMES and all(p.target_path.shape[1] == 7 for p in problems)
    ref = "/root/reference/cppflow"
    if os.path.isdir(os.path.join(ref, "problems")):
        theirs = U.get_all_problems(os.path.join(ref, "problems"), os.path.join(ref, "paths"), device="cpu")
        names = {p.full_name for p in theirs}
        assert {"panda__2cubes", "panda__1cube", "fetch__hello", "fetch_arm__s"} <= names
        by = {p.full_name: p for p in theirs}
        assert by["fetch__hello"].robot is by["fetch__circle"].robot or len(by["fetch__circle"].obstacles) > 0
        assert len(by["panda__2cubes"].obstacles) == 2


def test_offset_and_resample():
    rb = get_robot("panda")
    path = np.array([[0, 0, 0, 1, 0, 0, 0], [0.1, 0, 0, 1, 0, 0, 0], [0.3, 0, 0, 0, 1, 0, 0.0]])
    Rz90 = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    out = offset_target_path(rb, path, "world", [1, 2, 3], Rz90)
    np.testing.assert_allclose(out[:, :3], path[:, :3] + [1, 2, 3])
    np.testing.assert_allclose(np.abs(out[0, 3:]), [math.sqrt(0.5), 0, 0, math.sqrt(0.5)], atol=1e-12)
    # a non-identity waypoint orientation pins the ORDER: R_waypoint * R_offset (klampt's so3.mul(R_i, R_offset), data_type_utils.py:79-82)
    from cppflow_amd.data_type_utils import _quat_to_matrix

    c, s_ = math.cos(math.pi / 4), math.sin(math.pi / 4)
    rx90 = np.array([[0.2, 0, 0, c, s_, 0, 0]])  # 90 deg about x
    Rx90 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=float)
    got = _quat_to_matrix(offset_target_path(rb, rx90, "world", [0, 0, 0], Rz90)[0, 3:7])
    np.testing.assert_allclose(got, Rx90 @ np.array(Rz90, dtype=float), atol=1e-12)
    assert not np.allclose(got, np.array(Rz90, dtype=float) @ Rx90, atol=1e-3)
    r = resample_path(path, 7)
    assert r.shape == (7, 7)
    np.testing.assert_allclose(r[:, 0], np.linspace(0, 0.3, 7), atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(r[:, 3:], axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(r[0], path[0]) and np.testing.assert_allclose(r[-1], path[-1])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present (GPU box)")
def test_loader_reads_the_reference_problem_files():
    """The loader consumes the reference's own yaml / csv files where they lie (read as data)."""
    kw = dict(problems_dir=os.path.join(REFERENCE, "problems"), paths_dir=os.path.join(REFERENCE, "paths"), device="cpu")
    p = problem_from_filename(None, "panda__2cubes", **kw)
    assert p.n_timesteps == 200 and len(p.obstacles_cuboids) == 2
    np.testing.assert_allclose(p.target_path[0].numpy(), [0.45, 0.54, 0.79, 1, 0, 0, 0], atol=1e-6)
    np.testing.assert_allclose(p.obstacles_Tcuboids[1][:3, 3].numpy(), [-0.25, 0.3, 0.75], atol=1e-7)
    f = problem_from_filename(None, "fetch__hello", **kw)
    assert f.robot.ndof == 8 and f.n_timesteps == 553 and len(f.obstacles_cuboids) == 0
    a = problem_from_filename(None, "", filepath_override="/root/reference/tests/fetch_arm__s__truncated.yaml", **kw)
    assert a.robot.name == "fetch_arm" and a.n_timesteps == 59
    raw = load_path_csv(os.path.join(REFERENCE, "paths", "s_truncated.csv"))
    np.testing.assert_allclose(a.target_path[:, :3].numpy(), raw[:, :3] + [1.0 - 0.086875, 0.3, 0.55 + 0.37743], atol=1e-6)


def test_golden_reference_paths_fixture():
    """tests/golden/reference_paths.npz: the target paths of BASELINE configs C1-C4, generated by make_golden.py from the
    reference's data files with the loader above."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_paths.npz"))
    assert z["fetch_arm__s__truncated"].shape == (59, 7) and z["panda__1cube_first64"].shape == (64, 7)
    assert z["fetch__hello_first256"].shape == (256, 7) and z["panda__2cubes_resampled256"].shape == (256, 7)
    for k in z.files:
        if z[k].ndim == 2 and z[k].shape[1] == 7:
            np.testing.assert_allclose(np.linalg.norm(z[k][:, 3:], axis=1), 1.0, atol=1e-6)


def test_get_only_non_colliding_qpaths():
    qpaths = [torch.full((3, 2), float(i)) for i in range(4)]
    self_c = torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0], [0, 0, 0]], dtype=torch.bool)
    env_c = torch.tensor([[0, 0, 0], [0, 0, 0], [0, 0, 1], [0, 0, 0]], dtype=torch.bool)
    kept = get_only_non_colliding_qpaths(qpaths, self_c, env_c)
    assert [int(k[0, 0]) for k in kept] == [0, 3]


def _dp_search_reference_semantics(q, costs_ext):
    """Scalar restatement of the recurrence of cppflow/search.py:55-97 (dp_search_slow) for a cross-check."""
    k, T, d = q.shape
    costs = np.zeros((k, T))
    memo = np.zeros((k, T), dtype=int)
    costs[:, 0] = costs_ext[:, 0]
    for t in range(1, T):
        for ki in range(k):
            dq = np.abs(np.remainder(q[ki, t] - q[:, t - 1] + np.pi, 2 * np.pi) - np.pi).max(axis=1)
            c = np.maximum(dq, costs[:, t - 1]) + costs_ext[ki, t]
            memo[ki, t] = int(np.argmin(c))
            costs[ki, t] = c[memo[ki, t]]
    i = int(np.argmin(costs[:, -1]))
    path = np.zeros((T, d))
    for t in range(T - 1, -1, -1):
        path[t] = q[i, t]
        i = memo[i, t]
    return path


def test_dp_search_oracle_matches_slow_and_vectorised_recurrences():
    """The oracle's dp_search against a scalar restatement of dp_search_slow (cppflow/search.py:55-97) and the torch
    restatement of the vectorised recurrence (cppflow/search.py:145-173), with a prismatic joint (scaled by 5, :119-121)."""
    from cppflow_amd.robot_zoo import ROBOT_SPECS
    from oracle import ref_torch
    from tests import helpers as H

    rng = np.random.RandomState(0)
    k, T = 9, 14
    for name, d in (("panda", 7), ("fetch", 8)):
        q = rng.uniform(-1, 1, size=(k, T, d)).astype(np.float32)
        ext = ((rng.rand(k, T) < 0.2) * 1000 + (rng.rand(k, T) < 0.2) * 100).astype(np.float32)
        for o in (H.oracle64(name), H.oracle32(name)):
            idx, _ = o.dp_search(q, ext)
            path = q[idx, np.arange(T)]
            got_t = ref_torch.dp_search(ref_torch.TorchRobot(ROBOT_SPECS[name]()), torch.tensor(q), torch.tensor(ext)).numpy()
            np.testing.assert_allclose(path, got_t, atol=0)
            if name == "panda":
                np.testing.assert_allclose(path, _dp_search_reference_semantics(q.astype(np.float64), ext), atol=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        search.dp_search(get_robot("panda"), torch.zeros((2, 3, 7)), None, None, q_costs=torch.zeros((2, 3)))


def test_seed_shard_and_packed_layout():
    for S, world in ((1024, 8), (10, 4), (3, 8)):
        spans = [seed_shard(S, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == S
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    n = 12
    packed = torch.zeros(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8)
    cost, pe, re, sm, em, jm = unpack_rows(packed, n)
    cost += 1.5
    re += 2.5
    jm += 7
    assert packed[:4].view(torch.float32).item() == 1.5 and packed[8 * n : 8 * n + 4].view(torch.float32).item() == 2.5
    assert packed[14 * n].item() == 7 and pe.sum().item() == 0 and sm.sum().item() == 0 and em.sum().item() == 0
    assert Robot.PACKED_BYTES_PER_ROW == PACKED_BYTES_PER_ROW


def test_plan_container_mirrors_the_reference_properties():
    """`Plan` (cppflow/data_types.py:86-350): with a metrics row (as cppf_plan_metrics produces) every scalar property is
    read from it; without one the torch fall-backs give the same numbers; the verdict is the reference's conjunction."""
    from cppflow_amd.config import SUCCESS_THRESHOLD_initial_q_norm_dist
    from cppflow_amd.data_types import DEFAULT_CONSTRAINTS, Plan, PlanNp
    from cppflow_amd.evaluation_utils import angular_changes

    torch.manual_seed(0)
    T = 12
    limits = [(0.0, 0.4)] + [(-3.0, 3.0)] * 3
    q = torch.cat([torch.linspace(0.1, 0.12, T)[:, None], 0.3 + 0.01 * torch.randn(T, 3).cumsum(0)], dim=1)
    pos, rot = 1e-5 * torch.rand(T), 1e-4 * torch.rand(T)
    kw = dict(q_path=q, q_path_revolute=q[:, 1:], q_path_prismatic=q[:, :1], pose_path=torch.zeros(T, 7),
              target_path=torch.zeros(T, 7), robot_joint_limits=limits, self_colliding_per_ts=torch.zeros(T, dtype=torch.bool),
              env_colliding_per_ts=torch.zeros(T, dtype=torch.bool), positional_errors=pos, rotational_errors=rot,
              provided_initial_configuration=q[0:1] + 0.01, constraints=DEFAULT_CONSTRAINTS)  # fmt: skip
    host = Plan(**kw)
    row = torch.tensor([host.max_positional_error_cm, host.mean_positional_error_cm, host.max_rotational_error_deg,
                        host.mean_rotational_error_deg, host.mjac_deg, host.mjac_cm, host.path_length_rad, host.path_length_m,
                        0.0, 0.0, 0.0, host.initial_q_norm_dist, 0, 0, 0, 0])  # fmt: skip
    dev = Plan(**kw, metrics=row)
    for name in ("max_positional_error_cm", "max_positional_error_mm", "mean_positional_error_mm", "mean_rotational_error_deg",
                 "max_rotational_error_deg", "mjac_deg", "mjac_cm", "path_length_rad", "path_length_m", "initial_q_norm_dist"):  # fmt: skip
        assert getattr(dev, name) == pytest.approx(getattr(host, name), rel=1e-6), name
    assert host.path_length_rad == pytest.approx(float(angular_changes(q[:, 1:]).abs().sum()))
    assert host.max_positional_error_mm == pytest.approx(10 * host.max_positional_error_cm)
    assert host.is_valid and dev.is_valid and not host.joint_limits_violated
    assert abs(host.initial_q_norm_dist - 0.02) < 1e-6 < SUCCESS_THRESHOLD_initial_q_norm_dist
    verdict, text = host.is_valid_(verbose=True)
    assert verdict and "self_collisions: True" in text and "Plan {" in str(host) and "trajectory length" in str(host)
    # each term of the conjunction can veto (data_types.py:232-244)
    bad = dict(kw)
    bad["self_colliding_per_ts"] = torch.tensor([True] + [False] * (T - 1))
    assert not Plan(**bad).is_valid
    bad = dict(kw)
    bad["provided_initial_configuration"] = q[0:1] + 0.2
    assert not Plan(**bad).is_valid
    bad = dict(kw)
    bad["q_path"] = q.clone()
    bad["q_path"][3, 0] = 0.5  # above the prismatic limit
    assert Plan(**bad).joint_limits_violated and not Plan(**bad).is_valid
    row_bad = row.clone()
    row_bad[8] = 1.0
    assert Plan(**kw, metrics=row_bad).joint_limits_violated
    bad = dict(kw)
    bad["positional_errors"] = pos + 1.0
    assert not Plan(**bad).is_valid
    # no prismatic joints: lengths / mjac in cm are 0 (data_types.py:146-149, 199-210)
    rev_only = dict(kw, q_path=q[:, 1:], q_path_prismatic=q[:, :0], robot_joint_limits=limits[1:], provided_initial_configuration=None)
    p = Plan(**rev_only)
    assert p.mjac_cm == 0.0 and p.path_length_m == 0.0 and p.initial_q_norm_dist == 0.0 and p.mjac_per_timestep_cm.shape == (T - 1,)
    # numpy view and the results-table row (data_types.py:120-139, 352-366)
    view = PlanNp(host)
    assert isinstance(view.q_path, np.ndarray) and view.mjac_deg == host.mjac_deg
    with pytest.raises(AttributeError):
        view.no_such_field
    import pandas as pd
    import time as _time

    df = {"df": pd.DataFrame(columns=list("abcdefghijkl")), "t0": _time.time()}
    host.append_to_results_df(df)
    assert len(df["df"]) == 1 and bool(df["df"].iloc[0, 1]) is True and df["df"].iloc[0, 10] == pytest.approx(host.path_length_rad)


def test_pose_path_error_helpers():
    """positional_errors / rotational_errors on pose paths (cppflow/evaluation_utils.py:134-141)."""
    from cppflow_amd.evaluation_utils import positional_errors, rotational_errors

    a = torch.tensor([[0.0, 0, 0, 1, 0, 0, 0], [1.0, 2, 3, 1, 0, 0, 0]])
    half = math.sqrt(0.5)
    b = torch.tensor([[0.0, 3, 4, 1, 0, 0, 0], [1.0, 2, 3, half, half, 0, 0]])
    assert torch.allclose(positional_errors(a, b), torch.tensor([5.0, 0.0]))
    r = rotational_errors(a, b)
    assert abs(float(r[1]) - math.pi / 2) < 1e-5 and float(r[0]) < 1e-3
    assert abs(float(rotational_errors(a[1:], -b[1:])[0]) - math.pi / 2) < 1e-5  # q and -q are the same rotation


def test_target_path_length_known_answers():
    """tests/problem_test.py:27-60 of the reference: a stationary path has zero length; 0.1 m + 0 + 0.1 m of translation is
    20 cm; rotations of 5, 15 and 50 degrees about z add up to 5 + 10 + 35 = 50 degrees (4 places)."""
    def problem(path):
        return Problem(DEFAULT_CONSTRAINTS, torch.tensor(path, dtype=torch.float32), None, get_robot("panda"), "kat", "kat")

    still = problem([[0, 0, 0, 1.0, 0, 0, 0]] * 4)
    assert still.path_length_cumultive_positional_change_cm == pytest.approx(0.0, abs=1e-7)
    assert still.path_length_cumulative_rotational_change_deg == pytest.approx(0.0, abs=1e-4)
    moving = problem([[0, 0, 0, 1.0, 0, 0, 0], [0.1, 0, 0, 0.9990482, 0, 0, 0.0436194],
                      [0.1, 0, 0, 0.9914449, 0, 0, 0.1305262], [0.2, 0, 0, 0.9063078, 0, 0, 0.4226183]])  # fmt: skip
    assert moving.path_length_cumultive_positional_change_cm == pytest.approx(20.0, abs=1e-5)
    assert moving.path_length_cumulative_rotational_change_deg == pytest.approx(50.0, abs=5e-4)


def test_mjac_consistency_property_of_the_reference():
    """tests/evaluation_utils_test.py:12-15 of the reference: the three joint-change measures agree with each other."""
    import torch

    from cppflow_amd.evaluation_utils import angular_changes, calculate_mjac_deg, calculate_per_timestep_mjac_deg

    torch.manual_seed(0)
    for _ in range(5):
        qpath = 3.0 * torch.randn((10, 3))
        assert abs(calculate_mjac_deg(qpath) - float(calculate_per_timestep_mjac_deg(qpath).max())) < 1e-5
        assert abs(calculate_mjac_deg(qpath) - float(torch.rad2deg(angular_changes(qpath).abs().max()))) < 1e-5
        assert float(angular_changes(qpath).abs().max()) <= 3.14159275
