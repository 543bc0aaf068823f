"""The control flow of run_lm_alternating_loss (cppflow/optimization.py:147-373) with every device call replaced by a scripted fake:
which step is taken when, the TL-convergence rule, the termination rules and what is returned.  CPU only (no GPU, no library)."""
import types

import pytest
import torch

from cppflow_amd import optimization as opt
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE


class Script:
    """per iteration: (tl after the step, flags after the step (pos, rot), whether x_is_valid finds a valid trajectory)"""

    def __init__(self, steps):
        self.steps, self.calls, self.i = steps, [], 0

    def install(self, mp):
        def pose(problem, state, params, return_residual=False):
            self.calls.append("pose")
            return state.x + 1.0

        def full(problem, state, params, return_residual=False):
            assert torch.equal(params.virtual_configs, state.x), "virtual configs := current x before a differencing step (:253)"
            self.calls.append("diff")
            return state.x + 100.0

        def clamp(robot, x, verbosity=0):
            return x

        def evaluate(problem, target, x, parallel_count):
            m = torch.zeros((parallel_count, 16))
            m[:, 6] = self.steps[self.i][0] / parallel_count
            return m

        def valid(problem, constraints, target, x, parallel_count, verbosity=0, seed_metrics=None):
            _, flags, ok = self.steps[self.i]
            self.i += 1
            return (x if ok else None), 0, (flags[0], flags[1], True, True, False, False)

        mp.setattr(opt, "levenberg_marquardt_only_pose", pose)
        mp.setattr(opt, "levenberg_marquardt_full", full)
        mp.setattr(opt, "clamp_to_joint_limits", clamp)
        mp.setattr(opt, "evaluate_seeds", evaluate)
        mp.setattr(opt, "x_is_valid", valid)


def _problem(W=4, d=3):
    robot = types.SimpleNamespace(ndof=d)
    problem = types.SimpleNamespace(robot=robot, n_timesteps=W, target_path=torch.zeros((W, 7)), constraints=None)
    x0 = torch.zeros((W, d))
    return opt.OptimizationProblem(problem, None, x0.clone(), problem.target_path, 0, 1, None), opt.OptimizationState(x0.clone(), 0, 0.0)


def _run(mp, steps, **kw):
    s = Script(steps)
    s.install(mp)
    p, st = _problem()
    args = dict(return_residuals=False, tmax_sec=None, max_n_steps=len(steps), return_if_valid_after_n_steps=len(steps),
                convergence_threshold=0.3)  # fmt: skip
    args.update(kw)
    r = opt.run_lm_alternating_loss(p, st, ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, **args)
    return s, r


def test_leads_with_pose_steps_and_switches_to_differencing_when_both_pose_flags_hold(monkeypatch):
    """initial flags (pos valid, rot invalid) lead with a pose step (:218-219, :251-258); a differencing step is taken exactly when
    BOTH pose flags of the previous evaluation hold -- whether or not the trajectory as a whole was valid"""
    steps = [(10.0, (False, False), False), (10.0, (True, False), False), (10.0, (True, True), False), (9.0, (False, True), False),
             (9.0, (True, True), True), (8.0, (True, True), True)]
    s, r = _run(monkeypatch, steps)
    assert s.calls == ["pose", "pose", "pose", "diff", "pose", "diff"]
    assert r.is_valid and r.n_steps_taken == 5 and r.parallel_seed_idx == 0
    # the last valid x is returned: the state after the sixth step (three pose steps, one differencing, one pose, one differencing)
    assert torch.equal(r.x_opt, torch.full((4, 3), 4 * 1.0 + 2 * 100.0))


def test_tl_convergence_stops_at_once_if_the_previous_step_was_valid(monkeypatch):
    """:275-297 -- the TL change between two consecutive DIFFERENCING steps below the threshold marks convergence; if the step before
    this one ended valid the loop stops right there, BEFORE evaluating validity again, and returns that earlier trajectory"""
    steps = [(10.0, (True, True), False), (10.0, (True, True), True), (9.9, (True, True), True), (0.0, (True, True), True)]
    s, r = _run(monkeypatch, steps)
    assert s.calls == ["pose", "diff", "diff"] and s.i == 2  # x_is_valid ran twice only
    assert r.is_valid and r.n_steps_taken == 2
    assert torch.equal(r.x_opt, torch.full((4, 3), 1.0 + 100.0))  # the state after step 1, not after the converging step


def test_converged_but_not_valid_keeps_going_until_a_valid_trajectory_shows_up(monkeypatch):
    """converged with the previous step NOT valid: keep stepping; the first valid evaluation afterwards ends the loop (:339-343)"""
    steps = [(10.0, (True, True), False), (10.0, (True, True), False), (9.9, (True, True), False), (9.8, (False, True), False),
             (9.8, (True, True), True), (1.0, (True, True), True)]
    s, r = _run(monkeypatch, steps)
    assert s.calls == ["pose", "diff", "diff", "diff", "pose"]
    assert r.is_valid and r.n_steps_taken == 4
    assert torch.equal(r.x_opt, torch.full((4, 3), 2 * 1.0 + 3 * 100.0))


def test_a_tl_change_above_the_threshold_is_not_convergence(monkeypatch):
    steps = [(10.0, (True, True), True), (9.0, (True, True), True), (8.0, (True, True), True), (7.0, (True, True), True)]
    s, r = _run(monkeypatch, steps)
    assert s.calls == ["pose", "diff", "diff", "diff"] and r.n_steps_taken == 3 and r.is_valid


def test_return_if_valid_after_n_steps(monkeypatch):
    """:351-358 -- with a valid trajectory in hand the loop ends at the first i > return_if_valid_after_n_steps"""
    steps = [(10.0, (False, False), True)] + [(10.0 - k, (False, False), False) for k in range(1, 8)]
    s, r = _run(monkeypatch, steps, return_if_valid_after_n_steps=2)
    assert len(s.calls) == 4 and r.n_steps_taken == 3 and r.is_valid
    assert torch.equal(r.x_opt, torch.full((4, 3), 1.0))  # the one valid trajectory, found at step 0


def test_never_valid_returns_the_current_x_after_max_n_steps(monkeypatch):
    steps = [(10.0, (False, False), False)] * 5
    s, r = _run(monkeypatch, steps)
    assert s.calls == ["pose"] * 5 and not r.is_valid and r.n_steps_taken == 4
    assert torch.equal(r.x_opt, torch.full((4, 3), 5.0))


def test_time_limit_returns_the_last_valid_trajectory(monkeypatch):
    """:345-350 -- tmax_sec reached: the last valid trajectory if there is one"""
    steps = [(10.0, (False, False), True), (10.0, (False, False), False), (10.0, (False, False), False)]
    s, r = _run(monkeypatch, steps, tmax_sec=0.0, max_n_steps=None, return_if_valid_after_n_steps=None)
    assert s.calls == ["pose"] and r.is_valid and r.n_steps_taken == 0
    assert torch.equal(r.x_opt, torch.full((4, 3), 1.0))


def test_argument_contract(monkeypatch):
    """without a time limit both step limits are required and ordered (the reference's asserts, :171-176)"""
    p, st = _problem()
    with pytest.raises(AssertionError):
        opt.run_lm_alternating_loss(p, st, ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, False, None, 5, None, 0.3)
    with pytest.raises(AssertionError):
        opt.run_lm_alternating_loss(p, st, ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, False, None, 5, 6, 0.3)


def test_x_is_valid_picks_the_first_seed_that_passes_everything_and_reports_the_last_examined_flags():
    """x_is_valid over per-seed maxima (cppflow/optimization_utils.py:836-923, thresholds strict `<` as evaluation_utils.py:29-75):
    seeds are examined in order; a seed failing a threshold is skipped BEFORE its collisions are looked at (their flags keep the
    value of the previous seed that got that far, None at first); the first seed passing thresholds and both collision checks is
    returned with its index; with none, the flags of the last seed examined come back."""
    from cppflow_amd.data_types import Constraints
    from cppflow_amd.optimization_utils import x_is_valid

    W, d = 3, 2
    problem = types.SimpleNamespace(n_timesteps=W)
    c = Constraints(max_allowed_position_error_cm=0.01, max_allowed_rotation_error_deg=0.1, max_allowed_mjac_deg=3.0,
                    max_allowed_mjac_cm=2.0)  # fmt: skip

    def metrics(rows):  # (pos cm, rot deg, mjac deg, mjac cm, n_self, n_env)
        m = torch.zeros((len(rows), 16))
        for i, (p, r, a, b, ns, ne) in enumerate(rows):
            m[i, 0], m[i, 2], m[i, 4], m[i, 5], m[i, 9], m[i, 10] = p, r, a, b, ns, ne
        return m

    ok = (0.005, 0.05, 1.0, 0.0, 0, 0)
    x = torch.arange(4 * W * d, dtype=torch.float32).reshape(4 * W, d)
    # seed 0 fails the position threshold (exactly AT the threshold is not below it), 1 collides with itself, 2 with the environment, 3 passes
    m = metrics([(0.01, 0.05, 1.0, 0.0, 0, 0), (0.005, 0.05, 1.0, 0.0, 2, 0), (0.005, 0.05, 1.0, 0.0, 0, 1), ok])
    xs, i, flags = x_is_valid(problem, c, None, x, 4, seed_metrics=m)
    assert i == 3 and torch.equal(xs, x[3 * W :]) and flags == (True, True, True, True, False, False)
    # nobody passes: the flags are those of the LAST seed examined; its collisions were never looked at, so the collision flags
    # still hold what the last seed that got that far left there
    m = metrics([(0.005, 0.05, 1.0, 0.0, 1, 0), (0.005, 0.2, 5.0, 0.0, 0, 0)])
    xs, i, flags = x_is_valid(problem, c, None, x[: 2 * W], 2, seed_metrics=m)
    assert xs is None and i is None and flags == (True, False, False, True, True, None)
    m = metrics([(0.5, 0.05, 1.0, 3.0, 0, 0)])
    xs, i, flags = x_is_valid(problem, c, None, x[:W], 1, seed_metrics=m)
    assert xs is None and flags == (False, True, True, False, None, None)
    with pytest.raises(AssertionError):
        x_is_valid(problem, c, None, x[:W], 2, seed_metrics=m)
