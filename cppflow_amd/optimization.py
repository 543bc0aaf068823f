"""The LM optimiser surface of the reference (`cppflow/optimization.py`): dataclasses (`:26-57`), the batched pose-only
step (`:61-92`), the loop (`:147-373`) and the entry point (`:376-426`).

What runs where
  * `levenberg_marquardt_only_pose` = ONE launch of the fused kernel with K = 1 and no clamp (x_new, and J / e scaled
    exactly as the reference returns them when `return_residual=True`).
  * `levenberg_marquardt_full` = the coupled step (`:95-144`) as a block-tridiagonal solve per trajectory on the device
    (`cppf_lm_full_step`), for any number of seeds.
  * `run_lm_alternating_loss` keeps the reference's Python control flow (alternation rule, TL convergence, termination)
    around those launches.
  * `run_lm_pose_refinement` is the batched form the MI355X path is built for: all S seeds x W waypoints, K fused
    iterations, per-seed validity and collision masks / search cost in the same launch.
"""

import warnings
from dataclasses import dataclass
from time import time
from typing import Dict, Optional

import torch

from cppflow_amd.config import ENV_COLLISIONS_IGNORED, SELF_COLLISIONS_IGNORED
from cppflow_amd.data_types import Constraints, Problem
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, OptimizationParameters
from cppflow_amd.optimization_utils import LmResidualFns, clamp_to_joint_limits, evaluate_seeds, x_is_valid
from cppflow_amd.utils import make_text_green_or_red


@dataclass
class OptimizationProblem:
    problem: Problem
    constraints: Constraints
    seed: torch.Tensor
    target_path: torch.Tensor
    verbosity: int
    parallel_count: int
    results_df: Optional[Dict]

    @property
    def robot(self):
        return self.problem.robot

    @property
    def n_timesteps(self) -> int:
        return self.problem.n_timesteps


@dataclass
class OptimizationState:
    x: torch.Tensor
    n_steps: int
    t0: float


@dataclass
class OptimizationResult:
    x_opt: torch.Tensor
    n_steps_taken: int
    is_valid: bool
    parallel_seed_idx: int


def _unstacked_target(opt_problem: OptimizationProblem) -> torch.Tensor:
    """The kernels index target[row % W]; the reference's stacked [k*W, 7] tensor is just W rows repeated."""
    W = opt_problem.n_timesteps
    t = opt_problem.target_path
    return t if t.shape[0] == W else t[:W]


def levenberg_marquardt_only_pose(
    opt_problem: OptimizationProblem,
    opt_state: OptimizationState,
    opt_params: OptimizationParameters,
    return_residual: bool = False,
):
    """One batched pose-only LM step: `x + (J^T J + lambda I)^-1 J^T e` with rows of J / e scaled by alpha_rotation
    (rows 0:3) and alpha_position (rows 3:6).  Returns x_new, or (x_new, J [n,6,d], e [n,6,1]) -- J and e scaled, as the
    reference returns them (cppflow/optimization.py:77-80, 90-92)."""
    n, ndof = opt_state.x.shape
    assert ndof == opt_problem.robot.ndof
    assert opt_problem.target_path.shape[0] in (n, opt_problem.n_timesteps), "target_path must be [n,7] or [W,7]"
    res = opt_problem.robot.lm_pose_steps(
        opt_state.x,
        _unstacked_target(opt_problem),
        lm_lambda=opt_params.lm_lambda,
        alpha_position=opt_params.alpha_position,
        alpha_rotation=opt_params.alpha_rotation,
        n_steps=1,
        clamp=False,
        return_residual=return_residual,
    )
    if return_residual:
        return res["x"], res["J"], res["e"]
    return res["x"]


def levenberg_marquardt_full(
    opt_problem: OptimizationProblem,
    opt_state: OptimizationState,
    opt_params: OptimizationParameters,
    return_residual: bool = False,
):
    """The coupled LM step (cppflow/optimization.py:116-144): pose / differencing / virtual-config / capsule-collision
    residuals of whole trajectories, solved as a block-tridiagonal system per trajectory on the device
    (`cppf_lm_full_step`).  Unlike the reference (`assert parallel_count == 1`, :128) any number of seeds is accepted:
    opt_state.x is [parallel_count * W, ndof] and every trajectory is smoothed independently in the same launch."""
    opt_problem.problem.bind_obstacles()
    x_new = opt_problem.robot.lm_full_step(
        opt_state.x, _unstacked_target(opt_problem), opt_params, virtual_configs=opt_params.virtual_configs
    )
    if not return_residual:
        return x_new
    # inspection path: the dense (J, r) the reference would have factored (the step above never forms them)
    assert opt_problem.parallel_count == 1, "the dense residual / Jacobian are defined for one trajectory (:128)"
    jacobian, residual = LmResidualFns.get_r_and_J(
        opt_params, opt_problem.robot, opt_state.x, _unstacked_target(opt_problem),
        Tcuboids=opt_problem.problem.obstacles_Tcuboids, cuboids=opt_problem.problem.obstacles_cuboids,
    )  # fmt: skip
    return x_new, jacobian, residual


def run_lm_alternating_loss(
    opt_problem: OptimizationProblem,
    opt_state: OptimizationState,
    params_diff: OptimizationParameters,
    params_pose: OptimizationParameters,
    return_residuals: bool,
    tmax_sec: Optional[float],
    max_n_steps: Optional[int],
    return_if_valid_after_n_steps: Optional[int],
    convergence_threshold: float,
    verbosity: int = 0,
    save_images: bool = False,
    results_df: Optional[Dict] = None,
    on_pose_valid: str = "differencing",
):
    """The alternating loop of cppflow/optimization.py:147-373 with its bookkeeping and termination rules.

    Per iteration: if both pose flags are valid take the coupled differencing step (virtual configs := current x, :253),
    else a pose-only step; clamp; evaluate `x_is_valid`; TL-convergence (:275-297) and termination (:326-358) as in the
    reference.  `on_pose_valid` = "stop" / "continue" replace the differencing branch by stopping / more pose steps."""
    assert not return_residuals and not save_images and results_df is None, "debug outputs are not supported"
    assert on_pose_valid in ("differencing", "stop", "continue")
    if tmax_sec is None:
        assert (max_n_steps is not None) and (return_if_valid_after_n_steps is not None)
        assert return_if_valid_after_n_steps <= max_n_steps
    if max_n_steps is None:
        assert tmax_sec is not None
        max_n_steps = 10**6
    robot = opt_problem.robot
    target = _unstacked_target(opt_problem)
    printc = print if verbosity > 1 else (lambda *a, **k: None)
    W = opt_problem.n_timesteps
    # copies: the loop overwrites virtual_configs (the reference copies for the same reason, :184-187)
    params_diff = OptimizationParameters(**params_diff.__dict__)
    params_pose = OptimizationParameters(**params_pose.__dict__)

    tls_post_differencing = []
    last_valid, last_valid_idx, valid_seed_idx = None, -1, 0
    pose_pos_valid, pose_rot_valid = True, False  # the reference's initial values (:218-219): lead with a pose step
    converged = False
    t0 = time()
    i = -1
    for i in range(max_n_steps):
        took_differencing = False
        if pose_pos_valid and pose_rot_valid and on_pose_valid == "stop":
            printc("  pose is valid -- stopping (on_pose_valid='stop')")
            break
        if pose_pos_valid and pose_rot_valid and on_pose_valid == "differencing":
            printc(f"i: {i}  ----> differencing")
            params_diff.virtual_configs = opt_state.x.clone()  # :253
            x_new = levenberg_marquardt_full(opt_problem, opt_state, params_diff)
            took_differencing = True
        else:
            printc(f"i: {i}  --> only pose")
            x_new = levenberg_marquardt_only_pose(opt_problem, opt_state, params_pose)
        opt_state.x = clamp_to_joint_limits(robot, x_new)  # :259
        opt_state.n_steps += 1

        # one evaluation of every trajectory per iteration: validity maxima, collision counts and the TL measure (the summed
        # revolute path length, :221-227) come back in a single [S,16] host tensor
        seed_metrics = evaluate_seeds(opt_problem.problem, target, opt_state.x, opt_problem.parallel_count)
        tl_new = float(seed_metrics[:, 6].sum())
        printc(f"  tl: {tl_new}")
        stop_now = False
        if took_differencing:  # :275-297
            if not converged and len(tls_post_differencing) > 0:
                diff = abs(tl_new - tls_post_differencing[-1])
                if diff < convergence_threshold:
                    converged = True
                    if last_valid_idx == i - 1:
                        stop_now = True
            tls_post_differencing.append(tl_new)
        if stop_now:
            break

        x_sol, seed_idx, flags = x_is_valid(
            opt_problem.problem, opt_problem.constraints, target, opt_state.x, opt_problem.parallel_count, verbosity=verbosity,
            seed_metrics=seed_metrics,
        )
        pose_pos_valid, pose_rot_valid = flags[0], flags[1]
        if x_sol is not None:
            last_valid_idx, last_valid, valid_seed_idx = i, opt_state.x.clone(), seed_idx
            if converged:
                printc(make_text_green_or_red("  x is valid and TL has converged, exiting", True))
                break
            printc(make_text_green_or_red("  x is valid, continuing", True))
        if tmax_sec is not None and time() - t0 > tmax_sec:
            if last_valid is not None:
                opt_state.x = last_valid.clone()
            break
        if last_valid is not None and return_if_valid_after_n_steps is not None and i > return_if_valid_after_n_steps:
            break
    x_return = last_valid if last_valid is not None else opt_state.x
    return OptimizationResult(
        x_opt=x_return, n_steps_taken=max(i, 0), is_valid=last_valid is not None, parallel_seed_idx=valid_seed_idx
    )


def run_lm_optimization(
    problem: Problem,
    x_seed: torch.Tensor,
    tmax_sec: Optional[float],
    max_n_steps: int,
    return_if_valid_after_n_steps: int,
    convergence_threshold: float,
    parallel_count: int = 1,
    results_df: Optional[Dict] = None,
    verbosity: int = 1,
    on_pose_valid: str = "differencing",
) -> OptimizationResult:
    """Optimise a trajectory (or `parallel_count` seeds at once): x_seed is [parallel_count * W, ndof]
    (cppflow/optimization.py:376-426).  The target path is NOT stacked: rows index it modulo W."""
    if SELF_COLLISIONS_IGNORED:
        warnings.warn("robot-robot are collisions will be ignored during LM optimization")
    if ENV_COLLISIONS_IGNORED:
        warnings.warn("environment-robot collisions will be ignored during LM optimization")
    assert problem.target_path.shape == (problem.n_timesteps, 7)
    assert problem.n_timesteps * parallel_count == x_seed.shape[0]
    assert x_seed.shape[1] == problem.robot.ndof
    assert isinstance(max_n_steps, int), f"error: max_n_steps must be int, is {type(max_n_steps)}"
    opt_problem = OptimizationProblem(
        problem, problem.constraints, x_seed, problem.target_path, verbosity, parallel_count, results_df
    )
    opt_state = OptimizationState(x_seed.clone(), 0, time())
    return run_lm_alternating_loss(
        opt_problem, opt_state, ALT_LOSS_V2_1_DIFF, ALT_LOSS_V2_1_POSE, return_residuals=False, verbosity=verbosity,
        tmax_sec=tmax_sec, max_n_steps=max_n_steps, return_if_valid_after_n_steps=return_if_valid_after_n_steps,
        convergence_threshold=convergence_threshold, save_images=False, results_df=results_df, on_pose_valid=on_pose_valid,
    )  # fmt: skip


@dataclass
class PoseRefinementResult:
    x: torch.Tensor  # [S*W, d]
    pos_err_m: torch.Tensor  # [S, W]
    rot_err_rad: torch.Tensor  # [S, W]
    self_mask: torch.Tensor  # bool [S, W]
    env_mask: torch.Tensor  # bool [S, W]
    jlim_mask: torch.Tensor  # bool [S, W]
    ext_cost: torch.Tensor  # float [S, W]: 100*jlim + 1000*env + 1000*self (cppflow/search.py:146-150)
    packed: torch.Tensor  # the uint8 buffer the six per-row outputs live in (what the all-gather ships)


def run_lm_pose_refinement(
    problem: Problem,
    x_seeds: torch.Tensor,
    n_steps: int,
    params_pose: OptimizationParameters = ALT_LOSS_V2_1_POSE,
    x_out: Optional[torch.Tensor] = None,
    packed_out: Optional[torch.Tensor] = None,
) -> PoseRefinementResult:
    """All seeds x waypoints through `n_steps` fused { pose step ; clamp } iterations in ONE launch, with pose errors,
    collision / joint-limit masks and the search cost of the result written to one packed buffer."""
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    W = problem.n_timesteps
    assert x_seeds.dim() == 2 and x_seeds.shape[0] % W == 0, tuple(x_seeds.shape)
    S, n = x_seeds.shape[0] // W, x_seeds.shape[0]
    robot = problem.robot
    problem.bind_obstacles()
    robot.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)
    if packed_out is None:
        packed_out = torch.empty(robot.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=x_seeds.device)
    r = robot.lm_pose_steps(
        x_seeds, problem.target_path, params_pose.lm_lambda, params_pose.alpha_position, params_pose.alpha_rotation,
        n_steps=n_steps, clamp=True, x_out=x_out, packed_out=packed_out,
    )  # fmt: skip
    return PoseRefinementResult(
        x=r["x"], pos_err_m=r["pos_err_m"].view(S, W), rot_err_rad=r["rot_err_rad"].view(S, W),
        self_mask=r["self_mask"].view(S, W).view(torch.bool), env_mask=r["env_mask"].view(S, W).view(torch.bool),
        jlim_mask=r["jlim_mask"].view(S, W).view(torch.bool), ext_cost=r["ext_cost"].view(S, W), packed=packed_out,
    )  # fmt: skip
