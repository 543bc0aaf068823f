"""Planner surface of the reference (`cppflow/planners.py`): `Planner` / `PlannerSearcher` / `CppFlowPlanner` with
`generate_plan(problem) -> PlannerResult`, wired to the device pipeline

    candidate q-paths [k,T,d]  ->  collision masks + search cost (one launch)  ->  dp_search (device)
                               ->  run_lm_optimization (fused LM launches + coupled differencing steps)

The reference draws its k candidate paths from IKFlow, a pretrained conditional normalizing flow (`planners.py:155-172`);
neither the package nor its weights exist here, so candidates come from a `seed_provider(problem, k) -> [k,T,d]` callable.
`LmIkSeedProvider` is a plain numerical stand-in built on this package's own LM kernel (random restarts at waypoint 0,
then warm-started tracking along the path): it is NOT IKFlow, only a way to exercise the pipeline end to end; anything
that returns a [k,T,d] tensor (an IKFlow wrapper included) can be dropped in.

Seed sharding.  When torch.distributed is initialised with more than one rank (one process per GPU), `_run_pipeline`
shards the k candidates: every rank asks its seed provider for k / world of them (the provider must return RANK-DISTINCT candidates:
`LmIkSeedProvider` offsets its generator by the rank), evaluates -- optionally refines, `candidate_lm_steps` -- its own, and
`cppflow_amd.distributed.sharded_candidate_evaluation` all-gathers the packed per-row outputs and the paths, so that every rank runs
the same `dp_search` over all k candidates (cppflow/planners.py:231-274, cppflow/search.py:146-151) and returns the same plan.
"""

from time import time
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from cppflow_amd.collision_detection import qpaths_batched_collisions
from cppflow_amd.config import OPTIMIZATION_CONVERGENCE_THRESHOLD, SUCCESS_THRESHOLD_initial_q_norm_dist
from cppflow_amd.data_type_utils import plan_from_qpath
from cppflow_amd.data_types import PlannerResult, PlannerSettings, Problem, TimingData
from cppflow_amd.evaluation_utils import get_mjacs
from cppflow_amd.optimization import run_lm_optimization
from cppflow_amd.search import dp_search

DEFAULT_RERUN_NEW_K = 125  # planners.py:47

SeedProvider = Callable[[Problem, int], torch.Tensor]


def add_search_path_mjac(debug_info: Dict, problem: Problem, qpath_search: torch.Tensor) -> None:
    """Diagnostics of the searched path the reference records (cppflow/planners.py:50-73): its maximum joint changes and its
    closest approach to a joint limit (cm for a leading prismatic joint, degrees for the rest) -- one min/max pass over
    [T, d] on the device, one copy back."""
    mjac_deg, mjac_cm = get_mjacs(problem.robot, qpath_search)
    debug_info["search_path_mjac-cm"], debug_info["search_path_mjac-deg"] = mjac_cm, mjac_deg
    limits = torch.tensor(problem.robot.actuated_joints_limits, dtype=qpath_search.dtype, device=qpath_search.device)  # [d, 2]
    margin = torch.minimum((qpath_search - limits[:, 0]).abs().min(dim=0).values,
                           (qpath_search - limits[:, 1]).abs().min(dim=0).values).cpu()  # fmt: skip
    lead_prismatic = problem.robot.has_prismatic_joints  # the reference treats joint 0 as THE prismatic joint (:60)
    debug_info["search_path_min_dist_to_jlim_cm"] = 100 * float(margin[0]) if lead_prismatic else -1
    rest = margin[1:] if lead_prismatic else margin
    debug_info["search_path_min_dist_to_jlim_deg"] = min(float(torch.rad2deg(rest.min())), 10000) if rest.numel() else 10000


class LmIkSeedProvider:
    """k candidate joint-space paths for a problem by numerical IK (stand-in for IKFlow, see module docstring)."""

    def __init__(self, seed: int = 0, damping: float = 1e-2, n_restart_steps: int = 40, n_track_steps: int = 6):
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0  # rank-distinct candidates under sharding
        self._gen = torch.Generator().manual_seed(seed + 7919 * rank)
        self._damping, self._n_restart, self._n_track = damping, n_restart_steps, n_track_steps

    def __call__(self, problem: Problem, k: int) -> torch.Tensor:
        rb, dev = problem.robot, problem.target_path.device
        T, d = problem.n_timesteps, rb.ndof
        lo = torch.tensor([l for l, _ in rb.actuated_joints_limits], dtype=torch.float32)
        hi = torch.tensor([u for _, u in rb.actuated_joints_limits], dtype=torch.float32)
        q = (lo + (hi - lo) * (0.1 + 0.8 * torch.rand((k, d), generator=self._gen))).to(dev)
        out = torch.empty((k, T, d), dtype=torch.float32, device=dev)
        for t in range(T):
            tgt = problem.target_path[t : t + 1].contiguous()
            steps = self._n_restart if t == 0 else self._n_track
            q = rb.lm_pose_steps(q, tgt, self._damping, 3.5, 0.35, n_steps=steps, clamp=True)["x"]
            out[:, t] = q
        return out


class Planner:
    def __init__(self, settings: PlannerSettings, robot, seed_provider: Optional[SeedProvider] = None, process_group=None,
                 candidate_lm_steps: int = 0):
        """`process_group` / `candidate_lm_steps`: the sharded candidate stage (module docstring); with `candidate_lm_steps` > 0 every
        (candidate, waypoint) row takes that many fused pose-only LM iterations before the masks are evaluated (one launch)."""
        self._cfg = settings
        self._robot = robot
        self._seed_provider = seed_provider if seed_provider is not None else LmIkSeedProvider()
        self._group = process_group
        self._candidate_lm_steps = int(candidate_lm_steps)

    @property
    def robot(self):
        return self._robot

    @property
    def name(self) -> str:
        return str(self.__class__.__name__)

    def set_settings(self, settings: PlannerSettings) -> None:
        self._cfg = settings

    def _run_pipeline(self, problem: Problem, **kwargs) -> Tuple[torch.Tensor, bool, TimingData, dict, tuple]:
        """Candidates -> collision masks -> dp_search (cppflow/planners.py:191-292)."""
        existing = kwargs.get("rerun_data")
        k = self._cfg.k if existing is None else DEFAULT_RERUN_NEW_K
        world = dist.get_world_size(self._group) if dist.is_available() and dist.is_initialized() else 1
        t0 = time()
        if world > 1:
            from cppflow_amd.distributed import padded_shard_size, sharded_candidate_evaluation

            k_local = padded_shard_size(k, problem.n_timesteps, world)  # (k is rounded up to world * k_local candidates)
            qs = self._seed_provider(problem, k_local)  # this rank's [k_local, T, d]
            assert qs.dim() == 3 and tuple(qs.shape) == (k_local, problem.n_timesteps, self.robot.ndof), tuple(qs.shape)
        else:
            qs = self._seed_provider(problem, k)  # [k, T, d]
            assert qs.dim() == 3 and qs.shape[1:] == (problem.n_timesteps, self.robot.ndof), tuple(qs.shape)
        time_seeds = time() - t0
        if self._cfg.return_only_1st_plan:
            return qs[0], False, TimingData(-1, time_seeds, 0.0, 0.0, 0.0, 0.0), {}, (qs[0], None, None)

        t0 = time()
        if world > 1:
            qs, self_viol, env_viol = sharded_candidate_evaluation(problem, qs, self._candidate_lm_steps, self._group)
        elif self._candidate_lm_steps > 0:
            from cppflow_amd.distributed import sharded_candidate_evaluation

            qs, self_viol, env_viol = sharded_candidate_evaluation(problem, qs, self._candidate_lm_steps, None)
        else:
            self_viol, env_viol = qpaths_batched_collisions(problem, qs.contiguous())
        for name, v in (("self", self_viol), ("env", env_viol)):
            pct = float(v.float().mean()) * 100
            assert pct < 95.0, f"too many {name} collisions: {pct} %"  # planners.py:237,247
        if existing is not None:
            qs_prev, self_prev, env_prev = existing
            qs = torch.cat([qs_prev, qs], dim=0)
            self_viol, env_viol = torch.cat([self_prev, self_viol], dim=0), torch.cat([env_prev, env_viol], dim=0)
        if problem.initial_configuration is not None:
            qs[:, 0, :] = problem.initial_configuration
            self_viol[:, 0], env_viol[:, 0] = False, False  # assumed collision-free (planners.py:265-266)
        time_coll = time() - t0

        t0 = time()
        qpath_search = dp_search(self.robot, qs.contiguous(), self_viol, env_viol)
        time_dp = time() - t0
        return qpath_search, False, TimingData(-1, time_seeds, time_coll, 0.0, time_dp, 0.0), {}, (qs, self_viol, env_viol)


class PlannerSearcher(Planner):
    """dp_search over k candidate paths, no optimisation (cppflow/planners.py:301-336)."""

    def __init__(self, settings: PlannerSettings, robot, seed_provider: Optional[SeedProvider] = None, **kwargs):
        super().__init__(settings, robot, seed_provider, **kwargs)
        assert self._cfg.run_dp_search

    def generate_plan(self, problem: Problem, **kwargs) -> PlannerResult:
        assert problem.robot.name == self.robot.name
        t0 = time()
        qpath, _, td, debug_info, q_data = self._run_pipeline(problem, **kwargs)
        if self._cfg.do_rerun_if_large_dp_search_mjac:
            mjac_deg, mjac_cm = get_mjacs(problem.robot, qpath)
            if mjac_deg > self._cfg.rerun_mjac_threshold_deg or mjac_cm > self._cfg.rerun_mjac_threshold_cm:
                qpath, _, td, debug_info, _ = self._run_pipeline(problem, rerun_data=q_data)
        return PlannerResult(
            plan_from_qpath(qpath.detach(), problem),
            TimingData(time() - t0, td.ikflow, td.coll_checking, td.batch_opt, td.dp_search, 0.0), [], [], debug_info,
        )  # fmt: skip


class CppFlowPlanner(Planner):
    """Candidates -> dp_search -> LM optimisation (cppflow/planners.py:339-468)."""

    def generate_plan(self, problem: Problem, **kwargs) -> PlannerResult:
        t0 = kwargs.get("t0", time())
        rerun_data = kwargs.get("rerun_data")
        search_qpath, is_valid, td, debug_info, q_data = self._run_pipeline(problem, **kwargs)

        def out_of_time() -> bool:
            return time() - t0 > self._cfg.tmax_sec

        def result(qpath) -> PlannerResult:
            return PlannerResult(
                plan_from_qpath(qpath, problem),
                TimingData(time() - t0, td.ikflow, td.coll_checking, td.batch_opt, td.dp_search, td.optimizer), [], [],
                debug_info,
            )  # fmt: skip

        if self._cfg.return_only_1st_plan:
            return result(search_qpath)
        if self._cfg.do_rerun_if_large_dp_search_mjac:
            mjac_deg, mjac_cm = get_mjacs(problem.robot, search_qpath)
            if mjac_deg > self._cfg.rerun_mjac_threshold_deg or mjac_cm > self._cfg.rerun_mjac_threshold_cm:
                search_qpath, is_valid, td, debug_info, q_data = self._run_pipeline(problem, rerun_data=q_data)
        if out_of_time() or ((not self._cfg.anytime_mode_enabled) and is_valid):
            return result(search_qpath)

        t0_opt = time()
        budget = dict(max_n_steps=75, return_if_valid_after_n_steps=int(1e8),
                      convergence_threshold=OPTIMIZATION_CONVERGENCE_THRESHOLD) if self._cfg.anytime_mode_enabled else dict(
            max_n_steps=20, return_if_valid_after_n_steps=0, convergence_threshold=1e6)  # fmt: skip  (planners.py:402-422)
        opt = run_lm_optimization(problem, search_qpath.contiguous(), tmax_sec=self._cfg.tmax_sec - (time() - t0),
                                  verbosity=self._cfg.verbosity, **budget)  # fmt: skip
        td.optimizer = time() - t0_opt
        debug_info["n_optimization_steps"] = opt.n_steps_taken
        x_opt = opt.x_opt.detach()
        if opt.is_valid:
            if problem.initial_configuration is None:
                return result(x_opt)
            if torch.norm(problem.initial_configuration - x_opt[0]) < SUCCESS_THRESHOLD_initial_q_norm_dist:
                return result(x_opt)
            swapped = torch.cat((problem.initial_configuration, x_opt[1:]), dim=0)
            return result(swapped) if plan_from_qpath(swapped, problem).is_valid else result(x_opt)
        if self._cfg.do_rerun_if_optimization_fails and rerun_data is None and not out_of_time():
            return self.generate_plan(problem, rerun_data=q_data, t0=t0)
        return result(x_opt)
