#!/usr/bin/env python3
"""End-to-end example on one MI355X: load a problem in the reference's yaml/csv format, draw candidate paths, run
collision masks -> dp_search -> LM optimisation, print the plan.   python scripts/plan_example.py [problem] [k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppflow_amd.data_type_utils import problem_from_filename  # noqa: E402
from cppflow_amd.data_types import PlannerSettings  # noqa: E402
from cppflow_amd.planners import CppFlowPlanner, LmIkSeedProvider  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "panda__line"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 175  # the reference's default candidate count (scripts/evaluate.py:264-272)
problem = problem_from_filename(None, name, device="cuda:0")
print(problem)
planner = CppFlowPlanner(PlannerSettings(k=k, tmax_sec=5.0, anytime_mode_enabled=False, verbosity=0), problem.robot,
                         LmIkSeedProvider(seed=0))
planner.generate_plan(problem)  # warm-up (library load, first launches)
torch.cuda.synchronize()
t0 = time.perf_counter()
result = planner.generate_plan(problem)
torch.cuda.synchronize()
print(result.plan)
print(f"total {1e3 * (time.perf_counter() - t0):.1f} ms  |  seeds {1e3 * result.timing.ikflow:.1f}  masks "
      f"{1e3 * result.timing.coll_checking:.2f}  dp_search {1e3 * result.timing.dp_search:.2f}  optimiser "
      f"{1e3 * result.timing.optimizer:.2f} ms  ({result.debug_info.get('n_optimization_steps')} LM steps)")
