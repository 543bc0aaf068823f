"""Batched capsule collision masks with the reference's names (`cppflow/collision_detection.py:9-86`).

Each function is one launch of the collision kernel over all k*T rows; the distance matrices of the reference
(`[k*T, P]`, one per obstacle) are never written to HBM -- the min / "< 0" / OR reduction happens in registers.
The klampt exact-mesh variants (`:89-131`) are out of scope.
"""

from typing import List

import torch


def get_only_non_colliding_qpaths(
    qpaths: List[torch.Tensor], self_colliding: torch.Tensor, env_colliding: torch.Tensor
) -> List[torch.Tensor]:
    """Keep the qpaths that collide with nothing at every timestep (cppflow/collision_detection.py:9-24)."""
    assert len(qpaths) == self_colliding.shape[0] == env_colliding.shape[0]
    keep = torch.logical_or(self_colliding, env_colliding).sum(dim=1) == 0
    return [qpaths[i] for i in keep.nonzero()[:, 0].tolist()]


def qpaths_batched_env_collisions(problem, q: torch.Tensor) -> torch.Tensor:
    """q [k, T, d] -> bool [k, T]: the config touches any of the problem's cuboids (collision_detection.py:27-49)."""
    assert q.dim() == 3, f"q must be [k x ntimesteps x n_dofs], is {tuple(q.shape)}"
    problem.bind_obstacles()
    return problem.robot.collision_masks(q, only=("env",))["env_mask"]


def qpaths_batched_self_collisions(problem, q: torch.Tensor) -> torch.Tensor:
    """q [k, T, d] -> bool [k, T]: some checked capsule pair overlaps (collision_detection.py:52-69)."""
    assert q.dim() == 3, f"q must be [k x ntimesteps x n_dofs], is {tuple(q.shape)}"
    return problem.robot.collision_masks(q, only=("self",))["self_mask"]


def qpaths_batched_collisions(problem, q: torch.Tensor):
    """Both masks from ONE launch (the native form of planners.py:234-251): (self [k,T], env [k,T])."""
    problem.bind_obstacles()
    r = problem.robot.collision_masks(q, only=("self", "env"))
    return r["self_mask"], r["env_mask"]


def self_colliding_configs_capsule(problem, qpath: torch.Tensor) -> torch.Tensor:
    """qpath [T, d] -> bool [T] (collision_detection.py:72-74)."""
    return problem.robot.collision_masks(qpath.unsqueeze(0), only=("self",))["self_mask"][0]


def env_colliding_configs_capsule(problem, qpath: torch.Tensor) -> torch.Tensor:
    """qpath [T, d] -> bool [T] (collision_detection.py:77-86)."""
    problem.bind_obstacles()
    return problem.robot.collision_masks(qpath.unsqueeze(0), only=("env",))["env_mask"][0]


def env_colliding_links_capsule(problem, q: torch.Tensor) -> List[str]:
    """Names of the links whose capsule penetrates an obstacle at configuration q [ndof]
    (cppflow/collision_detection.py:135-145); column i of the distances belongs to the i-th capsule link."""
    ordered_links = list(problem.robot._collision_capsules_by_link.keys())
    hit = set()
    for cuboid, Tcuboid in zip(problem.obstacles_cuboids, problem.obstacles_Tcuboids):
        dists = problem.robot.env_collision_distances(q.reshape(1, -1), cuboid, Tcuboid)[0]
        hit.update(ordered_links[i] for i in torch.nonzero(dists < 0).reshape(-1).tolist())
    return sorted(hit)
