"""Problem loading in the reference's file formats (`cppflow/data_type_utils.py:55-219`): a problem yaml
(`robot`, `path_name`, `path_offset_frame`, `path_xyz_offset`, `path_R_offset`, optional `obstacle_xyz_offset` +
`obstacles`) next to a path csv (`time,x,y,z,qw,qx,qy,qz`, `cppflow/paths/README.md:9`).

The reference resolves files inside its own package and leans on klampt for the frame offset and the quaternion algebra;
here directories are explicit arguments (default: this package's `problems/` and `paths/`) and the algebra is numpy.
"""

import csv
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import yaml

from cppflow_amd.config import ENV_COLLISIONS_IGNORED, SELF_COLLISIONS_IGNORED
from cppflow_amd.data_types import DEFAULT_CONSTRAINTS, Constraints, Plan, Problem
from cppflow_amd.robots import Robot, get_robot

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PROBLEMS_DIR = os.path.join(_HERE, "problems")
DEFAULT_PATHS_DIR = os.path.join(_HERE, "paths")


def _quat_to_matrix(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q / np.linalg.norm(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def _matrix_to_quat(R: np.ndarray) -> np.ndarray:
    """w-first unit quaternion with w >= 0 (the sign is immaterial to every consumer on the hot path)."""
    m = R
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def offset_target_path(
    robot: Robot, target_path: np.ndarray, path_offset_frame: str, xyz_offset: List[float], R_offset: List[List[float]]
) -> np.ndarray:
    """Shift every pose by `xyz_offset` expressed in `path_offset_frame` (evaluated at q = 0; the frame must be unrotated
    w.r.t. the world) and post-multiply every orientation by `R_offset` (cppflow/data_type_utils.py:55-84)."""
    path = np.array(target_path, dtype=np.float64, copy=True)
    if path_offset_frame == "world":
        frame_xyz = np.zeros(3)
    else:
        T = robot.link_frame_at_zero(path_offset_frame)
        np.testing.assert_allclose(T[:3, :3], np.eye(3), atol=1e-8, err_msg="path_offset_frame must be unrotated")
        frame_xyz = T[:3, 3]
    path[:, 0:3] += np.asarray(xyz_offset, dtype=np.float64) + frame_xyz
    R_off = np.asarray(R_offset, dtype=np.float64)
    if not np.allclose(R_off, np.eye(3), atol=1e-12):
        for i in range(path.shape[0]):
            path[i, 3:7] = _matrix_to_quat(_quat_to_matrix(path[i, 3:7]) @ R_off)
    return path


def get_obstacles(problem_dict: Dict) -> Tuple[List[Dict], List[torch.Tensor], List[torch.Tensor]]:
    """yaml `obstacles` -> (dicts, Tcuboids [4,4], cuboids [6]) exactly as cppflow/data_type_utils.py:87-125 builds them:
    cuboid = (-sx/2,-sy/2,-sz/2, sx/2,sy/2,sz/2); Tcuboid has R = I, t = (x,y,z) + obstacle_xyz_offset and [3,3] left 0.
    Tensors are host tensors: the kernels take obstacles through the kernel-argument segment."""
    parsed_list, Ts, cuboids = [], [], []
    for obs in problem_dict.get("obstacles", []) or []:
        parsed: Dict[str, float] = {}
        for d in obs:
            parsed.update(d)
        off = problem_dict["obstacle_xyz_offset"]
        parsed["x"] += off[0]
        parsed["y"] += off[1]
        parsed["z"] += off[2]
        assert abs(parsed["roll"]) < 1e-8 and abs(parsed["pitch"]) < 1e-8 and abs(parsed["yaw"]) < 1e-8
        sx, sy, sz = parsed["size_x"], parsed["size_y"], parsed["size_z"]
        cuboids.append(torch.tensor([-sx / 2, -sy / 2, -sz / 2, sx / 2, sy / 2, sz / 2], dtype=torch.float32))
        T = torch.zeros((4, 4), dtype=torch.float32)
        T[:3, :3] = torch.eye(3)
        T[0, 3], T[1, 3], T[2, 3] = parsed["x"], parsed["y"], parsed["z"]
        Ts.append(T)
        parsed_list.append(parsed)
    return parsed_list, Ts, cuboids


def load_path_csv(filepath: str) -> np.ndarray:
    """[T, 7] = x y z qw qx qy qz (the leading time column is dropped)."""
    with open(filepath, "r") as f:
        rows = [[float(v) for v in row] for i, row in enumerate(csv.reader(f)) if i > 0 and len(row) > 0]
    return np.array(rows, dtype=np.float64)[:, 1:]


def problem_from_filename(
    constraints: Optional[Constraints],
    problem_filename: str,
    filepath_override: Optional[str] = None,
    robot: Optional[Robot] = None,
    problems_dir: str = DEFAULT_PROBLEMS_DIR,
    paths_dir: str = DEFAULT_PATHS_DIR,
    device: Optional[str] = None,
) -> Problem:
    """Parse a problem yaml + its path csv into a `Problem` (cppflow/data_type_utils.py:148-219)."""
    if filepath_override is None:
        assert "yaml" not in problem_filename, "problem_filename should not include the .yaml file extension"
        filepath = os.path.join(problems_dir, problem_filename + ".yaml")
    else:
        filepath = filepath_override
    with open(filepath, "r") as f:
        problem_dict = yaml.load(f, Loader=yaml.FullLoader)
    if robot is None:
        robot = get_robot(problem_dict["robot"])
    else:
        assert "obstacles" not in problem_dict, f"obstacles found for {problem_filename} but a robot was provided"
    obstacles, Ts, cuboids = get_obstacles(problem_dict)
    path_name = problem_dict["path_name"]
    original = load_path_csv(os.path.join(paths_dir, path_name + ".csv"))
    target = offset_target_path(
        robot, original, problem_dict["path_offset_frame"], problem_dict["path_xyz_offset"], problem_dict["path_R_offset"]
    )
    if device is None:
        device = "cuda:0" if torch.cuda.is_available() else "cpu"
    return Problem(
        constraints if constraints is not None else DEFAULT_CONSTRAINTS,
        torch.tensor(target, dtype=torch.float32, device=device),
        None,
        robot,
        path_name,
        problem_filename,
        obstacles,
        Ts,
        cuboids,
        [],
    )


def problem_from_arrays(
    robot: Robot,
    target_path: np.ndarray,
    obstacles_xyz_size: Optional[List[Tuple[float, ...]]] = None,
    name: str = "synthetic",
    constraints: Optional[Constraints] = None,
    device: Optional[str] = None,
) -> Problem:
    """Build a `Problem` from an already-offset [W,7] path and (x,y,z,sx,sy,sz) axis-aligned boxes."""
    from cppflow_amd.problems_synthetic import obstacle_arrays

    pairs = obstacle_arrays(obstacles_xyz_size or [])
    if device is None:
        device = "cuda:0" if torch.cuda.is_available() else "cpu"
    return Problem(
        constraints if constraints is not None else DEFAULT_CONSTRAINTS,
        torch.tensor(np.asarray(target_path), dtype=torch.float32, device=device),
        None,
        robot,
        name,
        f"{robot.name}__{name}",
        [dict(x=o[0], y=o[1], z=o[2], size_x=o[3], size_y=o[4], size_z=o[5]) for o in (obstacles_xyz_size or [])],
        [torch.tensor(T) for _, T in pairs],
        [torch.tensor(c) for c, _ in pairs],
        [],
    )


def resample_path(path: np.ndarray, n: int) -> np.ndarray:
    """Resample a pose path to n waypoints by arc length: positions lerp, orientations slerp (SURVEY.md 8d, config C4:
    the 200-row 2cubes path resampled to 256)."""
    path = np.asarray(path, dtype=np.float64)
    seg = np.linalg.norm(np.diff(path[:, :3], axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    if s[-1] <= 0:
        s = np.linspace(0.0, 1.0, path.shape[0])
    u = np.linspace(0.0, s[-1], n)
    out = np.zeros((n, 7))
    idx = np.clip(np.searchsorted(s, u, side="right") - 1, 0, path.shape[0] - 2)
    for k in range(n):
        i = idx[k]
        h = s[i + 1] - s[i]
        a = 0.0 if h <= 0 else (u[k] - s[i]) / h
        out[k, :3] = (1 - a) * path[i, :3] + a * path[i + 1, :3]
        q0, q1 = path[i, 3:7], path[i + 1, 3:7]
        dot = float(np.dot(q0, q1))
        if dot < 0:
            q1, dot = -q1, -dot
        if dot > 0.9995:
            q = (1 - a) * q0 + a * q1
        else:
            th = np.arccos(np.clip(dot, -1, 1))
            q = (np.sin((1 - a) * th) * q0 + np.sin(a * th) * q1) / np.sin(th)
        out[k, 3:7] = q / np.linalg.norm(q)
    return out


def _problem_files(problems_dir: str) -> List[str]:
    return sorted(f[: -len(".yaml")] for f in os.listdir(problems_dir) if f.endswith(".yaml"))


def _has_obstacles(name: str, problems_dir: str) -> bool:
    with open(os.path.join(problems_dir, name + ".yaml"), "r") as f:
        return len((yaml.load(f, Loader=yaml.FullLoader) or {}).get("obstacles", []) or []) > 0


# The reference hard-codes the names of its problem files (cppflow/data_type_utils.py:24-52); here the lists are read off
# the problems directory that ships with the package (point `problems_dir` at the reference's own directory to get its set).
ALL_PROBLEM_FILENAMES: List[str] = _problem_files(DEFAULT_PROBLEMS_DIR)
ALL_OBS_PROBLEM_FILENAMES: List[str] = [n for n in ALL_PROBLEM_FILENAMES if _has_obstacles(n, DEFAULT_PROBLEMS_DIR)]


def get_problem_dict(problem_names: List[str], problems_dir: str = DEFAULT_PROBLEMS_DIR, paths_dir: str = DEFAULT_PATHS_DIR,
                     device: Optional[str] = None) -> Dict[str, Problem]:  # fmt: skip
    """name -> Problem (cppflow/data_type_utils.py:222-236).  Obstacle-free problems of one robot share one Robot object
    (one device handle), which is what the reference does by hand for its Fetch problems; a problem with obstacles gets
    its own, because obstacles are state of the robot handle."""
    shared: Dict[str, Robot] = {}
    out: Dict[str, Problem] = {}
    for name in problem_names:
        with open(os.path.join(problems_dir, name + ".yaml"), "r") as f:
            spec = yaml.load(f, Loader=yaml.FullLoader)
        robot = None
        if not spec.get("obstacles"):
            robot = shared.setdefault(spec["robot"], get_robot(spec["robot"]))
        out[name] = problem_from_filename(None, name, robot=robot, problems_dir=problems_dir, paths_dir=paths_dir, device=device)
    return out


def get_all_problems(problems_dir: str = DEFAULT_PROBLEMS_DIR, paths_dir: str = DEFAULT_PATHS_DIR,
                     device: Optional[str] = None) -> List[Problem]:  # fmt: skip
    """Every problem of the directory, in name order (cppflow/data_type_utils.py:239-241)."""
    names = _problem_files(problems_dir)
    d = get_problem_dict(names, problems_dir, paths_dir, device)
    return [d[n] for n in names]


def plans_from_qpaths(qpaths: torch.Tensor, problem: Problem) -> List[Plan]:
    """Evaluate S joint-space paths [S,T,d] against one problem in three launches -- FK of every waypoint, the capsule
    collision masks, and the per-path `Plan` metrics (cppf_plan_metrics) -- instead of the reference's per-path host
    evaluation (cppflow/data_type_utils.py:244-276).  Collisions are the capsule masks of the hot path (the reference
    re-checks the final path with klampt's meshes here, which is out of scope)."""
    rb = problem.robot
    assert qpaths.dim() == 3 and qpaths.shape[1:] == (problem.n_timesteps, rb.ndof), tuple(qpaths.shape)
    S, T, d = qpaths.shape
    q = qpaths.contiguous()
    flat = q.view(S * T, d)
    pose = rb.forward_kinematics(flat).view(S, T, 7)
    pe, re = rb.pose_error_metrics(flat, problem.target_path)
    problem.bind_obstacles()
    masks = rb.collision_masks(q, only=("self", "env"))
    self_c, env_c = masks["self_mask"], masks["env_mask"]
    if SELF_COLLISIONS_IGNORED:
        self_c = torch.zeros_like(self_c)
    if ENV_COLLISIONS_IGNORED:
        env_c = torch.zeros_like(env_c)
    metrics = rb.plan_metrics(flat, problem.target_path, self_c.reshape(-1), env_c.reshape(-1),
                              problem.initial_configuration).cpu()  # fmt: skip
    rev, pris = rb.split_configs_to_revolute_and_prismatic(flat)
    rev, pris = rev.view(S, T, -1), pris.view(S, T, -1)
    pe, re = pe.view(S, T), re.view(S, T)
    return [
        Plan(
            q_path=q[s], q_path_revolute=rev[s], q_path_prismatic=pris[s], pose_path=pose[s], target_path=problem.target_path,
            robot_joint_limits=rb.actuated_joints_limits, self_colliding_per_ts=self_c[s].bool(),
            env_colliding_per_ts=env_c[s].bool(), positional_errors=pe[s], rotational_errors=re[s],
            provided_initial_configuration=problem.initial_configuration, constraints=problem.constraints, metrics=metrics[s],
        )  # fmt: skip
        for s in range(S)
    ]


def plan_from_qpath(qpath: torch.Tensor, problem: Problem) -> Plan:
    """One path [T,d] -> Plan (cppflow/data_type_utils.py:244-276)."""
    assert isinstance(qpath, torch.Tensor), f"qpath must be a torch.Tensor, got {type(qpath)}"
    assert qpath.shape == (problem.n_timesteps, problem.robot.ndof), tuple(qpath.shape)
    return plans_from_qpaths(qpath.unsqueeze(0), problem)[0]