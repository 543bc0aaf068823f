"""Obstacle sets of the reference's Panda problems, restated as data (`cppflow/problems/panda__2cubes.yaml:10-28`,
`panda__1cube.yaml:10-22`): (x, y, z, size_x, size_y, size_z), axis-aligned, and the conversion to the
(cuboid[6], Tcuboid[4,4]) pair of `cppflow/data_type_utils.py:109-124`."""

from typing import List, Sequence, Tuple

import numpy as np

PANDA_2CUBES_OBSTACLES = [(0.2, 0.3, 0.4, 0.15, 0.15, 0.15), (-0.25, 0.3, 0.75, 0.15, 0.15, 0.15)]
PANDA_1CUBE_OBSTACLES = [(0.0, 0.2, 0.7, 0.25, 0.25, 0.25)]


def obstacle_arrays(obstacles: Sequence[Tuple[float, ...]]) -> List[Tuple[np.ndarray, np.ndarray]]:
    out = []
    for x, y, z, sx, sy, sz in obstacles:
        cuboid = np.array([-sx / 2, -sy / 2, -sz / 2, sx / 2, sy / 2, sz / 2], dtype=np.float32)
        T = np.zeros((4, 4), dtype=np.float32)  # element [3,3] stays 0, as in the reference loader
        T[:3, :3] = np.eye(3, dtype=np.float32)
        T[:3, 3] = (x, y, z)
        out.append((cuboid, T))
    return out
