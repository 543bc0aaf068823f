"""cppflow_amd -- MI355X-native implementation of jstmn/cppflow's batched LM-IK refinement hot path.

Python keeps the reference's call surface for that path (`run_lm_optimization`, `levenberg_marquardt_only_pose`,
`get_6d_pose_errors`, `clamp_to_joint_limits`, `qpaths_batched_*_collisions`, `Problem`, `Constraints`, ...); every
compute call is one launch of a hand-written gfx950 kernel through the C ABI in include/cppflow_hip.h.
"""

__version__ = "0.1.0"
