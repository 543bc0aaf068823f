"""Constants of the reference's `cppflow/config.py` that the hot path reads (`:8-30`).  Unlike the reference this module
does NOT change torch's global default device / dtype: callers pass device tensors explicitly."""

import torch

DEFAULT_TORCH_DTYPE = torch.float32
DEVICE = "cuda:0" if torch.cuda.is_available() else "cpu"

VERBOSITY = 2

SUCCESS_THRESHOLD_initial_q_norm_dist = 0.2
DEFAULT_RERUN_MJAC_THRESHOLD_DEG = 13.0
DEFAULT_RERUN_MJAC_THRESHOLD_CM = 3.42
OPTIMIZATION_CONVERGENCE_THRESHOLD = 0.005

# LM optimization switches read by x_is_valid (cppflow/optimization_utils.py:889,896).  The exact-mesh (klampt) checks
# they guard are outside this build's scope; the capsule masks take their place (DESIGN.md "out of scope").
SELF_COLLISIONS_IGNORED = False
ENV_COLLISIONS_IGNORED = False
DEBUG_MODE_ENABLED = False
