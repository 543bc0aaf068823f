"""Small helpers of the reference's `cppflow/utils.py` that the hot path and its tests use."""

import os
import random
from time import time

import numpy as np
import torch

from cppflow_amd.config import DEFAULT_TORCH_DTYPE, DEVICE


def set_seed(seed: int = 0) -> None:
    """Same seeding as cppflow/utils.py:196-204 (torch CPU + all GPUs, numpy, random)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(0)


def to_torch(x, device: str = DEVICE, dtype: torch.dtype = DEFAULT_TORCH_DTYPE) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x
    return torch.tensor(x, device=device, dtype=dtype)


def to_numpy(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return x


def cm_to_m(x):
    return x / 100.0


def m_to_mm(x):
    return x * 1000.0


def cm_to_mm(x):
    return x * 10.0


def make_text_green_or_red(text: str, print_green: bool) -> str:
    return ("\033[1;32m" if print_green else "\033[1;31m") + str(text) + "\033[0m"


class TimerContext:
    """Wall-clock context manager (cppflow/utils.py:130-143); exceptions propagate as RuntimeError like there."""

    def __init__(self, name: str, enabled: bool = True):
        self._name, self._enabled = name, enabled

    def __enter__(self):
        self._t0 = time()
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None:
            raise RuntimeError(f"Error caught by TimerContext('{self._name}'): '{exc}'") from exc
        if self._enabled:
            print(f" --> {self._name} took {round(time() - self._t0, 6)}s")
        return False
