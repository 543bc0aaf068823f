#!/usr/bin/env python3
"""Developer measurement: where does a step of the resident dp_search go?  Needs the diagnostic library of
scripts/make_dp_stamp_build.py (CPPFLOW_HIP_LIB=build_var/lib_dpstamp.so): dp_persistent4_kernel stamps the chip-wide 100 MHz
counter at four points of every step in every workgroup.  One k = 175, T = 256 search (the reference's shape), then per step:
  hand-off    = a wavefront has ALL its cost words  -  the last of those words was published (store issued) by its owner
  wait->barrier, barrier->publish (LDS read, DPP minimum, store), publish->next step's wait begins (loop overhead)
and the step period itself (publish of step t+1 - publish of step t, chip-wide maxima)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

dev = torch.device("cuda:0")
rb = get_robot("panda")
k, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (175, 256)
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.rand((k, T, rb.ndof), generator=g).to(dev)
ext = torch.zeros((k, T), device=dev)
for _ in range(20):
    rb.dp_search(q, ext, method="resident")
torch.cuda.synchronize()
rb.dp_search(q, ext, method="resident")
torch.cuda.synchronize()
lib = _hip.lib()
lib.cppf_debug_dp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.cppf_debug_dp_stamps.restype = ctypes.c_int
buf = np.zeros((256, 64, 16), dtype=np.uint64)
assert lib.cppf_debug_dp_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
G = (k + 3) // 4  # workgroups
nw = min(8, (2 * 256) // 64)
st = buf[1:T, :G].astype(np.int64)  # steps 1 .. T-1
t0 = st[st > 0].min()
us = (st - t0) * 0.01
s0, s1, s2, s3 = us[..., 0], us[..., 1:9], us[..., 9], us[..., 10:14]
# publish time of destination b at step t: workgroup b // 4, slot b % 4
pub = np.full((T - 1, 4 * G), np.nan)
for i in range(4):
    pub[:, i::4] = s3[..., i]
pub = pub[:, :k]
# wave w of a workgroup (512 lanes = 2 halves x 256 sources): lanes 64 w .. 64 w + 63 -> sources (64 w) % 256 .. + 63
need = np.zeros((T - 1, G, 8))
for w in range(8):
    a0 = (64 * w) % 256
    srcs = np.arange(a0, min(a0 + 64, k))
    if len(srcs) == 0:
        need[:, :, w] = np.nan
        continue
    last_pub = np.nanmax(pub[:-1, srcs], axis=1)  # step t-1's publishes of those sources, for steps 2 ..
    need[1:, :, w] = last_pub[:, None]
    need[0, :, w] = np.nan
handoff = s1 - need  # [T-1, G, 8]
period = np.diff(np.nanmax(pub, axis=1))


def q5(a):
    a = a[np.isfinite(a)]
    return "p10 %.2f  median %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(a, [10, 50, 90, 100]))


print(f"dp_search k = {k}, T = {T}: {G} workgroups; step period (chip-wide last publish to last publish): {q5(period[5:])} us;  whole search ~ {np.nanmax(pub) - np.nanmin(s0):.1f} us")
print("  hand-off (all cost words of a wavefront there - the last of them published):   ", q5(handoff[5:]))
print("  slowest wavefront's arrival -> barrier passed (LDS image written, s_barrier):   ", q5((s2 - np.nanmax(s1, axis=2))[5:]))
print("  barrier -> destination published (4 LDS reads, DPP minimum, memo + cost store): ", q5((s3 - s2[..., None])[5:]))
print("  published -> the next step's wait begins (loop, the prefetched operands' use):  ", q5((s0[1:] - np.nanmax(s3, axis=2)[:-1])[5:]))
print("  cost-independent part done -> first wavefront has its words (the exposed wait): ", q5((np.nanmin(s1, axis=2) - s0)[5:]))
lag = np.nanmax(pub, axis=1) - np.nanmin(pub, axis=1)
print("  spread of the publishes of one step over the workgroups (last - first):          ", q5(lag[5:]))
