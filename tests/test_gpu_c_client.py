"""The C boundary without Python in the loop: tests/c_client/abi_client.c (plain C99 + the HIP runtime's C API) drives
libcppflow_hip.so through include/cppflow_hip.h in its own process; its outputs must equal the Python mirror's bit for bit."""

import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from cppflow_amd import _hip
from tests import helpers as H
from tests.test_abi import build_c_client

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name,S,W", [("panda", 6, 128), ("fetch", 9, 37)])
def test_c_client_matches_python_mirror(tmp_path, name, S, W):
    from cppflow_amd.robot_model import MAX_DOF
    from cppflow_amd.robots import get_robot

    exe = build_c_client(str(tmp_path))
    rb = get_robot(name)
    obs = H.PANDA_2CUBES
    cuboids = np.stack([np.asarray(c, dtype=np.float32) for c, _ in obs])
    Rt = np.stack([np.concatenate([np.asarray(T)[:3, :3].ravel(), np.asarray(T)[:3, 3]]) for _, T in obs]).astype(np.float32)
    rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
    rb.set_joint_limit_padding(np.deg2rad(1.5), 0.03)
    jl_lo, jl_hi = (np.zeros(MAX_DOF, dtype=np.float32) for _ in range(2))  # float[CPPF_MAX_DOF] in the C client
    lo, hi = rb.padded_joint_limits()
    jl_lo[: rb.ndof], jl_hi[: rb.ndof] = lo, hi
    K = 4
    x0, target = H.lm_problem(name, S, W, seed=77)
    x0, target = np.ascontiguousarray(x0, dtype=np.float32), np.ascontiguousarray(target, dtype=np.float32)
    desc = _hip.chain_to_desc(H.chain(name))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.array([S, W, K, len(obs)], dtype=np.int32).tobytes())
        f.write(bytes(desc))
        for a in (cuboids, Rt, jl_lo, jl_hi, x0, target):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(_hip.LIB_PATH) + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    run = subprocess.run([exe, str(fin), str(fout)], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "allgather ok on 1 rank(s)" in run.stdout, run.stdout  # RCCL through the C ABI (cppf_comm_init_all / cppf_allgather_bytes)
    assert "lifetime ok" in run.stdout, run.stdout  # robot destroyed before its batch: launch refused, no fault (cppflow_hip.h "Ownership")
    n, d = S * W, rb.ndof
    raw = np.fromfile(fout, dtype=np.uint8)
    off = 0

    def take(count, dtype):
        nonlocal off
        nbytes = count * np.dtype(dtype).itemsize
        a = raw[off : off + nbytes].view(dtype)
        off += nbytes
        return a

    c_x, c_cost, c_pe, c_re = take(n * d, np.float32), take(n, np.float32), take(n, np.float32), take(n, np.float32)
    c_self, c_env, c_jl = take(n, np.uint8), take(n, np.uint8), take(n, np.uint8)
    c_sum, c_fk = take(S * 8, np.float32), take(n * 7, np.float32)
    assert off == raw.size

    packed = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
    summary = torch.empty((S, 8), dtype=torch.float32, device=DEV)
    r = rb.lm_pose_steps(torch.tensor(x0, device=DEV), torch.tensor(target, device=DEV), 1e-6, 3.5, 0.35, n_steps=K,
                         packed_out=packed, summary_out=summary)  # fmt: skip
    fk = rb.forward_kinematics(r["x"])
    for got, want in ((c_x, r["x"]), (c_cost, r["ext_cost"]), (c_pe, r["pos_err_m"]), (c_re, r["rot_err_rad"]),
                      (c_self, r["self_mask"]), (c_env, r["env_mask"]), (c_jl, r["jlim_mask"]), (c_sum, summary), (c_fk, fk)):  # fmt: skip
        assert np.array_equal(got, want.cpu().numpy().reshape(-1)), name
    assert c_self.sum() + c_env.sum() + c_jl.sum() >= 0 and np.isfinite(c_x).all()
    rb.set_obstacles([], [])
    rb.set_joint_limit_padding(None, None)
