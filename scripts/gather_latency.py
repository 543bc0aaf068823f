#!/usr/bin/env python3
"""Latency of ONE small all-gather on an otherwise idle stream (developer tool; one rank = what a one-GPU box can measure):
torch.distributed's nccl backend (c10d -> RCCL) against the C ABI's cppf_allgather_bytes (ncclAllGather on the caller's stream)."""
import ctypes, os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppflow_amd import _hip

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
lib = _hip.lib()
uid = (ctypes.c_char * 128)()
_hip.check(lib.cppf_comm_unique_id(uid))
comm = ctypes.c_void_p()
_hip.check(lib.cppf_comm_init_rank(uid, 0, 1, 0, ctypes.byref(comm)))
os.dup2(fd, 1)
for nbytes in (32 * 128 * 8, 32 * 1024 * 8 * 8):
    src = torch.rand(nbytes // 4, device=dev); dst = torch.empty_like(src)
    aux = torch.cuda.Stream(device=dev)
    def c10d():
        with torch.cuda.stream(aux):
            w = dist.all_gather_into_tensor(dst, src, async_op=True); w.wait()
    def cabi():
        _hip.check(lib.cppf_allgather_bytes(comm, src.data_ptr(), dst.data_ptr(), nbytes, aux.cuda_stream))
    for name, fn in (("c10d all_gather_into_tensor(async) + wait", c10d), ("cppf_allgather_bytes", cabi)):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        host, gpu = [], []
        for _ in range(100):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(aux); t0 = time.perf_counter(); fn(); host.append(time.perf_counter() - t0); b.record(aux)
            torch.cuda.synchronize(); gpu.append(a.elapsed_time(b))
        print(f"{nbytes:8d} B  {name:45s} host {1e6*np.median(host):7.1f} us   on-stream {1e3*np.median(gpu):7.1f} us")
lib.cppf_comm_destroy(comm)
dist.destroy_process_group()
