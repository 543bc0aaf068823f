#!/usr/bin/env python3
"""dp_search latency (developer tool): the table form (cppf_dp_search_tabled), the resident single-launch form and one launch per
waypoint (cppf_dp_search), HIP events, medians."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppflow_amd import _hip
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
for name in ("panda", "fetch", "chain12"):
    rb = get_robot(name)
    for k, T in ((175, 256), (64, 256), (96, 256), (128, 256), (175, 64), (256, 256), (300, 256), (512, 256), (1024, 256), (1024, 64)):
        q = torch.rand((k, T, rb.ndof), device=dev)
        ext = torch.zeros((k, T), device=dev)
        qT = torch.empty((T, k, rb.ndof), device=dev); cT = torch.empty((T, k), device=dev); mT = torch.empty((T, k), dtype=torch.int32, device=dev)
        bp = torch.empty((T, rb.ndof), device=dev); bi = torch.empty(T, dtype=torch.int32, device=dev)
        h = rb._handle(dev)
        def call(mode):
            _hip.check(_hip.lib().cppf_dp_search(h, q.data_ptr(), ext.data_ptr(), k, T, 5.0, qT.data_ptr(), cT.data_ptr(), mT.data_ptr(), bp.data_ptr(), bi.data_ptr(), mode, torch.cuda.current_stream(dev).cuda_stream))
        out = []
        for mode in (_hip.DP_RESIDENT, _hip.DP_LAUNCHES) + ((-_hip.DP_RESIDENT,) if k > 256 else ()):
            # (a negative mode: the resident launch in its 512-lane form beyond 256 candidates, CPPF_TUNE_DP_PERSISTENT = 2 -- the A/B
            # of the 1 024-lane form that is the default since round 4)
            rb.debug_set("dp_persistent", 2 if mode < 0 else 1)
            call_m = lambda: call(abs(mode))
            for _ in range(5): call_m()
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call_m(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
            out.append(np.median(ts))
        rb.debug_set("dp_persistent", 1)
        assert int(bi[0].item()) >= 0
        tabled = float("nan")
        if k <= 256:
            import ctypes
            nt = ctypes.c_size_t(0)
            _hip.check(_hip.lib().cppf_dp_table_floats(k, T, ctypes.byref(nt)))
            tab = torch.empty(nt.value, device=dev)
            def call_t():
                _hip.check(_hip.lib().cppf_dp_search_tabled(h, q.data_ptr(), ext.data_ptr(), k, T, 5.0, qT.data_ptr(), cT.data_ptr(), mT.data_ptr(), tab.data_ptr(), bp.data_ptr(), bi.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            for _ in range(5): call_t()
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call_t(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
            tabled = np.median(ts)
        print(f"{name:8s} dp_search k={k:5d} T={T:4d}   table + one compute unit {tabled:8.1f} us   resident single launch {out[0]:8.1f} us   per-waypoint launches {out[1]:8.1f} us" + (f"   resident, 512-lane form {out[2]:8.1f} us" if len(out) > 2 else ""), flush=True)
