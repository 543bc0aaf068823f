#!/usr/bin/env python3
"""Developer probe for `rocprofv3 --kernel-trace`: dp_search at k = 175 / 64, T = 256, each method 20 times (table, resident)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cppflow_amd.robots import get_robot
dev = torch.device("cuda:0")
rb = get_robot("panda")
for k in (175, 64):
    q = torch.rand((k, 256, 7), device=dev); ext = torch.zeros((k, 256), device=dev)
    for _ in range(20):
        rb.dp_search(q, ext, method="table")
    torch.cuda.synchronize()
    for _ in range(20):
        rb.dp_search(q, ext, method="resident")
    torch.cuda.synchronize()
