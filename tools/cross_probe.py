#!/usr/bin/env python3
"""Runs the DCN cross stack forward/backward a few times at configs[2] shape (for rocprofv3 --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, D, L = 16384, 1170, 6
x0 = torch.randn(B, D, device=dev) * 0.5
w = torch.randn(L, D, device=dev) / D ** 0.5
b = torch.randn(L, D, device=dev) * 0.1
dy = torch.randn(B, D, device=dev)
for _ in range(10):
    ops.cross_layers(x0, w, b)
    ops.cross_layers_bwd(x0, w, b, dy)
torch.cuda.synchronize()
