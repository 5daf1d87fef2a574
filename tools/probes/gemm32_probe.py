#!/usr/bin/env python3
"""The three fp32 MFMA DenseLayer launches of Deep&Cross's first layer, a few times each (for rocprofv3 --pmc / --kernel-trace runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, X, H = 16384, 1170, 1024
x = torch.randn(B, X, device=dev)
w = torch.randn(X, H, device=dev) * 0.03
dy = torch.randn(B, H, device=dev)
S = ops.dense32_bwd_weight_slabs(B, X, H)
slabs = torch.empty((S, X, H), device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    ops.dense32_fwd(x, w, None, relu=True)
    ops.dense32_bwd_input(dy, w)
    ops.dense32_bwd_weight(x, dy, slabs)
torch.cuda.synchronize()
print("slabs", S)
