#!/usr/bin/env python3
"""MapParameter.get at configs[4] shape (int64 keys, D = 128, 16384 x 26 keys) for rocprofv3 --kernel-trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.experimental import MapParameter  # noqa: E402

dev = torch.device("cuda:0")
B, F, Dm = 16384, 26, 128
m = MapParameter(key_dtype=torch.int64, value_shape=(Dm,), capacity=1 << 23, device=dev)
warm_keys = torch.randint(0, 2 ** 22, (B, F), dtype=torch.int64, device=dev)
m.get(warm_keys)
mode = sys.argv[1] if len(sys.argv) > 1 else "resident"
for i in range(10):
    if mode == "resident":
        m.get(warm_keys)
    else:
        m.get(torch.randint(0, 2 ** 40, (B, F), dtype=torch.int64, device=dev))
torch.cuda.synchronize()
