#!/usr/bin/env python3
"""MapParameter.get at the configs[4] shape (int64 keys, D = 128, 16384 x 26 positions), resident keys or all-new keys only
(`python tools/map_probe.py resident|new`), for a rocprofv3 --kernel-trace --stats run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.experimental import MapParameter  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "resident"
B, F, Dm = 16384, 26, 128
m = MapParameter(key_dtype=torch.int64, value_shape=(Dm,), capacity=1 << 23, device=dev)
warm = torch.randint(0, 2 ** 22, (B, F), dtype=torch.int64, device=dev)
m.get(warm)
torch.cuda.synchronize()
for i in range(12):
    if mode == "resident":
        m.get(warm)
    else:
        m.get(torch.randint(0, 2 ** 40, (B, F), dtype=torch.int64, device=dev))
torch.cuda.synchronize()
print("live keys", len(m))
