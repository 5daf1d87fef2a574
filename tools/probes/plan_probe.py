#!/usr/bin/env python3
"""The plan (Unique + inverted index) alone, for a rocprofv3 --kernel-trace --stats run: 30 calls on 16384 x 26 uniform ids."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
cfg = WideDeepConfig()
dist_kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
batches = [synthetic_batch(cfg, dev, dist_kind, seed=1000 + i)[0] for i in range(4)]
for i in range(34):
    ops.sparse_plan(batches[i % 4])
torch.cuda.synchronize()
