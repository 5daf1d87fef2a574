#!/usr/bin/env python3
"""ms per step of the benchmarked Wide&Deep step (configs[1] shape; whole-step graphs, sinks of 5) with whatever libmrec_hip.so
MREC_HIP_LIB names, and a checksum of what 336 steps left in the tables and the net -- for A/B runs of builds that bench.py cannot
time kernel by kernel (-DMREC_STAMPS=0): python tools/step_time.py [vocab]"""
import hashlib
import os
import sys
import time
from statistics import median

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
cfg = WideDeepConfig(vocab_size=V, emb_dim=80, field_size=26, batch_size=16384)
eng = WideDeepEngine(cfg, dev)
batches = [synthetic_batch(cfg, dev, "uniform", seed=1000 + i) for i in range(8)]
S = 5
for i in range(6):
    eng.train_step(*batches[i % 8])
for i in range(60):
    eng.train_steps([batches[(i * S + j) % 8] for j in range(S)])
torch.cuda.synchronize()
blocks = []
for r in range(5):
    t0 = time.perf_counter()
    for i in range(6):
        eng.train_steps([batches[(i * S + j) % 8] for j in range(S)])
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t0) / 30 * 1e3)
rows = torch.unique(torch.cat([b[0].reshape(-1)[:4096] for b in batches])).long()
h = hashlib.sha256()
h.update(eng.deep_state[rows].cpu().numpy().tobytes())
h.update(eng.dense_flat.detach().cpu().numpy().tobytes())
print(f"{os.path.basename(os.environ.get('MREC_HIP_LIB', 'libmrec_hip.so'))}: median {median(blocks):.4f} ms/step (min {min(blocks):.4f}, max {max(blocks):.4f}); "
      f"state checksum after {eng.step_count} steps {h.hexdigest()[:16]}")
