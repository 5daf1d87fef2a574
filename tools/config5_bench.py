#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: DeepFM over MapParameter hash embeddings (int64 keys, dim 128, admission 2 /
eviction 100 steps), batch 16384 x 26 keys drawn Zipf-like from a 2^40 key space.  ms/step from HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.deepfm import DeepFMConfig, DeepFMHashEngine  # noqa: E402

dev = torch.device("cuda:0")
B, F, D = 16384, 26, 128
mlp_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = DeepFMConfig(data_emb_dim=D, data_field_size=F, batch_size=B, mlp_dtype=mlp_dtype)
eng = DeepFMHashEngine(cfg, dev, key_dtype=torch.int64, capacity=1 << 23, permit_filter_value=2, evict_filter_value=100)
rng = np.random.default_rng(7)


def batch():
    # per-slot Zipf over a private 2^35 range, scrambled into the 2^40 key space
    z = rng.zipf(1.1, size=(B, F)).astype(np.int64) % (1 << 35)
    keys = (z * 0x9E3779B1 + np.arange(F, dtype=np.int64)[None, :] * (1 << 35)) % (1 << 40)
    return (torch.from_numpy(keys).to(dev), torch.ones((B, F), device=dev),
            torch.from_numpy((rng.random((B, 1)) < 0.3).astype(np.float32)).to(dev))


batches = [batch() for _ in range(8)]
for i in range(4):
    eng.train_step(*batches[i % 8])
torch.cuda.synchronize()
steps = 20
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(steps):
    loss = eng.train_step(*batches[i % 8])
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / steps
print(f"configs[4] step (DeepFM + hash tables, int64 keys, D=128, permit 2, MLP {mlp_dtype}): {ms:.3f} ms = {B / ms * 1e3 / 1e6:.2f} M samples/s; "
      f"{len(eng.V)} keys resident; loss {float(loss):.5f}")
