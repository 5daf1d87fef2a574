#!/usr/bin/env python3
"""Deep&Cross training step at BASELINE configs[2] (batch 16384, 39 fields x 30, 6 cross layers, fp32):
ms/step and samples/s from HIP events.  A parity-test configuration, not the judged bench line."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine  # noqa: E402

dev = torch.device("cuda:0")
cfg = DeepCrossConfig()
eng = DeepCrossEngine(cfg, dev)
B, F = cfg.batch_size, cfg.field_size
g = torch.Generator(device=dev).manual_seed(1000)
ids = torch.randint(0, cfg.vocab_size, (B, F), dtype=torch.int32, device=dev, generator=g)
wts = torch.rand((B, F), device=dev, generator=g)
label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    eng.train_step(ids, wts, label)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    loss = eng.train_step(ids, wts, label)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / steps
print(f"DCN step: {ms:.3f} ms  = {B / ms * 1e3 / 1e6:.2f} M samples/s   loss {float(loss):.5f}")
