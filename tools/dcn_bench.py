#!/usr/bin/env python3
"""Deep&Cross training step at BASELINE configs[2] (batch 16384, 39 fields x 30, 6 cross layers, fp32):
ms/step and samples/s from HIP events.  A parity-test configuration, not the judged bench line."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine  # noqa: E402

dev = torch.device("cuda:0")
cfg = DeepCrossConfig(fp32_matmul=sys.argv[2] if len(sys.argv) > 2 else "x3")      # argv: [steps [x3 | exact]]
eng = DeepCrossEngine(cfg, dev)
B, F = cfg.batch_size, cfg.field_size
g = torch.Generator(device=dev).manual_seed(1000)
ids = torch.randint(0, cfg.vocab_size, (B, F), dtype=torch.int32, device=dev, generator=g)
wts = torch.rand((B, F), device=dev, generator=g)
label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(10):
    eng.train_step(ids, wts, label)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
runs = []
for _ in range(3):
    a.record()
    for _ in range(steps):
        loss = eng.train_step(ids, wts, label)
    b.record()
    torch.cuda.synchronize()
    runs.append(a.elapsed_time(b) / steps)
ms = sorted(runs)[1]
print(f"DCN step (fp32 net, MatMuls: {cfg.fp32_matmul}; {'one HIP graph' if eng._native and eng._graph is not None else 'kernel by kernel'}): "
      f"{ms:.3f} ms (median of 3 x {steps} steps: {', '.join(f'{r:.3f}' for r in runs)}) = {B / ms * 1e3 / 1e6:.2f} M samples/s   loss {float(loss):.5f}")
if eng._native:
    # the three DenseLayer GEMM kinds alone (HIP events, 20 calls each)
    from mindrec_amd import ops
    X, h1 = F * cfg.emb_dim, cfg.deep_layer_dim[0]
    x = torch.randn((B, X), device=dev); w = torch.randn((X, h1), device=dev) * 0.02; dy = torch.randn((B, h1), device=dev)
    slabs = torch.empty((ops.dense32_bwd_weight_slabs(B, X, h1), X, h1), device=dev)
    for name, fn, fl in (("forward  [16384,1170]x[1170,1024]", lambda: ops.dense32_fwd(x, w, None, relu=True), 2 * B * X * h1),
                         ("dgrad    [16384,1024]x[1024,1170]", lambda: ops.dense32_bwd_input(dy, w), 2 * B * X * h1),
                         (f"wgrad    [1170,16384]x[16384,1024] in {slabs.shape[0]} slabs", lambda: ops.dense32_bwd_weight(x, dy, slabs), 2 * B * X * h1)):
        for _ in range(3):
            fn()
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20
        print(f"  fp32 MFMA DenseLayer {name}: {t * 1e3:.1f} us = {fl / t / 1e9:.1f} TFLOP/s ({fl / t / 1e9 / 157.3 * 100:.0f} % of the 157.3 TF fp32 matrix peak)")
