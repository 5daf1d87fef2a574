#!/usr/bin/env python3
"""DeepFM training step at the reference's default shape (models/deepfm/default_config.yaml: vocab 184 965,
dim 80, 39 fields, batch 16 000): ms/step from HIP events, for both MLP dtypes.  Informational."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.deepfm import DeepFMConfig, DeepFMEngine  # noqa: E402

dev = torch.device("cuda:0")
for dt in (sys.argv[1:] or ("fp16", "bf16", "fp32")):     # fp16: the reference's convert_dtype (hand-written MFMA net); fp32: the fp32 kernels (three-part bf16 operands, or the fp32-input matrix instruction)
    cfg = DeepFMConfig(mlp_dtype=dt)
    eng = DeepFMEngine(cfg, dev)
    B, F = cfg.batch_size, cfg.data_field_size
    g = torch.Generator(device=dev).manual_seed(1000)
    ids = torch.randint(0, cfg.data_vocab_size, (B, F), dtype=torch.int32, device=dev, generator=g)
    wts = torch.rand((B, F), device=dev, generator=g)
    label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
    for _ in range(12):                    # (the MLP's HIP graphs are captured in the engine's third step; clocks settle)
        eng.train_step(ids, wts, label)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    runs = []
    for _ in range(3):                     # median of three blocks: a 20-step sample right behind the set-up read 3.8-4.5 ms for 3.34
        a.record()
        for _ in range(30):
            loss = eng.train_step(ids, wts, label)
        b.record()
        torch.cuda.synchronize()
        runs.append(a.elapsed_time(b) / 30)
    ms = sorted(runs)[1]
    print(f"DeepFM step (MLP {dt}, {'hand-written MFMA net' if eng._mfma else (f'hand-written fp32 net, MatMuls: {cfg.fp32_matmul}' if getattr(eng, '_f32net', False) else 'torch / library GEMMs')}): {ms:.3f} ms = {B / ms * 1e3 / 1e6:.2f} M samples/s   loss {float(loss):.5f}")
    del eng
