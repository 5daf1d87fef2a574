#!/usr/bin/env python3
"""TunableOp selections for the GEMM shapes of the DeepFM engines (reference default shape and BASELINE configs[4]);
writes a table of its own -- merge the new lines into mindrec_amd/tuned/tunableop_gfx950.csv (tools/merge_tuned.py)."""
import os
import sys

out = sys.argv[1]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "60"
os.environ["PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"] = "10"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.cuda.tunable as tn  # noqa: E402

tn.enable(True)
tn.tuning_enable(True)
tn.set_filename(out, insert_device_ordinal=False)
import mindrec_amd.wide_deep as wd  # noqa: E402
wd.enable_tuned_gemms = lambda: False            # do not load the shipped table: tune from scratch here
from mindrec_amd.deepfm import DeepFMConfig, DeepFMEngine, DeepFMHashEngine  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for dt in ("bf16", "fp32"):
    cfg = DeepFMConfig(mlp_dtype=dt)
    eng = DeepFMEngine(cfg, dev)
    B, F = cfg.batch_size, cfg.data_field_size
    ids = torch.randint(0, cfg.data_vocab_size, (B, F), dtype=torch.int32, device=dev, generator=g)
    wts = torch.rand((B, F), device=dev, generator=g)
    label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
    for _ in range(3):
        eng.train_step(ids, wts, label)
    torch.cuda.synchronize()
    del eng
B, F, D = 16384, 26, 128
for dt in ("bf16", "fp32"):
    cfg = DeepFMConfig(data_emb_dim=D, data_field_size=F, batch_size=B, mlp_dtype=dt)
    eng = DeepFMHashEngine(cfg, dev, key_dtype=torch.int64, capacity=1 << 21)
    keys = torch.randint(0, 2 ** 40, (B, F), dtype=torch.int64, device=dev, generator=g) % (1 << 20)
    wts = torch.ones((B, F), device=dev)
    label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
    for _ in range(3):
        eng.train_step(keys, wts, label)
    torch.cuda.synchronize()
    del eng
print("wrote", out)
