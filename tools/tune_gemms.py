#!/usr/bin/env python3
"""Generates mindrec_amd/tuned/tunableop_gfx950.csv: PyTorch TunableOp selections (hipBLASLt / rocBLAS
solution per GEMM shape) for the Wide&Deep and Deep&Cross MLPs at batch 16384.  Run on an MI355X:
    python tools/tune_gemms.py gpurun_out/tunableop_gfx950.csv
The engine loads the table with tuning DISABLED, so nothing is searched at run time; shapes that are
not in the table use the library default."""
import os
import sys

out = sys.argv[1]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "60"
os.environ["PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"] = "10"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.cuda.tunable as tn  # noqa: E402

tn.enable(True)
tn.tuning_enable(True)
tn.set_filename(out, insert_device_ordinal=False)
from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
for fields in (26, 39):
    # the fp32 MLP is the one that still runs on library GEMMs (the 16-bit MLPs run on mrec_dense_*)
    cfg = WideDeepConfig(vocab_size=1_000_000, field_size=fields, mlp_dtype="fp32")
    eng = WideDeepEngine(cfg, dev, tuned_gemms=False)
    b = synthetic_batch(cfg, dev, "uniform")
    for _ in range(3):
        eng.train_step(*b)
    torch.cuda.synchronize()
    del eng
cfg = DeepCrossConfig()
eng = DeepCrossEngine(cfg, dev)
wcfg = WideDeepConfig(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, field_size=cfg.field_size, batch_size=cfg.batch_size)
b = synthetic_batch(wcfg, dev, "uniform")
for _ in range(3):
    eng.train_step(*b)
torch.cuda.synchronize()
pass  # TunableOp flushes the table to PYTORCH_TUNABLEOP_FILENAME at process exit
print("wrote", out)
