#!/usr/bin/env python3
"""The reference's ONE published benchmark configuration (benchmarks/wide_deep/default_config.yaml:2-31; table in
benchmarks/README.md:76-80: 267 558 samples/s on one V100-16GB, MindSpore 1.8) on one MI355X -- context, not `value`:
vocab 5.86 M, dim 16, 39 fields, batch 16000, MLP 7 x 1024 in fp32 (use_mixed_precision: False), Dropout on (dropout_flag: True,
keep 0.5), sparse: False (dense [V, D] gradients + the L2 term, nn.Adam / nn.FTRL over every row every step).  Also the same net
with sparse lookups (LazyAdam + FTRL on the touched rows), the mode the row-sharded engine scales.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch  # noqa: E402


def run(sparse, steps=30, warm=8):
    cfg = WideDeepConfig(vocab_size=5_860_000, emb_dim=16, field_size=39, batch_size=16000, deep_layer_dim=[1024] * 7, mlp_dtype="fp32",
                         dropout_flag=True, sparse=sparse, l2_coef=8e-5)
    dev = torch.device("cuda:0")
    eng = WideDeepEngine(cfg, dev)
    bs = [synthetic_batch(cfg, dev, "zipf", seed=1000 + i) for i in range(4)]
    for i in range(warm):
        eng.train_step(*bs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = eng.train_step(*bs[i % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert torch.isfinite(loss).item()
    flops = 6 * cfg.batch_size * (39 * 16 * 1024 + 6 * 1024 * 1024 + 1024)
    out = {"ms_per_step": round(ms, 4), "samples_per_s": round(cfg.batch_size / ms * 1e3, 1), "mlp_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
           "hand_written_fp32_net": bool(eng._f32net)}
    del eng
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    line = {"what": "reference's published Wide&Deep benchmark configuration (benchmarks/wide_deep/default_config.yaml), 1 x MI355X, synthetic "
                    "Criteo-like ids (Zipf + the 13 constant dense-field ids), fp32 net with Dropout, fp32 DenseLayers on three-part bf16 operands (ops.x3_*)",
            "published_reference": {"samples_per_s": 267558, "hardware": "1 x Tesla V100-SXM2-16GB, MindSpore 1.8 (benchmarks/README.md:4-26,76-78)"},
            "sparse_false_as_published": run(False), "sparse_true": run(True)}
    line["ratio_vs_published_v100"] = round(line["sparse_false_as_published"]["samples_per_s"] / 267558, 2)
    print(json.dumps(line))
