#!/usr/bin/env python3
"""k_apply_main / k_apply_long / the lookup at tiny batches (a handful of workgroups on an idle chip) up to the benchmark batch,
for a rocprofv3 --kernel-trace run: how long is ONE wave's chain of dependent loads when nothing else is on the chip?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
V, D = 20_000_000, 80
ld = -(-(3 * D + 4) // 32) * 32
st = torch.zeros((V, ld), dtype=torch.float32, device=dev)
p, m, v = st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4]
dist, F = sys.argv[1], int(sys.argv[2])
for B in [int(x) for x in sys.argv[3].split(",")]:
    cfg = WideDeepConfig(vocab_size=V, emb_dim=D, field_size=F, batch_size=B)
    ids, wts, _ = synthetic_batch(cfg, dev, dist, seed=1000)
    N = B * F
    g = torch.randn((N, D), device=dev).to(torch.float16)
    gw = torch.randn(B, device=dev)
    plan = ops.sparse_plan(ids)
    out = torch.empty((N, D), dtype=torch.float16, device=dev)
    for _ in range(30):
        ops.gather_rows_wide(p, ids, wts, D, out=out)
        ops.sparse_lazy_adam_wide_(p, m, v, plan, g, wts, gw, F, D, beta1_power=0.5, beta2_power=0.9, grad_scale=1 / 1024)
    torch.cuda.synchronize()
