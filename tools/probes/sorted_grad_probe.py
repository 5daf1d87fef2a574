#!/usr/bin/env python3
"""What would the sparse apply gain if the layer-0 input gradient were WRITTEN in the order of the step's inverted index
(row i of the buffer = the gradient of sorted entry i) instead of gathered by position?  Emulated without touching the
producer: the same plan with sorted_pos replaced by 0..n-1 and the gradient rows / per-position weights permuted beforehand
(untimed), so a window's 8 gradient rows are 1280 contiguous bytes.  (The per-sample wide gradient then is read at i / F: the
access pattern of a sorted copy, other values.)  Folded layout, the engine's kernels, each timed alone."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch  # noqa: E402
from tools.embed_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
V, D, B = 200_000_000, 80, 16384
ld = -(-(3 * D + 4) // 32) * 32
st = torch.zeros((V, ld), dtype=torch.float32, device=dev)
p, m, v = st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4]
w, wa = st[:, D:D + 1], st[:, D + 1:D + 2]
ops.fill_normal_(p, 1000, 0.01); ops.fill_normal_(w, 1001, 0.01); wa.fill_(1.0)
for dist, F in (("uniform", 26), ("zipf", 39), ("zipf", 26), ("uniform", 39)):
    cfg = WideDeepConfig(vocab_size=V, emb_dim=D, field_size=F, batch_size=B)
    nb = 4
    batches = [synthetic_batch(cfg, dev, dist, seed=1000 + i) for i in range(nb)]
    N = B * F
    g = torch.randn((N, D), device=dev).to(torch.float16)
    gw = torch.randn(B, device=dev)
    plans = [ops.sparse_plan(b[0]) for b in batches]
    splans, sg, swts = [], [], []
    for pl, b in zip(plans, batches):
        sp = copy.copy(pl)
        order = pl.sorted_pos[:N].long()
        sp.sorted_pos = torch.arange(N, dtype=torch.int32, device=dev)
        splans.append(sp)
        sg.append(g[order].contiguous())
        swts.append(b[1].reshape(-1)[order].contiguous().view(B, F))
    kw = dict(beta1_power=0.5, beta2_power=0.9, grad_scale=1 / 1024)
    t_a = timeit(lambda i: ops.sparse_lazy_adam_wide_(p, m, v, plans[i % nb], g, batches[i % nb][1], gw, F, D, **kw), 30)
    t_s = timeit(lambda i: ops.sparse_lazy_adam_wide_(p, m, v, splans[i % nb], sg[i % nb], swts[i % nb], gw, F, D, **kw), 30)
    print(f"{dist:8s} x {F} fields  U/N {plans[0].U / N:.3f}:  apply, gradient rows gathered by position {t_a[0] * 1e3:7.1f} us   "
          f"gradient rows in index order {t_s[0] * 1e3:7.1f} us")
