#!/usr/bin/env python3
"""Kernel-level microbenchmark of the embedding path (no MLP): HIP-event time and algorithmic GB/s
of plan / gather / wide_sum / sparse LazyAdam / sparse FTRL at BASELINE config-2 shapes.
Used for tuning; bench.py is the judged measurement."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, embedding_bytes, synthetic_batch  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    evs = []
    for i in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=200_000_000)
    ap.add_argument("--dim", type=int, default=80)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--layout", default="split", choices=["split", "fused", "folded"])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    V, D = a.vocab, a.dim
    cfg = WideDeepConfig(vocab_size=V, emb_dim=D, field_size=a.fields, batch_size=a.batch)
    if a.layout == "folded":
        return folded(a, dev, cfg)
    if a.layout == "fused":      # p|m|v of one row contiguous (960 B), wide w|accum|linear|pad contiguous (16 B)
        st = torch.empty((V, 3 * D), dtype=torch.float32, device=dev)
        p, m, v = st[:, :D], st[:, D:2 * D], st[:, 2 * D:]
        ws = torch.empty((V, 4), dtype=torch.float32, device=dev)
        w, wa, wl = ws[:, 0:1], ws[:, 1:2], ws[:, 2:3]
        m.zero_(); v.zero_(); wa.fill_(1.0); wl.zero_()
    else:
        p = torch.empty((V, D), dtype=torch.float32, device=dev)
        m = torch.zeros_like(p); v = torch.zeros_like(p)
        w = torch.empty((V, 1), dtype=torch.float32, device=dev)
        wa = torch.ones_like(w); wl = torch.zeros_like(w)
    ops.fill_normal_(p, 1000, 0.01)
    ops.fill_normal_(w, 1001, 0.01)
    nb = 4
    batches = [synthetic_batch(cfg, dev, a.dist, seed=1000 + i) for i in range(nb)]
    N = a.batch * a.fields
    g = torch.randn((N, D), device=dev)
    gw = torch.randn((N, 1), device=dev)
    plans = [ops.sparse_plan(b[0]) for b in batches]
    U = plans[0].U
    by = embedding_bytes(N, U, D)
    res = {}
    res["plan"] = timeit(lambda i: ops.sparse_plan(batches[i % nb][0]), a.iters)
    out = torch.empty((N, D), dtype=torch.float32, device=dev)
    res["lookup"] = timeit(lambda i: ops.gather_rows(p, batches[i % nb][0], batches[i % nb][1], out=out), a.iters)
    res["wide_lookup"] = timeit(lambda i: ops.wide_sum(w, batches[i % nb][0], batches[i % nb][1]), a.iters)
    res["apply_deep"] = timeit(lambda i: ops.sparse_lazy_adam_(p, m, v, plans[i % nb], g, batches[i % nb][1],
                                                               beta1_power=0.5, beta2_power=0.9, grad_scale=1 / 1024), a.iters)
    res["apply_wide"] = timeit(lambda i: ops.sparse_ftrl_(w, wa, wl, plans[i % nb], gw, None, grad_scale=1 / 1024), a.iters)
    print(f"[{a.tag}] V={V} D={D} N={N} U/N={U / N:.4f} dist={a.dist} layout={a.layout}")
    tot = 0.0
    for k, (med, mn) in res.items():
        gb = by.get(k, 0) / (med * 1e-3) / 1e9 if k in by else float("nan")
        tot += med
        print(f"  {k:12s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us   {gb:8.1f} GB/s  ({gb / 80:.1f}% of 8 TB/s)")
    emb_bytes = by["lookup"] + by["apply_deep"]
    t = res["lookup"][0] + res["apply_deep"][0]
    print(f"  EmbeddingLookup+sparse-apply: {emb_bytes / 1e6:.1f} MB in {t * 1e3:.1f} us = {emb_bytes / (t * 1e-3) / 1e9:.1f} GB/s "
          f"({emb_bytes / (t * 1e-3) / 8e12 * 100:.1f}% of 8 TB/s);  whole embedding path {tot * 1e3:.1f} us")


def folded(a, dev, cfg):
    """The engine's one-GPU layout: 1-KB fused rows [p | w accum linear pad | m | v], 16-bit rows out / row gradients in, the
    wide branch riding the deep kernels -- lookup + apply of BOTH tables = two kernels, each timed alone (no plan, no GEMM
    beside them)."""
    V, D, F, B = a.vocab, a.dim, a.fields, a.batch
    ld = -(-(3 * D + 4) // 32) * 32
    st = torch.zeros((V, ld), dtype=torch.float32, device=dev)
    p, m, v = st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4]
    w, wa = st[:, D:D + 1], st[:, D + 1:D + 2]
    ops.fill_normal_(p, 1000, 0.01); ops.fill_normal_(w, 1001, 0.01); wa.fill_(1.0)
    nb = 4
    batches = [synthetic_batch(cfg, dev, a.dist, seed=1000 + i) for i in range(nb)]
    N = B * F
    g = torch.randn((N, D), device=dev).to(torch.bfloat16)
    gw = torch.randn(B, device=dev)
    plans = [ops.sparse_plan(b[0]) for b in batches]
    U = plans[0].U
    by = embedding_bytes(N, U, D, act_bytes=2)
    lookup_b = by["lookup"] + U * 4 + N * 4
    apply_b = by["apply_deep"] + U * 24 + B * 4
    out = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
    t_l = timeit(lambda i: ops.gather_rows_wide(p, batches[i % nb][0], batches[i % nb][1], D, out=out), a.iters)
    t_a = timeit(lambda i: ops.sparse_lazy_adam_wide_(p, m, v, plans[i % nb], g, batches[i % nb][1], gw, F, D, beta1_power=0.5,
                                                      beta2_power=0.9, grad_scale=1 / 1024), a.iters)
    t_p = timeit(lambda i: ops.sparse_plan(batches[i % nb][0]), a.iters)
    print(f"[{a.tag}] V={V} D={D} N={N} U/N={U / N:.4f} dist={a.dist} layout=folded (1-KB fused rows, bf16 rows / gradients)")
    for name, (med, mn), nbytes in (("lookup deep+wide", t_l, lookup_b), ("apply deep+wide", t_a, apply_b)):
        print(f"  {name:18s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / (med * 1e-3) / 1e9:8.1f} GB/s "
              f"({nbytes / (med * 1e-3) / 8e12 * 100:.1f}% of 8 TB/s)")
    print(f"  {'plan':18s} median {t_p[0] * 1e3:8.1f} us  min {t_p[1] * 1e3:8.1f} us")
    t = t_l[0] + t_a[0]
    print(f"  EmbeddingLookup + sparse apply, deep AND wide tables, each kernel alone: {(lookup_b + apply_b) / 1e6:.1f} MB in {t * 1e3:.1f} us = "
          f"{(lookup_b + apply_b) / (t * 1e-3) / 1e9:.1f} GB/s ({(lookup_b + apply_b) / (t * 1e-3) / 8e12 * 100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
