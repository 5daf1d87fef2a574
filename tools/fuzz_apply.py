#!/usr/bin/env python3
"""Randomised parity sweep of the sparse path against the oracle: random n, D, vocabulary size, id distribution, key
dtype, gradient dtype, row scale on/off.  Unique / inverse must match exactly; LazyAdam / FTRL / segment-sum rows are
held to 2e-5 row-relative (runs that stay inside a window are in fact bit-exact, see tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindrec_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150


def rel(a, b):
    # row-relative, with a floor at a twentieth of the typical row: a narrow row (D = 1) whose handful of gradients nearly
    # cancel has no magnitude of its own to be relative to, and a different (fixed) order of the same adds shows there
    rowmax = np.abs(b).max(axis=1)
    den = np.maximum(rowmax, max(0.05 * float(np.median(rowmax)), 1e-20))
    return float((np.abs(a - b).max(axis=1) / den).max())


worst = 0.0
for c in range(cases):
    D = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 30, 40, 80, 128, 130, 256, 260]))
    n = int(rng.choice([1, 7, 63, 64, 65, 1000, 4097, 20000, 70001]))
    V = int(rng.choice([1, 5, 100, 10_000, 1_000_000]))
    kind = rng.choice(["uniform", "zipf", "hot", "same"])
    dt = rng.choice([np.int32, np.int64])
    if kind == "uniform":
        ids = rng.integers(0, V, size=n)
    elif kind == "zipf":
        ids = np.minimum(rng.zipf(1.2, size=n) - 1, V - 1)
    elif kind == "hot":
        ids = rng.integers(0, V, size=n); ids[rng.random(n) < 0.5] = V // 2
    else:
        ids = np.full(n, V - 1)
    ids = ids.astype(dt)
    heavy = bool(np.bincount(ids.astype(np.int64)).max() > 64)      # long runs: sums of many terms in a different order
    use_scale = bool(rng.random() < 0.5)
    wts = rng.random(n).astype(np.float32) if use_scale else None
    g = rng.standard_normal((n, D)).astype(np.float32)
    tid = torch.from_numpy(ids).to(dev)
    plan = ops.sparse_plan(tid)
    u, inv = O.unique(ids)
    assert plan.U == u.size and np.array_equal(plan.uniq.cpu().numpy(), u) and np.array_equal(plan.inv.cpu().numpy(), inv), ("unique", c, D, n, V, kind)
    p = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    m = (rng.standard_normal((V, D)) * 0.001).astype(np.float32)
    v = (rng.random((V, D)) * 1e-4).astype(np.float32)
    tp, tm, tv = (torch.from_numpy(x.copy()).to(dev) for x in (p, m, v))
    tw = torch.from_numpy(wts).to(dev) if use_scale else None
    bf = bool(rng.random() < 0.3) and D % 2 == 0
    tg = torch.from_numpy(g).to(dev)
    if bf:
        tg = tg.to(torch.bfloat16)
        g = tg.float().cpu().numpy()
    ops.sparse_lazy_adam_(tp, tm, tv, plan, tg, tw, beta1_power=0.81, beta2_power=0.998, grad_scale=1 / 64)
    O.sparse_lazy_adam(p, m, v, ids, g, wts, b1_pow=0.81, b2_pow=0.998, grad_scale=1 / 64)
    e = max(rel(tp.cpu().numpy(), p), rel(tm.cpu().numpy(), m))
    tol = 1e-3 if heavy else 2e-5                               # thousands of copies of one id: any-order fp32 sums
    assert e <= tol, ("adam", c, D, n, V, kind, bf, e)
    worst = max(worst, 0.0 if heavy else e)
    # FTRL on a D = 1 slice of the same ids
    w1 = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32); a1 = np.ones((V, 1), np.float32); l1 = np.zeros((V, 1), np.float32)
    tw1, ta1, tl1 = (torch.from_numpy(x.copy()).to(dev) for x in (w1, a1, l1))
    g1 = rng.standard_normal((n, 1)).astype(np.float32)
    ops.sparse_ftrl_(tw1, ta1, tl1, plan, torch.from_numpy(g1).to(dev), tw, grad_scale=1 / 64)
    O.sparse_ftrl(w1, a1, l1, ids, g1, wts, grad_scale=1 / 64)
    assert np.allclose(tw1.cpu().numpy(), w1, rtol=1e-3, atol=1e-6) and np.allclose(ta1.cpu().numpy(), a1, rtol=1e-3, atol=1e-6), ("ftrl", c, n, V, kind)
    # segment sum
    s = ops.segment_sum(plan, torch.from_numpy(g).to(dev) if not bf else tg.float(), tw).cpu().numpy()[: u.size]
    rs = O.segment_sum(g * (wts[:, None] if use_scale else 1.0), inv, u.size)
    assert np.allclose(s, rs, rtol=1e-4, atol=2e-3 if heavy else 1e-5), ("segsum", c, D, n, V, kind)
print(f"{cases} random cases passed; worst row-relative error where no id has more than 64 copies: {worst:.2e}")
