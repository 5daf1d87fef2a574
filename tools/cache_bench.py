#!/usr/bin/env python3
"""Host-DRAM feature-cache tier on the measurement bar: a [V, 3D] table whose home is pinned host
memory, driven through a device cache of C rows with Zipf-distributed ids at BASELINE batch shape.  Reports the hit
rate and the time of one lookup + sparse-apply round (prepare = Unique + probe + evict/write-back + miss fetch or
first-touch init; then the same gather / LazyAdam kernels as the resident table)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.feature_cache import HostBackedTable  # noqa: E402

dev = torch.device("cuda:0")
V, D, C, B, F = 4_000_000, 80, 1_000_000, 16384, 26
t0 = time.perf_counter()
hb = HostBackedTable(V, D, C, dev, seed=1000, sigma=0.01)
print(f"table [{V}, {3 * D}] fp32 = {V * 3 * D * 4 / 1e9:.1f} GB pinned on the host, device cache {C} rows = {C * 3 * D * 4 / 1e9:.2f} GB "
      f"(set up in {time.perf_counter() - t0:.1f} s)")
rng = np.random.default_rng(3)


def batch(alpha):
    slot = V // F
    z = np.minimum(rng.zipf(alpha, size=(B, F)) - 1, slot - 1)
    return torch.from_numpy((z + slot * np.arange(F)[None, :]).astype(np.int64)).to(dev)


for alpha in (1.05, 1.2):
    hb.stats = {k: 0 for k in hb.stats}
    wts = torch.ones((B, F), device=dev)
    g = torch.randn((B * F, D), device=dev)
    pre = [batch(alpha) for _ in range(28)]
    for i in range(8):                     # warm the cache
        ids = pre[i]
        plan, rows_pos = hb.prepare(ids)
        ops.sparse_lazy_adam_(hb.p, hb.slots[0], hb.slots[1], plan, g, wts)
    hb.stats = {k: 0 for k in hb.stats}
    torch.cuda.synchronize()
    steps = 20
    # prepare alone (device time; the host only queues): a synchronising call inside it would raise
    torch.cuda.set_sync_debug_mode("error")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(steps // 2):
        plan, rows_pos = hb.prepare(pre[8 + i])
    ev[1].record()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    tp = ev[0].elapsed_time(ev[1]) * 1e-3 / (steps // 2) * steps
    # the round: prepare + lookup + sparse apply, queued back to back
    t0 = time.perf_counter()
    for i in range(steps // 2, steps):
        ids = pre[8 + i]
        torch.cuda.set_sync_debug_mode("error")
        plan, rows_pos = hb.prepare(ids)
        torch.cuda.set_sync_debug_mode("default")
        emb = hb.gather(rows_pos, wts.reshape(-1))
        ops.sparse_lazy_adam_(hb.p, hb.slots[0], hb.slots[1], plan, g, wts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps - steps // 2)
    s = hb.stats
    tot = s["hits"] + s["misses"]
    print(f"Zipf({alpha}): unique-id hit rate {s['hits'] / max(tot, 1) * 100:5.1f} %  misses/step {s['misses'] / steps:8.0f} "
          f"(first touch {s['first_touch'] / steps:6.0f}, evictions {s['evictions'] / steps:6.0f})   lookup+apply round {dt * 1e3:7.2f} ms "
          f"= {B / dt / 1e6:5.2f} M samples/s; prepare alone {tp / steps * 1e3:6.2f} ms; host syncs inside prepare: 0 (sync debug mode)")
