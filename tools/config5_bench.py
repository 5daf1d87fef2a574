#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: DeepFM over MapParameter hash embeddings (int64 keys, dim 128, admission 2 /
eviction 100 steps), batch 16384 x 26 keys drawn Zipf-like from a 2^40 key space.  ms/step from HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.deepfm import DeepFMConfig, DeepFMHashEngine  # noqa: E402

dev = torch.device("cuda:0")
B, F, D = 16384, 26, 128
mlp_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = DeepFMConfig(data_emb_dim=D, data_field_size=F, batch_size=B, mlp_dtype=mlp_dtype)
eng = DeepFMHashEngine(cfg, dev, key_dtype=torch.int64, capacity=1 << 23, permit_filter_value=2, evict_filter_value=100)
rng = np.random.default_rng(7)


def batch():
    # per-slot Zipf over a private 2^35 range, scrambled into the 2^40 key space
    z = rng.zipf(1.1, size=(B, F)).astype(np.int64) % (1 << 35)
    keys = (z * 0x9E3779B1 + np.arange(F, dtype=np.int64)[None, :] * (1 << 35)) % (1 << 40)
    return (torch.from_numpy(keys).to(dev), torch.ones((B, F), device=dev),
            torch.from_numpy((rng.random((B, 1)) < 0.3).astype(np.float32)).to(dev))


batches = [batch() for _ in range(8)]
for i in range(4):
    eng.train_step(*batches[i % 8])
torch.cuda.synchronize()
steps = 20
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(steps):
    loss = eng.train_step(*batches[i % 8])
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / steps
print(f"configs[4] step (DeepFM + hash tables, int64 keys, D=128, permit 2, MLP {mlp_dtype}): {ms:.3f} ms = {B / ms * 1e3 / 1e6:.2f} M samples/s; "
      f"{len(eng.V)} keys resident; loss {float(loss):.5f}")

# ---- one JSON line with the roofline of the hash get and of the sparse apply at this shape (profiles/rNN_config5_line.json) ---------------
import json  # noqa: E402
from mindrec_amd import ops  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


keys, wts, label = batches[0]
N = B * F
state = {}


def get():                                   # HashEmbeddingLookup.construct on both tables: Unique -> MapTensorGet -> Gather back (+ the mask)
    d, rows_v, pos_v, rows_w, pos_w = eng._lookup(keys, insert=True)
    state.update(d=d, rows_v=rows_v)
    state["vx"] = ops.gather_rows(eng.V.values, pos_v.view(B, F), wts)
    state["lin"] = ops.wide_sum(eng.W.values, pos_w.view(B, F), wts)


get_ms = timed(get)
U = state["d"].U
g = torch.randn((N, D), device=dev) * 1e-3
plan = ops.group_by_inverse(state["d"])
plan.uniq_buf = eng.V.admitted_rows(state["rows_v"])
kw = eng._adam_kw(1.0 / cfg.loss_scale)


def apply():                                 # Unique'd row gradients -> segment-sum + LazyAdam on the admitted rows of the D = 128 table
    ops.sparse_lazy_adam_(eng.V.values, eng.V.slots["moment1"]["table"], eng.V.slots["moment2"]["table"], plan, g, wts.reshape(-1), **kw)


apply_ms = timed(apply)
s_ = 8                                       # int64 keys
get_b = N * s_ + U * D * 4 + N * D * 4 + N * (s_ + 8)                # deep lookup + the D = 1 table's
apply_b = N * s_ + N * D * 4 + U * 6 * D * 4
line = {"metric": "samples/sec DeepFM + MapParameter hash embedding (BASELINE configs[4], its one-GPU shape)", "value": round(B / ms * 1e3, 1),
        "unit": "samples/s", "n_gpus": 1, "ms_per_step": round(ms, 4), "higher_is_better": True, "dtype": "f32 rows, MLP " + mlp_dtype, "data": "synthetic",
        "config": {"workload": f"DeepFM over two MapParameters: int64 keys from a 2^40 space (Zipf per slot), batch {B} x {F} keys, dim {D}, "
                               f"permit_filter_value 2, evict_filter_value 100, capacity 2^23 rows; unique keys per batch {U} of {N}",
                   "unique_frac": round(U / N, 4), "keys_resident": len(eng.V)},
        "roofline": {"bound": "hbm", "kernel": "k_apply_main<4,long,UpdAdam,float> (+ k_apply_long): segment-sum + LazyAdam on the admitted rows",
                     "achieved": round(apply_b / (apply_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(apply_b / (apply_ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None, "algorithmic_bytes": apply_b,
                     "avg_ms": round(apply_ms, 5), "timing": "HIP events around 10 back-to-back calls (eager, outside the step)"},
        "roofline_get": {"bound": "hbm / latency", "kernels": "k_dedup_insert / k_dedup_rank (Unique), k_map_probe -> k_map_place -> k_map_finish x 2 tables, "
                                                               "k_gather_rows (D = 128) + k_wide_sum (D = 1)",
                         "achieved": round(get_b / (get_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(get_b / (get_ms * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes": get_b, "avg_ms": round(get_ms, 5),
                         "what": "resident keys (the batch was looked up before): Unique + index probe + row gather of both tables"},
        "cpu_baseline": None}
print(json.dumps(line))
