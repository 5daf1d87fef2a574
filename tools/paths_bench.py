#!/usr/bin/env python3
"""HIP-event timings + algorithmic GB/s of the secondary paths at BASELINE shapes:
DCN cross layers (configs[2]), MapParameter lookup/apply (configs[4] shape: int64 keys, D=128), DeepFM FM term."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops  # noqa: E402
from mindrec_amd.experimental import MapParameter  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2] * 1e-3


def show(name, sec, nbytes):
    print(f"{name:44s} {sec * 1e6:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / sec / 1e9:8.1f} GB/s ({nbytes / sec / 8e12 * 100:5.1f}% of 8 TB/s)")


B = 16384
# ---- DCN cross layers: x [B, 1170], 6 layers
D, L = 1170, 6
x0 = torch.randn(B, D, device=dev) * 0.5
w = torch.randn(L, D, device=dev) / D ** 0.5
b = torch.randn(L, D, device=dev) * 0.1
dy = torch.randn(B, D, device=dev)
show("cross_layers fwd (6 layers, one pass)", timeit(lambda: ops.cross_layers(x0, w, b)), 2 * B * D * 4)
show("cross_layers bwd (dx0, dw, db)", timeit(lambda: ops.cross_layers_bwd(x0, w, b, dy)), 3 * B * D * 4)
# ---- DeepFM FM term: vx [B, 39, 80]
vx = torch.randn(B, 39, 80, device=dev) * 0.1
fm, cs = ops.fm_forward(vx)
g = torch.zeros_like(vx)
dout = torch.randn(B, device=dev)
show("fm_forward [16384,39,80]", timeit(lambda: ops.fm_forward(vx)), B * 39 * 80 * 4 + B * 80 * 4)
show("fm_backward (g += ...)", timeit(lambda: ops.fm_backward_(g, vx, cs, dout)), 3 * B * 39 * 80 * 4)
# ---- MapParameter at config-5 shape: int64 keys, D = 128, 16384 x 26 ids
F, Dm = 26, 128
N = B * F
m = MapParameter(key_dtype=torch.int64, value_shape=(Dm,), capacity=1 << 23, device=dev)
keys = [torch.randint(0, 2 ** 40, (B, F), dtype=torch.int64, device=dev) for _ in range(3)]
warm_keys = torch.randint(0, 2 ** 22, (B, F), dtype=torch.int64, device=dev)          # recurring key set
m.get(warm_keys)
show("MapParameter.get, resident keys (probe+gather)", timeit(lambda: m.get(warm_keys)), N * 8 + 2 * N * Dm * 4)
i = [0]
def fresh():
    i[0] += 1
    return m.get(torch.randint(0, 2 ** 40, (B, F), dtype=torch.int64, device=dev))
show("MapParameter.get, all-new keys (insert+init)", timeit(fresh, iters=6, warm=1), N * 8 + 2 * N * Dm * 4)
