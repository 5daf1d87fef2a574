#!/usr/bin/env python3
"""Do HIP events recorded INSIDE a captured graph (event-record nodes) give elapsed times after a replay?
Decides whether bench.py can time k_apply_main when the whole step is one HIP graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from mindrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
V, D, n = 200000, 80, 100000
p = torch.randn(V, D, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
ids = torch.randint(0, V, (n,), device=dev, dtype=torch.int32)
g = torch.randn(n, D, device=dev)
plan = ops.sparse_plan(ids)
ops.sparse_lazy_adam_(p, m, v, plan, g)
torch.cuda.synchronize()
# eager reference
t = ops.KernelTimer(); t.arm(); ops.sparse_lazy_adam_(p, m, v, plan, g); torch.cuda.synchronize()
print("eager ms", t.ms())
s = torch.cuda.Stream()
tg = ops.KernelTimer()
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    gr.capture_begin()
    tg.arm()
    ops.sparse_lazy_adam_(p, m, v, plan, g)
    gr.capture_end()
torch.cuda.synchronize()
for i in range(3):
    gr.replay()
    torch.cuda.synchronize()
    try:
        print("graph replay", i, "ms", tg.ms())
    except Exception as e:
        print("graph replay", i, "FAILED:", e)
