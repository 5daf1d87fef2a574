#!/usr/bin/env python3
"""Device-side cost of the shard protocol's own kernels at BASELINE shape (per rank: 16384 x 26 ids, 8 shards,
bf16 rows moved as 40 fp32 words): route (bucket by owner), route_rows, unroute.  HIP events, one GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


B, F, S = 16384, 26, 8
n = B * F
ids = torch.randint(0, 200_000_000, (B, F), dtype=torch.int32, device=dev)
wts = torch.rand((B, F), device=dev)
print(f"shard_route ({n} int32 ids -> {S} owners): {timeit(lambda: ops.shard_route(ids, S)):7.1f} us")
send_local, perm, counts = ops.shard_route(ids, S)
rows16 = torch.randn((n, 80), device=dev).to(torch.bfloat16)
w1 = wts.reshape(n, 1)
print(f"route_rows  [n,1]  weights            : {timeit(lambda: ops.shard_route_rows(w1, perm, None)):7.1f} us")
g32 = rows16.view(torch.float32)
print(f"route_rows  [n,40] bf16 gradients     : {timeit(lambda: ops.shard_route_rows(g32, perm, None)):7.1f} us  ({2 * n * 160 / 1e6:.0f} MB moved)")
print(f"unroute     [n,40] bf16 rows          : {timeit(lambda: ops.shard_unroute(g32, perm, None)):7.1f} us")
print(f"unroute     [n,1]  wide values        : {timeit(lambda: ops.shard_unroute(w1, perm, None)):7.1f} us")
recv = send_local
rw = w1.view(-1)
table = torch.empty((25_000_000, 80), device=dev)
print(f"owner gather bf16 (masked)            : {timeit(lambda: ops.gather_rows(table, recv, rw, out_dtype=torch.bfloat16)):7.1f} us")
tw = torch.empty((25_000_000, 1), device=dev)
print(f"owner gather wide [n,1]               : {timeit(lambda: ops.gather_rows(tw, recv, rw)):7.1f} us")
print(f"counts.tolist() (host sync)           : {timeit(lambda: counts.tolist()):7.1f} us")
