"""Can RCCL collectives (world 1, backend "nccl") be captured into a HIP graph by torch.cuda.graph on this stack, and what do
they cost eager vs replayed?  One process, one GPU: every collective talks to itself."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

n = 425984
a = torch.randn(n, 44, device=dev)
b = torch.empty_like(a)
g = torch.randn(2_900_000, device=dev)
x = torch.randn(1 << 20, device=dev)


def step():
    y = x * 2.0
    dist.all_to_all_single(b, a)
    z = b[:1024].sum() + y[:1]
    dist.all_reduce(g)
    return z + g[:1]


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("eager  ms/step:", (time.perf_counter() - t0) / 20 * 1e3, flush=True)

try:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        out = step()
    torch.cuda.synchronize()
    b.zero_()
    gr.replay()
    torch.cuda.synchronize()
    print("capture ok; a2a result correct after replay:", bool(torch.equal(a, b)), flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    print("replay ms/step:", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
except Exception as e:       # noqa: BLE001
    print("capture FAILED:", type(e).__name__, str(e)[:500], flush=True)

# async all-reduce handle inside a capture
try:
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2, capture_error_mode="thread_local"):
        w = dist.all_reduce(g, async_op=True)
        y = x * 3.0
        w.wait()
        o = y[:1] + g[:1]
    gr2.replay()
    torch.cuda.synchronize()
    print("async all_reduce capture ok", flush=True)
except Exception as e:       # noqa: BLE001
    print("async capture FAILED:", type(e).__name__, str(e)[:500], flush=True)
dist.barrier()
dist.destroy_process_group()
