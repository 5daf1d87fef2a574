"""Which eager RCCL work can kill the process when a HIP-graph capture starts before torch's watchdog thread has retired it?
On HIP, querying an event fails (and invalidates the capture) while the stream it was recorded on is capturing
(tools/probes/captured_event_probe.py); ProcessGroupNCCL's watchdog queries the end event of every eager work.
  argv[1] = sync  : eager all_reduce(async_op=False) on the current stream, then at once a capture holding an async all_reduce
            async : eager all_reduce(async_op=True) (the group's internal stream), then the same capture
            own   : eager all_reduce(async_op=False) under an own side stream, then the same capture
The capture is held open 0.5 s so that the watchdog (polls every 100 ms) certainly looks at the eager work meanwhile."""
import os
import sys
import time

import torch
import torch.distributed as dist

mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 20, device=dev)
u = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
time.sleep(0.5)                      # the warm-up work is retired
side = torch.cuda.Stream()
if mode == "sync":
    dist.all_reduce(t)
elif mode == "async":
    w = dist.all_reduce(t, async_op=True)
    w.wait()
else:
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(t)
    torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    w = dist.all_reduce(u, async_op=True)
    time.sleep(0.5)
    w.wait()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print(f"{mode}: capture and replay fine, u[0] = {float(u[0])}", flush=True)
del g
dist.destroy_process_group()
