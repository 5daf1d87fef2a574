#!/usr/bin/env python3
"""Where does the HOST spend a sharded step?  cProfile over 60 steps of the row-shard protocol on one GPU (world 1, RCCL with
itself).  The shard step is host-bound on this stack; this lists what to cut."""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29519")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = WideDeepConfig(vocab_size=20_000_000)
eng = WideDeepEngine(cfg, dev, rank=0, world=1, shard_protocol=True)
batches = [synthetic_batch(cfg, dev, "uniform", seed=1000 + i) for i in range(4)]
for i in range(8):
    eng.train_step(*batches[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(60):
    eng.train_step(*batches[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
