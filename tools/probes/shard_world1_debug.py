"""Debug probe: the RCCL world-1 sharded engine at the test's small shape, with progress prints and a traceback dump if it stalls."""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "90")), exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch

mlp_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = WideDeepConfig(vocab_size=30_011, emb_dim=80, field_size=26, batch_size=512, deep_layer_dim=[64, 32], mlp_dtype=mlp_dtype,
                     graphs=os.environ.get("GRAPHS", "step"))
eng = WideDeepEngine(cfg, dev, rank=0, world=1, shard_protocol=True)
t0 = time.time()
for s in range(6):
    ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=50 + s)
    loss = eng.train_step(ids, wts, label)
    print(f"step {s}: issued at {time.time() - t0:.2f}s", flush=True)
    print(f"step {s}: loss {float(loss):.5f} graph={eng._step_graph is not None} at {time.time() - t0:.2f}s", flush=True)
bs = [synthetic_batch(cfg, dev, "zipf", seed=56 + s) for s in range(2)]
out = eng.train_steps(bs)
print("sink issued", flush=True)
print("sink losses", [float(x) for x in out], "sink graphs", {k: v is not None for k, v in eng._sink_graphs.items()}, flush=True)
print("overflow", eng.shard_overflow(), flush=True)
dist.barrier()
eng.release_graphs()
dist.destroy_process_group()
print("done", flush=True)
