"""The RCCL code path on hardware: the row-sharded engine with backend "nccl" (= RCCL), world_size 1 -- every
all_to_all_single talks to itself, the dense all-reduce runs asynchronously with work.wait(), and from the fourth step on
the WHOLE step -- routing kernels, the three all-to-alls, the MLP, the sparse apply, the all-reduce, the dense Adam -- replays
as ONE HIP graph with the RCCL kernels inside it.  With one shard the protocol is a pure permutation, so the step must
reproduce the one-GPU engine (ordinary path, no collectives) on the same
batches: losses and the tables to the tolerance of the two apply orders, dense parameters closely.
Runs in a child process (its own process group), one process on the card."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, port, out_dir, mlp_dtype, uniq=0.0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=30_011, emb_dim=80, field_size=26, batch_size=512, deep_layer_dim=[64, 32],
                         mlp_dtype=mlp_dtype, shard_unique_factor=uniq)
    eng = WideDeepEngine(cfg, dev, rank=0, world=1, shard_protocol=True)
    assert eng._sharded and eng.comm.__class__.__name__ == "_DirectComm"
    losses = []
    for s in range(6):                     # the whole-step graph is captured on step 4 and replayed from then on
        ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=50 + s)
        losses.append(float(eng.train_step(ids, wts, label)))
    # a sink of two more steps: ONE graph of two whole sharded steps (train_steps)
    bs = [synthetic_batch(cfg, dev, "zipf", seed=56 + s) for s in range(2)]
    losses += [float(x) for x in eng.train_steps(bs)]
    torch.cuda.synchronize()
    if mlp_dtype != "fp32":
        assert eng._step_graph is not None, "the sharded step did not become one HIP graph"
        assert any(v is not None for v in eng._sink_graphs.values()), "the sharded sink did not become one HIP graph"
    assert eng.shard_overflow() == 0
    np.savez(os.path.join(out_dir, "rccl.npz"), deep=eng.deep.cpu().numpy(), wide=eng.wide.cpu().numpy(),
             deep_m=eng.deep_m.cpu().numpy(), dense=eng.dense_flat.detach().cpu().numpy(), losses=np.array(losses))
    dist.barrier()
    eng.release_graphs()                   # graphs holding RCCL kernels must go before the process group does
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mlp_dtype,uniq", [("bf16", 0.0), ("fp16", 0.0), ("fp32", 0.0), ("fp16", 1.0)])      # (uniq: UNIQUE ids on the wire)
def test_sharded_engine_over_rccl_world1_matches_one_gpu_engine(dev, tmp_path, mlp_dtype, uniq):
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    mp.spawn(_worker, args=(_free_port(), str(tmp_path), mlp_dtype, uniq), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    cfg = WideDeepConfig(vocab_size=30_011, emb_dim=80, field_size=26, batch_size=512, deep_layer_dim=[64, 32], mlp_dtype=mlp_dtype)
    eng = WideDeepEngine(cfg, dev)
    losses = []
    for s in range(8):
        ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=50 + s)
        losses.append(float(eng.train_step(ids, wts, label)))
    assert np.allclose(r["losses"], losses, rtol=1e-5 if mlp_dtype == "fp32" else 1e-3)
    for name in ("deep", "deep_m", "wide"):
        full = getattr(eng, name).cpu().numpy()
        scale = np.abs(full).max()
        assert np.abs(r[name] - full).max() <= (2e-5 if mlp_dtype == "fp32" else 2e-2) * scale, name
        assert np.array_equal((r[name] != 0).any(axis=1), (full != 0).any(axis=1)) or name == "wide"
    ref = eng.dense_flat.detach().cpu().numpy()
    if mlp_dtype == "fp32":
        assert np.allclose(r["dense"], ref, rtol=1e-4, atol=1e-7)
    else:
        diff = np.abs(r["dense"] - ref)
        assert diff.max() <= 2.0 * cfg.adam_lr * 8 and np.mean(diff <= 5e-2 * np.abs(ref) + 1e-5) >= 0.99


def _race_worker(rank, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import time
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from mindrec_amd.wide_deep import _DirectComm
    comm = _DirectComm()
    t, u = torch.ones(1 << 20, device=dev), torch.ones(1 << 20, device=dev)
    comm.all_reduce(t)
    torch.cuda.synchronize()
    time.sleep(0.5)
    # an eager asynchronous reduction, NOT yet retired by torch's watchdog thread when the capture begins ...
    w = comm.all_reduce(t, async_op=True)
    w.wait()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        w = comm.all_reduce(u, async_op=True)         # ... pulls the group's internal stream into the capture
        time.sleep(0.5)                               # the watchdog looks at the eager work meanwhile (every 100 ms)
        w.wait()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    del g
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eager_async_reduction_cannot_abort_a_later_capture(dev):
    """On HIP an event cannot be queried while its stream captures, and ProcessGroupNCCL's watchdog queries the end event of every
    eager work it has not retired: an eager async_op=True reduction (event on the group's internal stream) followed by a captured one
    aborted the process from the watchdog thread whenever the watchdog was late (tools/probes/rccl_capture_race_probe.py).  The
    communicator's eager reductions run on a stream of its own instead; this would SIGABRT otherwise."""
    mp.spawn(_race_worker, args=(_free_port(),), nprocs=1, join=True)
