"""world_size-2, -4 and -8 gloo tests (CPU) of the row-sharded Wide&Deep step: two ranks, each with its own
batch and half of both tables, must reproduce the single-process step on the concatenated batch.
The kernels are stood in by the oracle (tests/_oracle_ops.py); what is under test is the engine's
host logic: routing, the all-to-all protocol, row-gradient exchange and gradient averaging."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _cfg(B, dropout=False):
    from mindrec_amd.wide_deep import WideDeepConfig
    return WideDeepConfig(vocab_size=997, emb_dim=8, field_size=39, batch_size=B, deep_layer_dim=[16, 8], mlp_dtype="fp32",
                          dropout_flag=dropout)


def _batch(cfg, seed):
    from mindrec_amd.wide_deep import synthetic_batch
    return synthetic_batch(cfg, "cpu", "zipf", seed=seed)


def _worker(rank, world, port, steps, out_dir, dropout=False, bypass=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepEngine
    torch.set_num_threads(1)
    cfg = _cfg(24, dropout)
    if bypass:
        # the PRODUCT's exchange: ShardStepMixin._exchange over _DirectComm.all_to_all_lists with the aliased send / receive
        # windows of one buffer and the rank's own chunk left in place (what runs over RCCL), here over gloo
        class _Bypass(OracleWideDeepEngine):
            _cpu_bypass = True
        eng = _Bypass(cfg, "cpu", rank=rank, world=world)
        assert eng._bypass and type(eng.comm).__name__ == "_DirectComm"
    else:
        eng = OracleWideDeepEngine(cfg, "cpu", rank=rank, world=world)
    losses = []
    for s in range(steps):
        ids, wts, label = _batch(cfg, seed=100 * s + rank)
        losses.append(float(eng.train_step(ids, wts, label)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), deep=eng.deep.numpy(), deep_m=eng.deep_m.numpy(),
             wide=eng.wide.numpy(), wide_accum=eng.wide_accum.numpy(), dense=eng.dense_flat.detach().numpy(),
             wide_b=eng.wide_b.numpy(), losses=np.array(losses))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,dropout,bypass", [(2, False, False), (4, False, False), (8, False, False), (2, True, False),
                                                  (2, False, True), (3, False, True)])
def test_sharded_step_matches_single_process(tmp_path, world, dropout, bypass):
    """world = 4 / 8 with V = 997 also cover shards of unequal length (250, 249, ... / 125, 125, ..., 124 rows);
    world = 8 is the driver's scaling-bench geometry (BASELINE configs[3]).  dropout: the ranks draw the Dropout mask of the
    one concatenated batch (row0 = rank * local batch)."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepEngine
    steps = 3
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path), dropout, bypass), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]

    # single process, concatenated batch of world x 24
    cfg1 = _cfg(24 * world, dropout)
    eng = OracleWideDeepEngine(cfg1, "cpu")
    losses = []
    for s in range(steps):
        parts = [_batch(_cfg(24), seed=100 * s + k) for k in range(world)]
        ids, wts, label = (torch.cat([p[i] for p in parts]) for i in range(3))
        losses.append(float(eng.train_step(ids, wts, label)))

    V = cfg1.vocab_size
    for name, full in (("deep", eng.deep), ("deep_m", eng.deep_m), ("wide", eng.wide), ("wide_accum", eng.wide_accum)):
        full = full.numpy()
        merged = np.empty_like(full)
        for k in range(world):
            merged[k::world] = r[k][name]                 # owner = id mod world, local row = id div world
        assert merged.shape[0] == V
        assert np.allclose(merged, full, rtol=2e-5, atol=1e-7), name
        touched = np.any(merged != 0, axis=1) if name != "wide_accum" else np.any(merged != 1, axis=1)
        assert touched.sum() > 10
    for k in range(world):
        assert np.allclose(r[k]["dense"], eng.dense_flat.detach().numpy(), rtol=2e-5, atol=1e-7)   # DP: replicas agree
        assert np.allclose(r[k]["wide_b"], eng.wide_b.numpy(), rtol=2e-5, atol=1e-8)
    for k in range(1, world):
        assert np.array_equal(r[0]["dense"], r[k]["dense"])
    # mean of the per-rank mean losses == loss of the concatenated batch
    assert np.allclose(sum(r[k]["losses"] for k in range(world)) / world, losses, rtol=1e-5)


def _overflow_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _oracle_engine import OracleWideDeepEngine
    from mindrec_amd.wide_deep_shard import ShardCapacityError
    torch.set_num_threads(1)
    cfg = _cfg(24)
    cfg.shard_capacity_factor = 1.0            # a bucket holds the mean share (rounded up to 64)
    eng = OracleWideDeepEngine(cfg, "cpu", rank=rank, world=world)
    ids, wts, label = _batch(cfg, seed=rank)
    flat = torch.arange(ids.numel(), dtype=ids.dtype)
    # rank 0: every id even -- all of its positions go to owner 0, whose bucket overflows; rank 1: evenly spread, drops nothing
    ids = ((flat % 400) * 2 if rank == 0 else flat % cfg.vocab_size).view_as(ids)
    eng.train_step(ids, wts, label)
    local = eng.shard_overflow()
    raised = False
    try:
        eng.train_step(ids, wts, label)        # the library itself raises at the next call, on EVERY rank
    except ShardCapacityError:
        raised = True
    # every rank grows the messages together and goes on: the same batch now fits (rank 0's 2 x 24 x 39 positions all go to owner 0)
    from mindrec_amd.wide_deep_shard import grow_shard_capacity
    grown = grow_shard_capacity(eng, 2.0)[0]
    before = eng.shard_overflow()
    eng.train_step(ids, wts, label)
    eng.check_shard_overflow()                 # (raises if the grown capacity dropped anything: it does not)
    np.savez(os.path.join(out_dir, f"ovf{rank}.npz"), local=local, raised=raised, grown=grown, more=eng.shard_overflow() - before)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dropped_positions_raise_on_every_rank(tmp_path):
    """positions that do not fit the fixed-capacity request are counted on the device; the engine -- not the caller --
    raises ShardCapacityError one call later, and on ALL ranks (the count rides the dense all-reduce), so nobody is left
    waiting inside a collective."""
    mp.spawn(_overflow_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"ovf{k}.npz") for k in range(2)]
    assert int(r[0]["local"]) > 0 and int(r[1]["local"]) == 0          # only rank 0 dropped something ...
    assert bool(r[0]["raised"]) and bool(r[1]["raised"])               # ... and both ranks raised
    assert float(r[0]["grown"]) == 2.0 and int(r[0]["more"]) == 0 and int(r[1]["more"]) == 0      # grow_shard_capacity: the batch fits, training goes on


def test_engine_refuses_cpu_without_kernels():
    from mindrec_amd.wide_deep import WideDeepEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        WideDeepEngine(_cfg(4), "cpu")


def test_checkpoint_roundtrip_and_shard_merge(tmp_path):
    """save -> train on -> load restores the exact state and training continues identically; two rank
    shards merge back into the single-process tables (eval.py:86-107 analogue)."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepEngine, load_checkpoint, merge_shards, save_checkpoint
    cfg = _cfg(24)
    a = OracleWideDeepEngine(cfg, "cpu")
    for s in range(2):
        a.train_step(*_batch(cfg, seed=s))
    save_checkpoint(a, tmp_path / "a.pt")
    ref = [float(a.train_step(*_batch(cfg, seed=10 + s))) for s in range(2)]
    b = OracleWideDeepEngine(cfg, "cpu")
    load_checkpoint(b, tmp_path / "a.pt")
    got = [float(b.train_step(*_batch(cfg, seed=10 + s))) for s in range(2)]
    assert got == ref
    assert torch.equal(a.deep, b.deep) and torch.equal(a.deep_v, b.deep_v) and torch.equal(a.wide_linear, b.wide_linear)
    assert torch.equal(a.dense_flat, b.dense_flat) and b.step_count == a.step_count
    # shards of a 2-rank layout interleave back to the full table
    shards = []
    for r in range(2):
        e = OracleWideDeepEngine(cfg, "cpu", rank=r, world=2)
        save_checkpoint(e, tmp_path / f"r{r}.pt")
        shards.append(tmp_path / f"r{r}.pt")
    full = merge_shards(shards)
    fresh = OracleWideDeepEngine(cfg, "cpu")
    assert torch.equal(full["deep"], fresh.deep) and torch.equal(full["wide"], fresh.wide)
    with pytest.raises(ValueError):
        load_checkpoint(OracleWideDeepEngine(cfg, "cpu", rank=1, world=2), tmp_path / "a.pt")


def test_train_steps_without_graphs_is_step_by_step():
    """train_steps (sink_size steps per host call) on an engine that has no whole-step graph -- here the CPU stand-in -- is the
    same as that many train_step calls, and hands out losses that later steps do not overwrite."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepEngine
    cfg = _cfg(24)
    a = OracleWideDeepEngine(cfg, "cpu")
    b = OracleWideDeepEngine(cfg, "cpu")
    bs = [_batch(cfg, seed=40 + s) for s in range(4)]
    la = [float(x) for x in a.train_steps(bs)]
    lb = [float(b.train_step(*x)) for x in bs]
    assert la == lb and len(set(la)) == 4
    assert torch.equal(a.deep, b.deep) and torch.equal(a.dense_flat, b.dense_flat) and a.step_count == b.step_count == 4

