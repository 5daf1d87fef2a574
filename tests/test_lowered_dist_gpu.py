"""GRAPH_MODE's lowering of a DISTRIBUTED train cell (mindrec_amd/lowering.py): two ranks sharing the test
box's one GPU (gloo group; device tensors staged through the host by the product's own communicator) drive
`mindspore._lower.lowered()` -- through `Model._run_step`, not a hand-built engine -- with a mindspore-style Wide&Deep train cell
that owns two `DistributedGradReducer`s (tests/_ms_models.py; the reference's flow: models/wide_deep/src/wide_and_deep.py:458-470,
487-489 set up by train_and_eval_distribute.py:123-140), fed the per-rank batches of the fixtures the REFERENCE's own distributed
script wrote:

  ref_wd_dp2_dynamic.npz  (--dynamic_embedding --sparse: row gradients)  -> lowered onto the ROW-SHARDED engine, never a one-rank one
  ref_wd_dp2.npz          (sparse=False: dense [V, D] table gradients)    -> refused; runs primitive by primitive WITH its all-reduce

Both must reproduce the reference's per-rank losses and final parameters."""
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir, case):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "compat")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    import mindspore
    from mindspore import _hip_kernels, context
    from mindspore.communication.management import init
    mindspore._kernels._install(_hip_kernels)
    context.set_context(mode=context.GRAPH_MODE, device_target="GPU", device_id=0)
    init("gloo")                                           # (RCCL needs one GPU per rank; the test box has one)
    context.set_auto_parallel_context(parallel_mode=context.ParallelMode.DATA_PARALLEL, gradients_mean=True, device_num=world)
    import _ms_models
    import _ref_fixtures as RF
    from mindrec_amd.lowering import LoweredStep
    z, cfg, comp = RF.load(case)
    dyn = bool(cfg["dynamic_embedding"])
    zi = {k[len("rank0/"):]: z[k] for k in z.files if k.startswith("rank0/init/")}
    if dyn:
        mindspore.set_seed(1000)
        from mindspore.common import initializer as I
        I._state["calls"] = int(z["rank0/seed/embedding_table"]) - (1000 * 1_000_003) - 1
    step, net = _ms_models.wide_deep_from_fixture(zi, cfg, comp, capacity=8192, reduce=True)
    model = mindspore.Model(step)
    losses = []
    for s in range(int(z["steps"])):
        batch = tuple(mindspore.Tensor(z[f"rank{rank}/{k}"][s]) for k in ("ids", "wts", "label"))
        lw, ld = model._run_step(step, batch)
        losses.append((float(lw.asnumpy()), float(ld.asnumpy())))
    low = step.__dict__["_lowered"]
    out = {"losses": np.array(losses), "refused": np.array(str(step.__dict__.get("_lowering_refused", "")))}
    if dyn:
        assert isinstance(low, LoweredStep) and low.sharded and low.engine.world == world and low.engine.rank == rank, \
            step.__dict__.get("_lowering_refused")
        assert low.engine.shard_overflow() == 0
        assert low.dirty
        with pytest.raises(RuntimeError, match="stale"):                      # a one-rank read of stale mirrors is refused, not answered
            from mindspore.train.serialization import save_checkpoint
            save_checkpoint(step, os.path.join(out_dir, f"never{rank}.ckpt"))
        model.sync_parameters()                                               # the collective refresh (Model.train calls it itself)
        assert not low.dirty
        for name, mp_ in (("deep", net.deep_table.embedding_table), ("wide", net.wide_table.embedding_table)):
            k, v = mp_.get_data()
            out[name + "_keys"], out[name + "_vals"] = k.asnumpy(), v.asnumpy()
        # evaluation through the MODEL CELL is the engine's collective forward over the shards
        net.set_train(False)
        b = tuple(mindspore.Tensor(z[f"rank{rank}/{k}"][0]) for k in ("ids", "wts"))
        logits, table = net(*b)
        ref = low.engine.predict(b[0].as_subclass(torch.Tensor), b[1].as_subclass(torch.Tensor))[0]
        assert table is net.table and np.array_equal(logits.asnumpy().reshape(-1), ref.cpu().numpy().reshape(-1))
        net.set_train(True)
    else:
        assert low is False and "dense table gradients" in str(out["refused"]), (low, out["refused"])
        out["deep"], out["wide"] = net.deep_table.embedding_table.asnumpy(), net.wide_table.embedding_table.asnumpy()
    out["wide_b"] = net.wide_bias.asnumpy()
    for i in range(net.n_layers):
        out[f"w{i}"], out[f"b{i}"] = getattr(net, f"layer{i}").weight.asnumpy(), getattr(net, f"layer{i}").bias.asnumpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    model.close()
    dist.destroy_process_group()


def _ref_losses(z, world):
    logs = json.loads(str(z["logs"]))
    pat = re.compile(r"wide_loss: ([0-9.eE+-]+), deep_loss: ([0-9.eE+-]+)")
    return [np.array([[float(v) for v in pat.search(ln).groups()] for ln in logs[f"loss_log{k}"]]) for k in range(world)]


@pytest.mark.timeout(600)
def test_distributed_row_gradient_cell_is_lowered_onto_the_row_sharded_engine(dev, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ref_fixtures as RF
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "ref_wd_dp2_dynamic"), nprocs=2, join=True)
    z, cfg, comp = RF.load("ref_wd_dp2_dynamic")
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    ref = _ref_losses(z, 2)
    for k in range(2):
        assert np.allclose(r[k]["losses"][:, 0], ref[k][:, 0], rtol=2e-6, atol=0), (k, r[k]["losses"], ref[k])
        assert np.allclose(r[k]["losses"][:, 1], ref[k][:, 1], rtol=2e-6, atol=0)
    trained = np.unique(np.concatenate([z[f"rank{k}/ids"].reshape(-1) for k in range(2)]).astype(np.int64))
    for name, fx in (("deep", "embedding_table"), ("wide", "wide_embeddinglookup.embedding_table")):
        # after the sync BOTH ranks' MapParameters export the whole table (the union of the shards)
        assert np.array_equal(np.sort(r[0][name + "_keys"]), np.sort(r[1][name + "_keys"]))
        keys, vals = r[0][name + "_keys"].astype(np.int64), r[0][name + "_vals"]
        order = np.argsort(keys)
        assert np.array_equal(keys[order], trained), name                       # every trained key, each exactly once
        rk, rv = z[f"rank0/final/{fx}::keys"].astype(np.int64), z[f"rank0/final/{fx}::values"]
        pos = np.searchsorted(rk, trained)
        assert np.array_equal(rk[pos], trained)
        got = vals[order]
        assert (RF.row_rel(got, rv[pos]) <= 1e-5) if got.shape[1] > 1 else np.allclose(got, rv[pos], rtol=1e-4, atol=1e-8), name
    for i in range(5):
        assert np.array_equal(r[0][f"w{i}"], r[1][f"w{i}"]) and np.array_equal(r[0][f"b{i}"], r[1][f"b{i}"])      # dense replicas stay identical
        assert np.allclose(r[0][f"w{i}"], z[f"rank0/final/dense_layer_{i + 1}.weight"], rtol=2e-4, atol=1e-7), i
    assert np.allclose(r[0]["wide_b"], z["rank0/final/wide_b"], rtol=1e-4, atol=1e-8)


@pytest.mark.timeout(600)
def test_distributed_dense_gradient_cell_is_refused_and_runs_with_its_all_reduce(dev, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ref_fixtures as RF
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "ref_wd_dp2"), nprocs=2, join=True)
    z, cfg, comp = RF.load("ref_wd_dp2")
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    ref = _ref_losses(z, 2)
    for k in range(2):
        assert str(r[k]["refused"]) == str(r[0]["refused"])                                    # the same decision on every rank
        assert np.allclose(r[k]["losses"], ref[k], rtol=2e-6, atol=0), (k, r[k]["losses"], ref[k])
    for name in ("deep", "wide", "wide_b"):
        assert np.array_equal(r[0][name], r[1][name]), name                                      # the replicas did NOT diverge
    assert RF.row_rel(r[0]["deep"], z["rank0/final/embedding_table"]) <= 1e-5
    assert np.allclose(r[0]["wide"], z["rank0/final/wide_embeddinglookup.embedding_table"], rtol=1e-4, atol=1e-8)
    for i in range(5):
        assert np.allclose(r[0][f"w{i}"], z[f"rank0/final/dense_layer_{i + 1}.weight"], rtol=2e-4, atol=1e-7), i
