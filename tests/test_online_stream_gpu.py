"""SURVEY 8(f) row 4 on the GPU: the online-learning path of examples/online_learning/online_train.py:30-46,71-86 -- an unbounded
row stream -> `GeneratorDataset(...).batch(B)` -> `RecModel.online_train(..., callbacks=[ModelCheckpoint(prefix, directory,
CheckpointConfig(save_checkpoint_steps=100, keep_checkpoint_max=5))])` -- through compat/mindspore + this repo's mindspore_rec
on the HIP kernel set: checkpoints appear every 100 steps, only the newest `keep_checkpoint_max` stay, and a fresh network
restored from one of them continues BIT-IDENTICALLY (tables, Adam moments, FTRL accumulators, the optimizers' step scalars)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

B, F, V, D = 64, 9, 3000, 8


class Stream:
    """A row source in the reference's shape: `__getitem__` ignores the index and hands out the NEXT row of the stream (the
    reference pops rows off a Kafka-fed buffer, online_train.py:35-43); here row number t is a function of t, so a restored
    run can be fast-forwarded."""

    def __init__(self, start=0):
        self.t = start

    def row(self, t):
        rng = np.random.default_rng(10_000 + t)
        ids = np.minimum(rng.zipf(1.3, size=F) + 2, V - 1).astype(np.int32)
        ids[:3] = np.arange(3)
        wts = np.ones(F, np.float32)
        wts[:3] = rng.random(3).astype(np.float32)
        return ids, wts, np.array([float(rng.random() < 0.3)], np.float32)

    def __getitem__(self, item):
        r = self.row(self.t)
        self.t += 1
        return r

    def __len__(self):
        return 2 ** 20 - 1


def _build(ms, dynamic=False):
    import _ms_models
    ms.set_seed(1000)
    net = _ms_models.WideDeep(V, D, F, B, [32, 16, 16, 8], sparse=True, dynamic=dynamic)
    step = _ms_models.WideDeepTrainStep(_ms_models.WideDeepLoss(net, 8e-5, with_l2=False), lazy=True)
    step.set_train()
    return step, net


def _table_state(t):
    """A table's contents in a comparable form: the tensor, or (keys, rows) of a MapParameter sorted by key."""
    if isinstance(t, torch.Tensor):
        return [t.as_subclass(torch.Tensor)]
    k, v = t.get_data()
    k, v = k.as_subclass(torch.Tensor), v.as_subclass(torch.Tensor)
    o = torch.argsort(k)
    return [k[o], v[o]]


@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("mode", ["pynative", "graph"])
def test_online_train_on_a_stream_with_periodic_checkpoints(dev, tmp_path, mode, dynamic):
    """mode "pynative": every step runs primitive by primitive on the HIP kernel set; "graph": the train cell is lowered to the
    fused engine (mindrec_amd/lowering.py) -- checkpoints are then read out of, and restored into, engine memory through the
    cell's re-bound Parameters.  dynamic: both tables are HashEmbeddingLookups over MapParameters (the reference's
    --dynamic_embedding, wide_and_deep.py:271-274); lowered, the two maps stand on the engine's one key index."""
    compat = os.path.abspath(os.path.join(HERE, "..", "compat"))
    if compat not in sys.path:
        sys.path.insert(0, compat)
    import mindspore as ms
    from mindspore import _hip_kernels, context
    import mindspore.dataset as ds
    from mindspore.train.callback import Callback, CheckpointConfig, ModelCheckpoint, TimeMonitor
    from mindspore.train.serialization import load_checkpoint, load_param_into_net
    from mindspore_rec import RecModel
    prev = ms._kernels._install(_hip_kernels)
    context.set_context(mode=context.GRAPH_MODE if mode == "graph" else context.PYNATIVE_MODE, device_target="GPU", device_id=0)
    try:
        class Watch(Callback):
            def __init__(self, stop_at):
                self.stop_at, self.losses = stop_at, []

            def step_end(self, run_context):
                p = run_context.original_args()
                self.losses.append(float(p.net_outputs[0].asnumpy()))
                if p.cur_step_num >= self.stop_at:
                    run_context.request_stop()

        # run A: 250 steps off the stream, a checkpoint every 100
        step, net = _build(ms, dynamic)
        data = ds.GeneratorDataset(Stream(), column_names=["id", "weight", "label"]).batch(B)
        assert data.get_dataset_size() == (2 ** 20 - 1 + B - 1) // B
        watch = Watch(250)
        ck = ModelCheckpoint(prefix="widedeep_train", directory=str(tmp_path), config=CheckpointConfig(save_checkpoint_steps=100, keep_checkpoint_max=2))
        RecModel(step).online_train(data, callbacks=[TimeMonitor(1), watch, ck], dataset_sink_mode=True)
        files = sorted(os.listdir(tmp_path))
        assert len(watch.losses) == 250 and len(files) == 2, files            # steps 100, 200 and the final one at 250: the newest two stay
        assert ck.latest_ckpt_file_name.endswith(".ckpt") and os.path.basename(ck.latest_ckpt_file_name) in files
        ck200 = [f for f in files if f != os.path.basename(ck.latest_ckpt_file_name)][0]

        # run B: a fresh network, restored from the step-200 checkpoint, fed the stream from row 200 * B on
        step2, net2 = _build(ms, dynamic)
        params = load_checkpoint(os.path.join(str(tmp_path), ck200))
        assert int(params["step_num"]) == 200
        missing = load_param_into_net(step2, {k: v for k, v in params.items() if k not in ("epoch_num", "step_num")})
        assert missing == [], missing
        assert step2.opt_deep.global_step == 200 and step2.opt_wide.global_step == 200
        data2 = ds.GeneratorDataset(Stream(start=200 * B), column_names=["id", "weight", "label"]).batch(B)
        watch2 = Watch(50)
        RecModel(step2).online_train(data2, callbacks=[watch2], dataset_sink_mode=True)
        assert watch2.losses == watch.losses[200:250]                          # bit for bit
        for a, b in zip(_table_state(net2.deep_table.embedding_table) + _table_state(net2.wide_table.embedding_table),
                        _table_state(net.deep_table.embedding_table) + _table_state(net.wide_table.embedding_table)):
            assert torch.equal(a, b)
        assert torch.equal(net2.layer0.weight, net.layer0.weight) and torch.equal(net2.wide_bias, net.wide_bias)
        assert (step.__dict__.get("_lowered") not in (None, False)) == (mode == "graph"), step.__dict__.get("_lowering_refused")
    finally:
        context.set_context(mode=context.GRAPH_MODE)
        ms._kernels._install(prev)
