"""GPU: the fixtures recorded from the REFERENCE's own Python (tests/golden/ref_*.npz; generator make_ref_fixtures.py, which
ran models/wide_deep/src/wide_and_deep.py, models/deep_and_cross/src/deep_and_cross.py and mindspore_rec/ unmodified over
compat/mindspore with the oracle's primitives) replayed on an MI355X:

  (1) through the fused engines (WideDeepEngine / DeepCrossEngine: the benchmarked product path, HIP kernels + HIP graphs),
  (2) through compat/mindspore on the HIP kernel set, driven by this repo's own mindspore_rec package and a mindspore-style
      script (tests/_ms_models.py) -- the reference's sources cannot travel to the GPU box.

Bars (north_star): keys / dedup bit-exact; fp32 embedding rows 1e-5 row-relative; losses 2e-6 (fp32 nets); the fp16 net is
compared within the reference's own double rounding (wide_and_deep.py:121-126 rounds the MatMul output and the bias sum
separately, the kernels round once)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_fixtures as RF  # noqa: E402


def _check_wd_engine(eng, z, cfg, losses, rows_tol=1e-5, dense_rtol=1e-4, loss_rtol=2e-6, wide_tol=1e-4):
    assert np.allclose(losses, z["loss_w"], rtol=loss_rtol, atol=0), (losses, z["loss_w"])
    for k, v in RF.wd_dense_state(eng).items():
        assert np.allclose(v, z["final/" + k], rtol=dense_rtol, atol=dense_rtol * 1e-3), k
    if cfg["dynamic_embedding"]:
        keys, rows = eng.index.export()
        keys, rows = keys.cpu().numpy(), rows.cpu().numpy().astype(np.int64)
        order = np.argsort(keys)
        assert np.array_equal(keys[order], z["final/embedding_table::keys"].astype(np.int64))          # the same keys were created
        assert np.array_equal(keys[order], z["final/wide_embeddinglookup.embedding_table::keys"].astype(np.int64))
        deep = eng.deep.cpu().numpy()[rows[order]]
        wide = eng.wide.cpu().numpy()[rows[order]]
        ref_d, ref_w = z["final/embedding_table::values"], z["final/wide_embeddinglookup.embedding_table::values"]
    else:
        deep, wide = eng.deep.cpu().numpy(), eng.wide.cpu().numpy()
        ref_d, ref_w = z["final/embedding_table"], z["final/wide_embeddinglookup.embedding_table"]
    assert RF.row_rel(deep, ref_d) <= rows_tol, RF.row_rel(deep, ref_d)
    assert np.abs(wide - ref_w).max() <= wide_tol * np.abs(ref_w).max()


@pytest.mark.parametrize("case", ["ref_wd_sparse", "ref_wd_dense", "ref_wd_dynamic"])
@pytest.mark.parametrize("graphs", ["step", "none"])
def test_wide_deep_engine_replays_reference_fixture(dev, case, graphs):
    from mindrec_amd.wide_deep import WideDeepEngine
    z, cfg, comp = RF.load(case)
    over = dict(graphs=graphs, hash_capacity=1 << 12)
    if cfg["dynamic_embedding"]:
        over["seed"] = int(z["deep_seed"])              # default rows are a function of (seed, key): deep = seed, wide = seed + 1
    eng = WideDeepEngine(RF.wd_config(cfg, comp, **over), dev)
    RF.wd_load_init(eng, z, dynamic=bool(cfg["dynamic_embedding"]))
    losses = RF.wd_replay(eng, z, dev)
    _check_wd_engine(eng, z, cfg, losses)
    if not cfg["sparse"]:      # the deep optimizer's loss: + l2_coef * sum(E^2) / 2 at the step's starting values (:356-360)
        assert np.isclose(eng.deep_loss(losses[-1]), z["loss_d"][-1], rtol=2e-6, atol=0) and z["loss_d"][-1] > z["loss_w"][-1]
    if not cfg["dynamic_embedding"]:
        untouched = np.ones(cfg["vocab_size"], bool)
        untouched[z["ids"].reshape(-1)] = False
        if cfg["sparse"]:                                   # lazy: rows the batches never touched are bit-identical to their initial values
            assert np.array_equal(eng.deep.cpu().numpy()[untouched], z["init/embedding_table"][untouched])
        assert np.allclose(eng.deep_m.cpu().numpy(), z["state/moment1/embedding_table"], rtol=1e-4, atol=1e-9)
        assert np.allclose(eng.wide_accum.cpu().numpy(), z["state/accum/wide_embeddinglookup.embedding_table"], rtol=1e-5, atol=0)
    # PredictWithSigmoid on the last batch (wide_and_deep.py:495-518)
    ids, wts = (torch.from_numpy(z[k][-1]).to(dev) for k in ("ids", "wts"))
    logit, prob = eng.predict(ids, wts)
    assert np.allclose(logit.cpu().numpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=1e-4, atol=1e-6)
    assert np.allclose(prob.cpu().numpy().reshape(-1), z["eval_probs"].reshape(-1), rtol=1e-5, atol=1e-7)


def test_wide_deep_engine_fp16_net_against_the_reference_fp16_run(dev):
    """use_mixed_precision=True (default_config.yaml:28), the benchmarked arithmetic.  The reference's DenseLayer rounds the fp16
    MatMul output, then the fp16 bias sum; the kernels add an fp32 bias to the fp32 accumulator and round once (DESIGN.md
    section 2): at most an fp16 ulp per activation, so the two runs are compared at 5e-4 on the loss, 1e-4 row-relative on the rows."""
    from mindrec_amd.wide_deep import WideDeepEngine
    z, cfg, comp = RF.load("ref_wd_mixed")
    eng = WideDeepEngine(RF.wd_config(cfg, comp), dev)
    assert eng._mfma
    RF.wd_load_init(eng, z)
    losses = RF.wd_replay(eng, z, dev)
    assert np.allclose(losses, z["loss_w"], rtol=5e-4, atol=0), (losses, z["loss_w"])
    touched = np.zeros(cfg["vocab_size"], bool)
    touched[z["ids"].reshape(-1)] = True
    deep, ref = eng.deep.cpu().numpy(), z["final/embedding_table"]
    assert np.array_equal(deep[~touched], ref[~touched])
    step = np.abs(ref[touched] - z["init/embedding_table"][touched]).max()
    assert np.abs(deep[touched] - ref[touched]).max() <= 0.15 * step          # the updates agree to 15 % of the largest update ...
    assert RF.row_rel(deep, ref) <= 1e-4                                       # ... and the rows to 1e-4 of their own scale


def test_deep_cross_engine_replays_reference_fixture(dev):
    from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine
    z, cfg, comp = RF.load("ref_dcn")
    eng = DeepCrossEngine(DeepCrossConfig(vocab_size=cfg["vocab_size"], emb_dim=cfg["emb_dim"], field_size=cfg["field_size"],
                                          batch_size=cfg["batch_size"], deep_layer_dim=list(cfg["deep_layer_dim"]),
                                          cross_layer_num=cfg["cross_layer_num"], learning_rate=comp["lr"], eps=comp["eps"],
                                          loss_scale=comp["loss_scale"]), dev)
    assert eng._native
    RF.dcn_load_init(eng, z)
    losses = np.array([float(eng.train_step(*(torch.from_numpy(z[k][s]).to(dev) for k in ("ids", "wts", "label"))))
                       for s in range(z["ids"].shape[0])])
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    for k, v in RF.dcn_state(eng).items():
        assert np.allclose(v, z["final/" + k], rtol=2e-4, atol=1e-7), k
    logit, prob = eng.predict(*(torch.from_numpy(z[k][-1]).to(dev) for k in ("ids", "wts")))
    assert np.allclose(logit.cpu().numpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", ["ref_deepfm", "ref_deepfm_mixed"])
def test_deepfm_engine_replays_reference_fixture(dev, case):
    """The reference's DeepFM (models/deepfm/src/deepfm.py through ModelBuilder; fp32 DenseLayers, and its default fp16 ones) on the
    HIP engine: losses, both tables, the net.  The fp16 case is held within the reference's own double rounding (its DenseLayer
    rounds the MatMul output and the bias sum separately and runs the OUTPUT layer in fp16 too, :135-145; the kernels round once
    and keep the output layer in fp32)."""
    from mindrec_amd.deepfm import DeepFMEngine
    z, cfg, comp = RF.load(case)
    mixed = bool(comp["convert_dtype"])
    eng = DeepFMEngine(RF.deepfm_config(cfg, comp), dev)
    assert eng._mfma == mixed and (mixed or eng._f32net)
    RF.deepfm_load_init(eng, z)
    losses = np.array([float(eng.train_step(*(torch.from_numpy(z[k][s]).to(dev) for k in ("ids", "wts", "label"))))
                       for s in range(z["ids"].shape[0])])
    assert np.allclose(losses, z["loss"], rtol=5e-4 if mixed else 2e-6, atol=0), (losses, z["loss"])
    for k, v in RF.deepfm_state(eng).items():
        if mixed:
            assert np.abs(v - z["final/" + k]).max() <= 0.15 * max(np.abs(z["final/" + k] - z["init/" + k]).max(), 1e-12) + 1e-7, k
        else:
            assert np.allclose(v, z["final/" + k], rtol=2e-4, atol=1e-7), k
    logit, _ = eng.predict(*(torch.from_numpy(z[k][-1]).to(dev) for k in ("ids", "wts")))
    assert np.allclose(logit.cpu().numpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=2e-2 if mixed else 1e-4, atol=1e-4 if mixed else 1e-6)


def test_train_and_eval_flow_on_the_engine_matches_the_reference_script(dev, tmp_path):
    """The same flow (see tests/test_ref_fixtures.py) with the product engine on the MI355X: loss.log / eval.log of the reference's
    own train_and_eval.py run."""
    from mindrec_amd.wide_deep import WideDeepEngine
    z, cfg, comp = RF.load("ref_train_eval_flow")
    with WideDeepEngine(RF.wd_config(cfg, comp), dev) as eng:
        RF.wd_load_init(eng, z)
        RF.check_train_eval_flow(z, RF.run_train_eval_flow(eng, z, dev, str(tmp_path)))


def test_reference_data_parallel_run_equals_the_engine_on_the_concatenated_batch(dev):
    """ref_wd_dp2.npz (models/wide_deep/train_and_eval_distribute.py, two processes) against the product engine on the MI355X fed the
    concatenation of the two ranks' batches: see tests/test_ref_fixtures.py."""
    from mindrec_amd.wide_deep import WideDeepEngine
    RF.check_data_parallel_fixture(lambda c: WideDeepEngine(c, dev), dev)


# ---- compat/mindspore on the HIP kernel set ---------------------------------------------------------------------------------------
@pytest.fixture
def ms_hip(dev):
    compat = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compat"))
    if compat not in sys.path:
        sys.path.insert(0, compat)
    import mindspore
    from mindspore import context
    from mindspore import _hip_kernels
    prev = mindspore._kernels._install(_hip_kernels)          # (a CPU test of the same session may have left its own set installed)
    context.set_context(mode=context.GRAPH_MODE, device_target="GPU", device_id=0)
    yield mindspore
    context.set_context(mode=context.GRAPH_MODE)
    mindspore._kernels._install(prev)


def test_hash_embedding_lookup_on_hip_matches_reference_package(ms_hip):
    import mindspore_rec
    z = np.load(os.path.join(RF.GOLDEN, "ref_hash_lookup.npz"))
    for i, v in enumerate(json.loads(str(z["variants"]))):
        kd = ms_hip.int32 if "int32" in v["key_dtype"] else ms_hip.int64
        ms_hip.set_seed(300 + i)
        layer = mindspore_rec.HashEmbeddingLookup(embedding_size=v["D"], key_dtype=kd, sparse=v["sparse"], max_norm=v["max_norm"], capacity=4096)
        assert layer.embedding_table.seed == v["seed"]
        for c in range(2):
            out = layer(ms_hip.Tensor(z[f"v{i}/keys"][c]))
            assert out.is_cuda
            ref = z[f"v{i}/out{c}"]
            assert np.array_equal(out.asnumpy(), ref) if v["max_norm"] is None else np.allclose(out.asnumpy(), ref, rtol=1e-6, atol=0)
        k, vals = layer.embedding_table.get_data()
        order = np.argsort(k.asnumpy())
        assert np.array_equal(k.asnumpy()[order], z[f"v{i}/table_keys"]) and np.array_equal(vals.asnumpy()[order], z[f"v{i}/table_values"])


@pytest.mark.parametrize("case", ["ref_wd_sparse", "ref_wd_dense", "ref_wd_dynamic", "ref_wd_mixed"])
def test_mindspore_style_script_on_hip_matches_reference(ms_hip, case):
    """PYNATIVE_MODE: the script's construct bodies run primitive by primitive on the HIP kernel set (no lowering)."""
    import _ms_models
    from mindspore import context
    context.set_context(mode=context.PYNATIVE_MODE)
    z, cfg, comp = RF.load(case)
    if cfg["dynamic_embedding"]:
        ms_hip.set_seed(1000)
        from mindspore.common import initializer as I
        I._state["calls"] = int(z["deep_seed"]) - (1000 * 1_000_003) - 1
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp, capacity=4096)
    losses = []
    for s in range(z["ids"].shape[0]):
        lw, ld = step(ms_hip.Tensor(z["ids"][s]), ms_hip.Tensor(z["wts"][s]), ms_hip.Tensor(z["label"][s]))
        assert lw.is_cuda
        losses.append((float(lw.asnumpy()), float(ld.asnumpy())))
    losses = np.array(losses)
    mixed = bool(cfg["use_mixed_precision"])
    assert np.allclose(losses[:, 0], z["loss_w"], rtol=5e-5 if mixed else 2e-6, atol=0), (losses[:, 0], z["loss_w"])
    assert np.allclose(losses[:, 1], z["loss_d"], rtol=5e-5 if mixed else 2e-6, atol=0)
    assert np.allclose(net.wide_bias.asnumpy(), z["final/wide_b"], rtol=1e-4, atol=1e-8)
    if cfg["dynamic_embedding"]:
        k, v = net.deep_table.embedding_table.get_data()
        order = np.argsort(k.asnumpy())
        assert np.array_equal(k.asnumpy()[order], z["final/embedding_table::keys"])
        if not mixed:
            assert RF.row_rel(v.asnumpy()[order], z["final/embedding_table::values"]) <= 1e-5
    elif not mixed:
        assert RF.row_rel(net.deep_table.embedding_table.asnumpy(), z["final/embedding_table"]) <= 1e-5


# ---- GRAPH_MODE's compile step: a recognised train cell is lowered to the fused engine (mindrec_amd/lowering.py) -----------------------
@pytest.mark.parametrize("case", ["ref_wd_sparse", "ref_wd_dense", "ref_wd_mixed"])
def test_lowered_wide_deep_script_matches_reference(ms_hip, case):
    """`mindspore.Model` recognises the script's train cell, moves its parameters into a WideDeepEngine, re-binds them as views of
    engine memory and runs every step as the engine's HIP graph; the losses and -- read back THROUGH THE CELL'S OWN Parameters --
    the trained tables and weights are the reference's."""
    import _ms_models
    from mindrec_amd.lowering import LoweredStep
    z, cfg, comp = RF.load(case)
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)
    model = ms_hip.Model(step)
    losses = []
    for s in range(z["ids"].shape[0]):
        batch = tuple(ms_hip.Tensor(z[k][s]) for k in ("ids", "wts", "label"))
        lw, ld = model._run_step(step, batch)
        losses.append((float(lw.asnumpy()), float(ld.asnumpy())))
    low = step.__dict__["_lowered"]
    assert isinstance(low, LoweredStep) and low.kind == "wide_deep", step.__dict__.get("_lowering_refused")
    mixed = bool(cfg["use_mixed_precision"])
    losses = np.array(losses)
    assert np.allclose(losses[:, 0], z["loss_w"], rtol=5e-4 if mixed else 2e-6, atol=0), (losses[:, 0], z["loss_w"])
    assert np.allclose(losses[:, 1], z["loss_d"], rtol=5e-4 if mixed else 2e-6, atol=0)
    # the cell's Parameters are the engine's memory
    assert net.deep_table.embedding_table.data_ptr() == low.engine.deep.data_ptr()
    deep = net.deep_table.embedding_table.asnumpy()
    assert RF.row_rel(deep, z["final/embedding_table"]) <= (1e-4 if mixed else 1e-5)
    if not mixed:
        assert np.allclose(net.wide_bias.asnumpy(), z["final/wide_b"], rtol=1e-4, atol=1e-8)
        assert np.allclose(net.layer0.weight.asnumpy(), z["final/dense_layer_1.weight"], rtol=1e-4, atol=1e-7)
    # an evaluation pass through the SAME model cell (eager primitives over the re-bound Parameters) sees the trained state
    net.set_train(False)
    logits, _ = net(ms_hip.Tensor(z["ids"][-1]), ms_hip.Tensor(z["wts"][-1]))
    assert np.allclose(logits.asnumpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=2e-2 if mixed else 1e-4, atol=1e-4 if mixed else 1e-6)


def test_lowered_hash_table_script_matches_reference(ms_hip):
    """The reference's --dynamic_embedding model (two HashEmbeddingLookups over MapParameters, wide_and_deep.py:271-274) lowered: the
    cell's two MapParameters stand on the engine's ONE key index and its row tables; losses, key set and rows are the reference's,
    read back through the cell's own MapParameters; an evaluation pass through the model cell reads the same memory."""
    import _ms_models
    from mindrec_amd.lowering import LoweredStep
    z, cfg, comp = RF.load("ref_wd_dynamic")
    ms_hip.set_seed(1000)
    from mindspore.common import initializer as I
    I._state["calls"] = int(z["deep_seed"]) - (1000 * 1_000_003) - 1
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp, capacity=4096)
    model = ms_hip.Model(step)
    losses = []
    for s in range(z["ids"].shape[0]):
        lw, ld = model._run_step(step, tuple(ms_hip.Tensor(z[k][s]) for k in ("ids", "wts", "label")))
        losses.append((float(lw.asnumpy()), float(ld.asnumpy())))
    low = step.__dict__["_lowered"]
    assert isinstance(low, LoweredStep) and low.kind == "wide_deep" and low.engine.index is not None, step.__dict__.get("_lowering_refused")
    losses = np.array(losses)
    assert np.allclose(losses[:, 0], z["loss_w"], rtol=2e-6, atol=0), (losses[:, 0], z["loss_w"])
    assert np.allclose(losses[:, 1], z["loss_d"], rtol=2e-6, atol=0)
    for name, mp in (("embedding_table", net.deep_table.embedding_table), ("wide_embeddinglookup.embedding_table", net.wide_table.embedding_table)):
        k, v = mp.get_data()
        order = np.argsort(k.asnumpy())
        assert np.array_equal(k.asnumpy()[order], z[f"final/{name}::keys"]), name
        assert RF.row_rel(v.asnumpy()[order], z[f"final/{name}::values"]) <= 1e-5, name
    assert len(net.deep_table.embedding_table) == len(low.engine.index)
    assert np.allclose(net.wide_bias.asnumpy(), z["final/wide_b"], rtol=1e-4, atol=1e-8)
    net.set_train(False)
    logits, _ = net(ms_hip.Tensor(z["ids"][-1]), ms_hip.Tensor(z["wts"][-1]))
    assert np.allclose(logits.asnumpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=1e-4, atol=1e-6)
    with pytest.raises(NotImplementedError, match="share the engine's one key index"):
        net.deep_table.embedding_table.erase(ms_hip.Tensor(z["ids"][0].reshape(-1)[:2]))


@pytest.mark.parametrize("case", ["ref_deepfm", "ref_deepfm_mixed"])
def test_lowered_deepfm_script_matches_reference(ms_hip, case):
    """A mindspore-style DeepFM train cell lowered onto DeepFMEngine: the reference's losses, and the trained tables / net read
    back through the cell's own (re-bound) Parameters."""
    import _ms_models
    from mindrec_amd.lowering import LoweredStep
    z, cfg, comp = RF.load(case)
    mixed = bool(comp["convert_dtype"])
    step, net = _ms_models.deepfm_from_fixture(z, cfg, comp)
    model = ms_hip.Model(step)
    losses = [float(model._run_step(step, tuple(ms_hip.Tensor(z[k][s]) for k in ("ids", "wts", "label"))).asnumpy()) for s in range(z["ids"].shape[0])]
    low = step.__dict__["_lowered"]
    assert isinstance(low, LoweredStep) and low.kind == "deepfm", step.__dict__.get("_lowering_refused")
    assert np.allclose(losses, z["loss"], rtol=5e-4 if mixed else 2e-6, atol=0), (losses, z["loss"])
    assert net.factors.data_ptr() == low.engine.V_l2.data_ptr()
    if not mixed:
        assert np.allclose(net.factors.asnumpy(), z["final/embedding_table"], rtol=2e-4, atol=1e-7)
        assert np.allclose(net.linear.asnumpy(), z["final/fm_w"], rtol=2e-4, atol=1e-7)
        assert np.allclose(net.layer1.bias.asnumpy(), z["final/dense_layer_2.bias"], rtol=2e-4, atol=1e-7)
    net.set_train(False)
    logits = net(ms_hip.Tensor(z["ids"][-1]), ms_hip.Tensor(z["wts"][-1]))
    assert np.allclose(logits.asnumpy().reshape(-1), z["eval_logits"].reshape(-1), rtol=2e-2 if mixed else 1e-4, atol=1e-4 if mixed else 1e-6)


def test_lowered_deep_cross_script_matches_reference(ms_hip):
    import _ms_models
    from mindrec_amd.lowering import LoweredStep
    z, cfg, comp = RF.load("ref_dcn")
    step, net = _ms_models.deep_cross_from_fixture(z, cfg, comp)
    model = ms_hip.Model(step)
    losses = [float(model._run_step(step, tuple(ms_hip.Tensor(z[k][s]) for k in ("ids", "wts", "label"))).asnumpy()) for s in range(z["ids"].shape[0])]
    low = step.__dict__["_lowered"]
    assert isinstance(low, LoweredStep) and low.kind == "deep_cross", step.__dict__.get("_lowering_refused")
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    assert np.allclose(net.out.weight.asnumpy(), z["final/dense_layer_3.weight"], rtol=2e-4, atol=1e-7)
    assert np.allclose(net.cross5.cross_bias.asnumpy(), z["final/cross_layer_6.cross_bias"], rtol=2e-4, atol=1e-7)
    assert np.allclose(net.lookup.embedding_table.asnumpy(), z["final/deep_embeddinglookup.embedding_table"], rtol=2e-4, atol=1e-7)


def test_a_model_the_engine_does_not_compute_is_not_lowered(ms_hip, caplog):
    """Structure alone does not prove the arithmetic: a Wide&Deep-shaped script whose hidden layers use tanh is recognised by
    structure, fails the verification against its own eager forward BEFORE anything of the cell is re-bound, and runs primitive by
    primitive -- with a WARNING, once -- instead of being computed wrongly."""
    import logging
    import _ms_models
    from mindspore import ops
    z, cfg, comp = RF.load("ref_wd_sparse")
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)
    for i in range(net.n_layers - 1):
        getattr(net, f"layer{i}").act = ops.Tanh()
    batch = tuple(ms_hip.Tensor(z[k][0]) for k in ("ids", "wts", "label"))
    ptr = net.layer0.weight.data_ptr()
    before = net.layer0.weight.asnumpy().copy()
    with caplog.at_level(logging.WARNING, logger="mindspore"):
        lw, ld = ms_hip.Model(step)._run_step(step, batch)
    assert step.__dict__["_lowered"] is False and "differ from the cell's eager forward" in step._lowering_refused
    assert any("NOT lowered" in r.getMessage() for r in caplog.records)
    assert net.layer0.weight.data_ptr() == ptr                    # the cell kept its own memory ...
    assert not np.array_equal(net.layer0.weight.asnumpy(), before) and np.isfinite(float(lw.asnumpy()))      # ... and trained eagerly


def test_a_loss_the_engine_does_not_compute_is_not_lowered(ms_hip):
    """The loss function is part of what the engine computes: a train cell whose loss is not the mean sigmoid cross-entropy of its
    logits is refused by the first-loss check, as is an FTRL with another lr_power."""
    import _ms_models
    z, cfg, comp = RF.load("ref_wd_sparse")
    batch = tuple(ms_hip.Tensor(z[k][0]) for k in ("ids", "wts", "label"))
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)

    class Doubled(type(step.loss_net)):
        def construct(self, ids, wts, label):
            lw, ld = super().construct(ids, wts, label)
            return lw * 2.0, ld * 2.0
    object.__setattr__(step.loss_net, "__class__", Doubled)          # (Cell.__setattr__ files plain attributes in its own dict)
    ms_hip.Model(step)._run_step(step, batch)
    assert step.__dict__["_lowered"] is False and "is not the mean sigmoid cross-entropy" in step._lowering_refused
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)
    step.opt_wide.lr_power = -0.4
    ms_hip.Model(step)._run_step(step, batch)
    assert step.__dict__["_lowered"] is False and "lr_power" in step._lowering_refused


# ---- the reference's own model code at CONFIGURATION size (tests/golden/make_ref_fixtures_cfgsize.py) ---------------------------------
def _cfgsize_init(comp):
    return {"init/" + k: RF.cfgsize_param(k, tuple(comp["shapes"][k]), comp["sigma"][k]) for k in comp["shapes"]}


def _check_summary(name, got, ref, rtol_sum, rtol_el, lr_steps):
    """sum / sum of squares / 32 elements of a dense parameter against what the reference's run left (Adam: an element whose gradient is
    ~0 may step +-lr either way, so elements are held to a few steps of lr on top of the relative bar)."""
    g = RF.cfgsize_summary(got)
    assert abs(g[1] - ref[1]) <= rtol_sum * abs(ref[1]) + 1e-12, (name, "sum of squares", g[1], ref[1])
    assert np.all(np.abs(g[2:] - ref[2:]) <= rtol_el * np.abs(ref[2:]) + lr_steps), (name, np.abs(g[2:] - ref[2:]).max())


@pytest.mark.timeout(600)
def test_deep_cross_engine_at_configuration_size_matches_the_reference_code(dev):
    """BASELINE configs[2] -- batch 16384, 39 x 30, 1170-1024-1024, 6 cross layers, fp32 -- run by the REFERENCE's DeepCrossModel /
    NetWithLossClass / TrainStepWrap (ref_dcn_cfg2.npz) against the product engine's large-shape dispatch (three-part GEMMs, batch
    slabs, the row-looking-up Adam over the table, the whole step as one graph from the third step on): losses, sampled table rows
    (touched and untouched), every dense parameter's sum of squares and 32 of its elements, evaluation logits."""
    from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine
    z, cfg, comp = RF.load("ref_dcn_cfg2")
    eng = DeepCrossEngine(DeepCrossConfig(vocab_size=cfg["vocab_size"], emb_dim=cfg["emb_dim"], field_size=cfg["field_size"],
                                          batch_size=cfg["batch_size"], deep_layer_dim=list(cfg["deep_layer_dim"]),
                                          cross_layer_num=cfg["cross_layer_num"], learning_rate=comp["lr"], eps=comp["eps"],
                                          loss_scale=comp["loss_scale"]), dev)
    assert eng._native
    RF.dcn_load_init(eng, _cfgsize_init(comp))
    ids, wts, label = RF.cfgsize_batches(comp["batch_seed"], comp["steps"], cfg["batch_size"], cfg["field_size"], cfg["vocab_size"])
    losses = np.array([float(eng.train_step(*(torch.from_numpy(a[s]).to(dev) for a in (ids, wts, label)))) for s in range(comp["steps"])])
    assert np.allclose(losses, z["loss"], rtol=1e-5, atol=0), (losses, z["loss"])
    st = RF.dcn_state(eng)
    table = st.pop("deep_embeddinglookup.embedding_table")
    assert np.array_equal(table[z["rows_free"]], z["table_free"]) or np.allclose(table[z["rows_free"]], z["table_free"], rtol=2e-4, atol=1e-7)
    lr_steps = 2.5 * comp["lr"] * comp["steps"]
    assert np.all(np.abs(table[z["rows_touched"]] - z["table_touched"]) <= 2e-4 * np.abs(z["table_touched"]) + lr_steps)
    assert np.mean(np.abs(table[z["rows_touched"]] - z["table_touched"]) <= 2e-4 * np.abs(z["table_touched"]) + 0.02 * comp["lr"]) > 0.99
    for k, v in st.items():
        _check_summary(k, v, z["sum/" + k], 1e-5, 2e-4, lr_steps)
    logit, _ = eng.predict(*(torch.from_numpy(a[-1]).to(dev) for a in (ids, wts)))
    assert np.allclose(logit.cpu().numpy().reshape(-1)[:256], z["eval_logits"], rtol=1e-3, atol=2e-4)


@pytest.mark.timeout(600)
def test_wide_deep_engine_at_the_benchmarked_shape_matches_the_reference_code(dev):
    """The benchmarked step's shape -- batch 16384, 39 fields, dim 80, the 1024-512-256-128 net in the reference's fp16, LazyAdam + FTRL
    on row gradients -- run by the REFERENCE's WideDeepModel / NetWithLossClass / TrainStepWrap (ref_wd_cfg1.npz; vocabulary 200 000)
    against the product engine with everything on: fused rows, the folded wide branch, the fused tail launch, graphs.  The fp16 net is
    held within the reference's own double rounding (its DenseLayer rounds the MatMul output and the bias sum separately, :121-126)."""
    from mindrec_amd.wide_deep import WideDeepEngine
    z, cfg, comp = RF.load("ref_wd_cfg1")
    eng = WideDeepEngine(RF.wd_config(cfg, comp), dev)
    assert eng._mfma and eng._fold_wide
    init = _cfgsize_init(comp)
    RF.wd_load_init(eng, init)
    ids, wts, label = RF.cfgsize_batches(comp["batch_seed"], comp["steps"], cfg["batch_size"], cfg["field_size"], cfg["vocab_size"])
    losses = np.array([float(eng.train_step(*(torch.from_numpy(a[s]).to(dev) for a in (ids, wts, label)))) for s in range(comp["steps"])])
    assert np.allclose(losses, z["loss_w"], rtol=5e-4, atol=0), (losses, z["loss_w"])
    deep, wide = eng.deep.cpu().numpy(), eng.wide.cpu().numpy()
    t, f = z["rows_touched"], z["rows_free"]
    assert np.array_equal(deep[f], z["deep_free"]) and np.array_equal(wide[f], z["wide_free"])       # LazyAdam / sparse FTRL: untouched rows do not move
    d0, w0 = init["init/embedding_table"], init["init/wide_embeddinglookup.embedding_table"]
    step = np.abs(z["deep_touched"] - d0[t]).max()
    assert np.abs(deep[t] - z["deep_touched"]).max() <= 0.15 * step                                    # the updates agree to 15 % of the largest update ...
    assert RF.row_rel(deep[t], z["deep_touched"]) <= 2e-4                                              # ... and the rows to 2e-4 of their own scale
    wstep = np.abs(z["wide_touched"] - w0[t]).max()
    assert np.abs(wide[t] - z["wide_touched"]).max() <= 0.15 * wstep
    for k, v in RF.wd_dense_state(eng).items():
        ref = z["sum/" + k]
        g = RF.cfgsize_summary(v)
        i0 = RF.cfgsize_summary(init["init/" + k])
        upd = np.abs(ref[2:] - i0[2:]).max()
        assert np.abs(g[2:] - ref[2:]).max() <= 0.15 * upd + 1e-7, (k, np.abs(g[2:] - ref[2:]).max(), upd)
    logit, _ = eng.predict(*(torch.from_numpy(a[-1]).to(dev) for a in (ids, wts)))
    assert np.allclose(logit.cpu().numpy().reshape(-1)[:256], z["eval_logits"], rtol=2e-2, atol=2e-3)
