"""CPU: the fixtures the REFERENCE's own Python produced (tests/golden/ref_*.npz, make_ref_fixtures.py) against (a) the
product engines' HOST logic driven by the oracle op set -- is the engines' composition (optimizer split, hyper-parameters, L2
term, sens, op order) the reference's? -- and (b) this repo's own `mindspore_rec` package over compat/mindspore with the same
CPU kernel set -- does it compute what the reference's package computes?  The GPU twin is tests/test_ref_fixtures_gpu.py."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_fixtures as RF  # noqa: E402


def test_reference_composition_record():
    """What the reference's TrainStepWrap did with its parameters when it ran over compat/mindspore (ref_composition.json)."""
    with open(os.path.join(RF.GOLDEN, "ref_composition.json")) as f:
        rep = json.load(f)
    assert sorted(rep["reference_ci_cases_passed"]) == ["test_online_learning_api_data_sink_mode_not_bool",
                                                        "test_online_learning_api_sink_size_is_negative",
                                                        "test_online_learning_api_sink_size_not_equal_one"]
    for case, opt_d in (("ref_wd_dense", "Adam"), ("ref_wd_sparse", "LazyAdam"), ("ref_wd_dynamic", "LazyAdam"), ("ref_wd_mixed", "LazyAdam")):
        c = rep[case]
        assert c["optimizer_d"] == opt_d and c["optimizer_w"] == "FTRL"
        assert c["weights_w"] == ["wide_b", "wide_embeddinglookup.embedding_table"]          # the wide bias belongs to FTRL
        assert (c["sens"], c["lr_d"], c["eps_d"], c["lr_w"], c["l1_w"], c["l2_w"], c["initial_accum_w"]) == (1024.0, 3.5e-4, 1e-8, 5e-2, 1e-8, 1e-8, 1.0)
    # mindrec_amd/lowering.py recognises the REFERENCE's train cells by structure (on the generator's host tensors it gets as far as
    # the device check; the hash-table model as far as the check that its MapParameters are the HIP ones)
    for case in ("ref_wd_dense", "ref_wd_sparse", "ref_wd_mixed", "ref_dcn", "ref_deepfm", "ref_deepfm_mixed"):
        assert rep[case]["lowering_on_cpu"] == "parameters are not on an MI355X", (case, rep[case]["lowering_on_cpu"])
    assert rep["ref_wd_dynamic"]["lowering_on_cpu"] == "the MapParameter's store is not the HIP one"
    assert rep["ref_wd_dense"]["no_l2loss"] is False and rep["ref_wd_sparse"]["no_l2loss"] is True
    assert (rep["ref_dcn"]["optimizer"], rep["ref_dcn"]["lr"], rep["ref_dcn"]["loss_scale"]) == ("Adam", 1e-4, 1000.0)
    for case in ("ref_deepfm", "ref_deepfm_mixed"):            # ModelBuilder.get_train_eval_net with default_config.yaml's train config
        c = rep[case]
        assert (c["optimizer"], c["lr"], c["eps"], c["loss_scale"], c["sens"], c["l2_coef"]) == ("Adam", 5e-4, 5e-8, 1024.0, 1024.0, 8e-5)
        assert c["weights"][:2] == ["fm_w", "embedding_table"] and len(c["weights"]) == 12


@pytest.mark.parametrize("case", ["ref_wd_sparse", "ref_wd_dense"])
def test_wide_deep_engine_host_logic_matches_reference(oracle, case):
    from _oracle_engine import OracleWideDeepEngine
    z, cfg, comp = RF.load(case)
    eng = OracleWideDeepEngine(RF.wd_config(cfg, comp), "cpu")
    RF.wd_load_init(eng, z)
    losses = RF.wd_replay(eng, z, "cpu")
    assert np.allclose(losses, z["loss_w"], rtol=2e-6, atol=0), (losses, z["loss_w"])
    if not cfg["sparse"]:
        # the deep optimizer's loss carries the L2 term at the step's starting values (wide_and_deep.py:356-360)
        assert np.isclose(eng.deep_loss(losses[-1]), z["loss_d"][-1], rtol=2e-6, atol=0) and z["loss_d"][-1] > z["loss_w"][-1]
    assert RF.row_rel(eng.deep.numpy(), z["final/embedding_table"]) <= 1e-5
    w_ref = z["final/wide_embeddinglookup.embedding_table"]
    assert np.abs(eng.wide.numpy() - w_ref).max() <= 1e-4 * np.abs(w_ref).max()
    for k, v in RF.wd_dense_state(eng).items():
        assert np.allclose(v, z["final/" + k], rtol=1e-4, atol=1e-7), k
    # optimizer state: Adam moments of the deep table, FTRL accumulators of the wide table and of wide_b
    assert np.allclose(eng.deep_m.numpy(), z["state/moment1/embedding_table"], rtol=1e-4, atol=1e-9)
    assert np.allclose(eng.wide_accum.numpy(), z["state/accum/wide_embeddinglookup.embedding_table"], rtol=1e-5, atol=0)
    assert np.allclose(float(eng.dense_m[eng._wb_off]), float(z["state/accum/wide_b"][0]), rtol=1e-5)


def test_deep_cross_engine_host_logic_matches_reference(oracle):
    from _oracle_engine import OracleDeepCrossEngine
    from mindrec_amd.deep_cross import DeepCrossConfig
    z, cfg, comp = RF.load("ref_dcn")
    eng = OracleDeepCrossEngine(DeepCrossConfig(vocab_size=cfg["vocab_size"], emb_dim=cfg["emb_dim"], field_size=cfg["field_size"],
                                                batch_size=cfg["batch_size"], deep_layer_dim=list(cfg["deep_layer_dim"]),
                                                cross_layer_num=cfg["cross_layer_num"]), "cpu")
    RF.dcn_load_init(eng, z)
    losses = np.array([float(eng.train_step(*(torch.from_numpy(z[k][s]) for k in ("ids", "wts", "label")))) for s in range(z["ids"].shape[0])])
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    for k, v in RF.dcn_state(eng).items():
        assert np.allclose(v, z["final/" + k], rtol=2e-4, atol=1e-7), k


def test_deepfm_engine_host_logic_matches_reference(oracle):
    """models/deepfm/src/deepfm.py (the reference's own DeepFMModel / NetWithLossClass / TrainStepWrap through ModelBuilder, run over
    compat/mindspore: ref_deepfm.npz) against the DeepFM engine's host logic on the oracle's primitives: one nn.Adam over BOTH
    tables and the net, the L2 term over the whole tables, loss scale 1024."""
    from _oracle_engine import OracleDeepFMEngine
    z, cfg, comp = RF.load("ref_deepfm")
    assert comp["optimizer"] == "Adam" and comp["weights"][:2] == ["fm_w", "embedding_table"] and comp["dropout_keep_prob_in_dense_layers"] == 1.0
    eng = OracleDeepFMEngine(RF.deepfm_config(cfg, comp), "cpu")
    RF.deepfm_load_init(eng, z)
    losses = np.array([float(eng.train_step(*(torch.from_numpy(z[k][s]) for k in ("ids", "wts", "label")))) for s in range(z["ids"].shape[0])])
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    for k, v in RF.deepfm_state(eng).items():
        assert np.allclose(v, z["final/" + k], rtol=2e-4, atol=1e-7), k
    assert np.allclose(eng.state["V"][0].numpy(), z["state/moment1/embedding_table"], rtol=1e-4, atol=1e-9)


# ---- this repo's own mindspore_rec + a mindspore-style script over compat/mindspore, CPU kernel set -------------------------------
@pytest.fixture
def ms_cpu(oracle):
    """compat/ on the path, host tensors, the oracle-backed kernel set installed for the duration of one test."""
    compat = os.path.join(os.path.dirname(RF.GOLDEN.rstrip("/")), "..", "compat")
    sys.path.insert(0, os.path.abspath(compat))
    import mindspore
    from mindspore import context
    import _ms_cpu_kernels
    prev = mindspore._kernels._install(_ms_cpu_kernels)
    prev_target = context.get_context("device_target")
    context.set_context(device_target="CPU")
    yield mindspore
    context.set_context(device_target=prev_target)
    mindspore._kernels._install(prev)


def test_own_hash_embedding_lookup_matches_reference_package(ms_cpu):
    """compat/mindspore_rec.HashEmbeddingLookup (one fused MapTensorGet) == the reference's Unique -> MapTensorGet -> Gather chain
    (mindspore_rec/ops/embedding.py:184-206) recorded in ref_hash_lookup.npz: outputs, inserted keys and rows, bit for bit."""
    import mindspore_rec
    assert "compat" in mindspore_rec.__file__
    z = np.load(os.path.join(RF.GOLDEN, "ref_hash_lookup.npz"))
    for i, v in enumerate(json.loads(str(z["variants"]))):
        kd = ms_cpu.int32 if "int32" in v["key_dtype"] else ms_cpu.int64
        ms_cpu.set_seed(300 + i)
        layer = mindspore_rec.HashEmbeddingLookup(embedding_size=v["D"], key_dtype=kd, sparse=v["sparse"], max_norm=v["max_norm"])
        assert layer.embedding_table.seed == v["seed"]
        for c in range(2):
            out = layer(ms_cpu.Tensor(z[f"v{i}/keys"][c])).asnumpy()
            ref = z[f"v{i}/out{c}"]
            assert out.shape == ref.shape
            assert np.array_equal(out, ref) if v["max_norm"] is None else np.allclose(out, ref, rtol=1e-6, atol=0)
        k, vals = layer.embedding_table.get_data()
        order = np.argsort(k.asnumpy())
        assert np.array_equal(k.asnumpy()[order], z[f"v{i}/table_keys"]) and np.array_equal(vals.asnumpy()[order], z[f"v{i}/table_values"])


def test_mindspore_style_deep_cross_script_matches_reference(ms_cpu):
    import _ms_models
    z, cfg, comp = RF.load("ref_dcn")
    step, net = _ms_models.deep_cross_from_fixture(z, cfg, comp)
    losses = [float(step(ms_cpu.Tensor(z["ids"][s]), ms_cpu.Tensor(z["wts"][s]), ms_cpu.Tensor(z["label"][s])).asnumpy()) for s in range(z["ids"].shape[0])]
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    assert np.allclose(net.out.weight.asnumpy(), z["final/dense_layer_3.weight"], rtol=2e-4, atol=1e-7)
    assert np.allclose(net.cross3.cross_weight.asnumpy(), z["final/cross_layer_4.cross_weight"], rtol=2e-4, atol=1e-7)
    from mindrec_amd import lowering
    assert lowering.lower_train_step(step) is None and step._lowering_refused == "parameters are not on an MI355X"


@pytest.mark.parametrize("case", ["ref_deepfm", "ref_deepfm_mixed"])
def test_mindspore_style_deepfm_script_matches_reference(ms_cpu, case):
    import _ms_models
    z, cfg, comp = RF.load(case)
    step, net = _ms_models.deepfm_from_fixture(z, cfg, comp)
    losses = [float(step(ms_cpu.Tensor(z["ids"][s]), ms_cpu.Tensor(z["wts"][s]), ms_cpu.Tensor(z["label"][s])).asnumpy()) for s in range(z["ids"].shape[0])]
    assert np.allclose(losses, z["loss"], rtol=2e-6, atol=0), (losses, z["loss"])
    assert np.allclose(net.factors.asnumpy(), z["final/embedding_table"], rtol=2e-4, atol=1e-7)
    assert np.allclose(net.linear.asnumpy(), z["final/fm_w"], rtol=2e-4, atol=1e-7)
    assert np.allclose(net.layer0.weight.asnumpy(), z["final/dense_layer_1.weight"], rtol=2e-4, atol=1e-7)
    from mindrec_amd import lowering
    assert lowering.lower_train_step(step) is None and step._lowering_refused == "parameters are not on an MI355X"


@pytest.mark.parametrize("case", ["ref_wd_sparse", "ref_wd_dense", "ref_wd_dynamic", "ref_wd_mixed"])
def test_mindspore_style_script_matches_reference(ms_cpu, case):
    """A Wide&Deep train step written against the compat surface (tests/_ms_models.py, own code) reproduces what the reference's
    model code computed, including hash tables created on first sight (ref_wd_dynamic) and the fp16 DenseLayers (ref_wd_mixed)."""
    import _ms_models
    z, cfg, comp = RF.load(case)
    if cfg["dynamic_embedding"]:
        ms_cpu.set_seed(1000)
        from mindspore.common import initializer as I
        I._state["calls"] = int(z["deep_seed"]) - (1000 * 1_000_003) - 1        # the next table takes the fixture's seed
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)
    assert sorted(p.name.split(".")[-1] + "@" + p.name.split(".")[-2] for p in step.w_wide) == ["embedding_table@wide_table", "wide_bias@net"]
    losses = []
    for s in range(z["ids"].shape[0]):
        lw, ld = step(ms_cpu.Tensor(z["ids"][s]), ms_cpu.Tensor(z["wts"][s]), ms_cpu.Tensor(z["label"][s]))
        losses.append((float(lw.asnumpy()), float(ld.asnumpy())))
    losses = np.array(losses)
    assert np.allclose(losses[:, 0], z["loss_w"], rtol=1e-6, atol=0) and np.allclose(losses[:, 1], z["loss_d"], rtol=1e-6, atol=0)
    assert np.allclose(net.wide_bias.asnumpy(), z["final/wide_b"], rtol=1e-5, atol=1e-9)
    if cfg["dynamic_embedding"]:
        k, v = net.deep_table.embedding_table.get_data()
        order = np.argsort(k.asnumpy())
        assert np.array_equal(k.asnumpy()[order], z["final/embedding_table::keys"])
        assert RF.row_rel(v.asnumpy()[order], z["final/embedding_table::values"]) <= 1e-5
    else:
        assert RF.row_rel(net.deep_table.embedding_table.asnumpy(), z["final/embedding_table"]) <= 1e-5


def test_hash_table_checkpoint_restores_before_the_optimizer_has_run(ms_cpu, tmp_path):
    """A fresh network restored from a checkpoint holds optimizer slots the optimizer has not created yet.  Rows of keys first seen
    AFTER the restore must still start from the optimizer's own initial values (FTRL's accumulator: initial_accum, not 0): six
    steps straight == three steps, save, restore into a fresh network, three steps -- on keys that keep arriving."""
    import _ms_models
    from mindspore.train.serialization import load_checkpoint, load_param_into_net, save_checkpoint
    Bq, Fq, Dq = 16, 5, 4

    def build():
        ms_cpu.set_seed(1000)
        net = _ms_models.WideDeep(10_000, Dq, Fq, Bq, [8, 4], sparse=True, dynamic=True, capacity=1024)
        step = _ms_models.WideDeepTrainStep(_ms_models.WideDeepLoss(net, 8e-5, with_l2=False), lazy=True)
        step.set_train()
        return step, net

    def batch(t):
        rng = np.random.default_rng(500 + t)
        ids = (rng.integers(0, 40, size=(Bq, Fq)) + 25 * t).astype(np.int32)          # the key range moves: new keys every step
        return (ms_cpu.Tensor(ids), ms_cpu.Tensor(rng.random((Bq, Fq)).astype(np.float32)),
                ms_cpu.Tensor((rng.random((Bq, 1)) < 0.3).astype(np.float32)))

    a, _ = build()
    straight = [float(a(*batch(t))[0].asnumpy()) for t in range(6)]
    b, _ = build()
    first = [float(b(*batch(t))[0].asnumpy()) for t in range(3)]
    ck = str(tmp_path / "half.ckpt")
    save_checkpoint(b, ck)
    c, _ = build()
    assert load_param_into_net(c, load_checkpoint(ck)) == []
    rest = [float(c(*batch(t))[0].asnumpy()) for t in range(3, 6)]
    assert first + rest == straight


def test_train_and_eval_flow_matches_the_reference_script(oracle, tmp_path):
    """ref_train_eval_flow.npz was written by the reference's OWN `test_train_eval(config)` (models/wide_deep/train_and_eval.py:66-104,
    run as it is over compat/mindspore with its src/callbacks.py and src/metrics.py; only the MindRecord reader was replaced).  The
    engine-level flow -- RecModel(WideDeepRunner(engine)) + this repo's LossCallBack / EvalCallBack / AUCMetric -- writes the same
    loss.log and eval.log: line format, epoch / step numbering, losses to 2e-6, the AUC after every epoch."""
    from _oracle_engine import OracleWideDeepEngine
    z, cfg, comp = RF.load("ref_train_eval_flow")
    assert json.loads(str(z["ckpts"])) == ["widedeep_train-1_4.ckpt", "widedeep_train-2_4.ckpt"]       # ModelCheckpoint(save_checkpoint_steps = steps per epoch)
    # the reference's eval.py (test_eval, :67-115, as it is) on the last of those checkpoints saw what the training run's last EvalCallBack saw
    assert float(z["eval_py_auc"]) == float(z["auc"][-1])
    eng = OracleWideDeepEngine(RF.wd_config(cfg, comp), "cpu")
    RF.wd_load_init(eng, z)
    RF.check_train_eval_flow(z, RF.run_train_eval_flow(eng, z, "cpu", str(tmp_path)))


def test_reference_data_parallel_run_equals_one_engine_on_the_concatenated_batch(oracle):
    """ref_wd_dp2.npz: models/wide_deep/train_and_eval_distribute.py run as it is by TWO processes (compat/mindspore over
    torch.distributed's gloo; its own sharded TFRecord reader, init(), ParallelMode.DATA_PARALLEL with gradients_mean, two
    DistributedGradReducers).  What the reference's data parallelism computes is what ONE engine computes on the concatenation of the
    ranks' batches -- the equivalence this repo's multi-GPU path is built on (tests/test_wide_deep_dist.py proves the row-sharded engine
    equal to the same single-process step): losses = the mean of the ranks' losses, parameters equal to summation order."""
    from _oracle_engine import OracleWideDeepEngine
    RF.check_data_parallel_fixture(lambda c: OracleWideDeepEngine(c, "cpu"), "cpu")


def test_checkpoint_names_are_mindspore_names(ms_cpu, tmp_path):
    """optimizer slots are saved under MindSpore's names (`moment1.<parameter>`, `accum.<parameter>`; the step scalars of
    the two optimizers under their cell path) -- `filter_prefix='moment1'` means what it means there -- and a file with the older
    cell-path names still loads."""
    import _ms_models
    from mindspore.train.serialization import load_checkpoint, load_param_into_net, save_checkpoint
    z, cfg, comp = RF.load("ref_wd_sparse")
    step, net = _ms_models.wide_deep_from_fixture(z, cfg, comp)
    batch = tuple(ms_cpu.Tensor(z[k][0]) for k in ("ids", "wts", "label"))
    step(*batch)
    path = save_checkpoint(step, str(tmp_path / "a.ckpt"))
    names = list(np.load(path).files)
    m1 = step.opt_deep._slot(net.layer0.weight, "moment1", 0.0).name          # `moment1.<the parameter's name when the optimizer was built>`
    w_name = m1[len("moment1."):]
    assert m1.startswith("moment1.") and m1 in names and f"moment2.{w_name}" in names and any(n.startswith("accum.") for n in names)
    assert "opt_deep.global_step" in names and "opt_wide.global_step" in names and "opt_deep.beta1_power" in names
    kept = load_checkpoint(path, filter_prefix="moment1")
    assert not any(k.startswith("moment1") for k in kept) and f"moment2.{w_name}" in kept
    # a second, fresh model takes everything back -- from this file and from one with the pre-round-5 names
    ref_m = np.load(path)[f"moment1.{w_name}"]
    old = {(("opt_deep." + k) if k.startswith(("moment1.", "moment2.")) else ("opt_wide." + k) if k.startswith(("accum.", "linear.")) else k): v
           for k, v in np.load(path).items()}
    np.savez(open(tmp_path / "old.ckpt", "wb"), **old)
    for f in ("a.ckpt", "old.ckpt"):
        step2, net2 = _ms_models.wide_deep_from_fixture(z, cfg, comp)
        step2(*batch)                                                         # (creates the optimizer slots)
        missing = load_param_into_net(step2, load_checkpoint(str(tmp_path / f)))
        assert not missing, (f, missing)
        assert np.array_equal(step2.opt_deep._slot(net2.layer0.weight, "moment1", 0.0).asnumpy(), ref_m)
        assert step2.opt_deep.global_step == 1
