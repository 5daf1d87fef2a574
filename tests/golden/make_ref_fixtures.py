"""BUILD-CONTAINER-ONLY generator of tests/golden/ref_*.npz: runs the REFERENCE's own Python -- unmodified, imported from
/root/reference -- and records what it computes.

What runs: models/wide_deep/src/wide_and_deep.py (WideDeepModel, NetWithLossClass, TrainStepWrap, PredictWithSigmoid :136-518),
models/deepfm/src/deepfm.py (DeepFMModel, NetWithLossClass, TrainStepWrap, PredictWithSigmoid, ModelBuilder :152-370),
models/deep_and_cross/src/deep_and_cross.py (DeepCrossModel, NetWithLossClass, TrainStepWrap :206-354),
mindspore_rec/ops/embedding.py (HashEmbeddingLookup :47-206), mindspore_rec/train/rec_model.py (RecModel :34-309) and the three
cases of ci/st/online_learning/test_online_learning.py:54-114.

What they run ON: `compat/mindspore` (this repo's own mindspore-named API; MindSpore itself is not installable here) with the
CPU kernel set tests/_ms_cpu_kernels.py = the oracle's restatements of the primitives.  So the fixtures pin the reference's
COMPOSITION -- which parameter goes to which optimizer, which hyper-parameters, the op order of construct, the L2 term, the
sens seeding, the three-forward step -- by the reference's code; the per-primitive formulas underneath (Adam, FTRL, Unique,
MapTensorGet defaults, initializer streams) stay this repo's restatement of MindSpore's published semantics [EXT].

Nothing under /root/reference travels: only the .npz files written here are committed.  Usage (build container):
    python tests/golden/make_ref_fixtures.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MREC_REFERENCE", "/root/reference")
for p in (os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "compat"), REF):
    if p not in sys.path:
        sys.path.insert(0, p)                   # REF first: `mindspore_rec` resolves to the REFERENCE's package

import mindspore  # noqa: E402
from mindspore import Tensor, context  # noqa: E402

import _ms_cpu_kernels  # noqa: E402

mindspore._kernels._install(_ms_cpu_kernels)
context.set_context(mode=context.GRAPH_MODE, device_target="CPU")

import mindspore_rec  # noqa: E402

assert mindspore_rec.__file__.startswith(REF), mindspore_rec.__file__


def _ref_module(model_dir, name):
    """Imports <REF>/models/<model_dir>/src/<name>.py as its own package tree (both models call their package `src`)."""
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, "models", model_dir))
    try:
        return importlib.import_module("src." + name)
    finally:
        sys.path.pop(0)


def _np(t):
    return t.asnumpy().copy() if hasattr(t, "asnumpy") else np.asarray(t)


def _batches(rng, S, B, F, V, keys=None):
    """Criteo-like: the first 3 fields are the constants 0..2 with fractional weights (datasets/criteo_1tb/process_data.py:138-147),
    the rest Zipf-distributed categorical ids with weight 1 (:149-162)."""
    ids = np.minimum(rng.zipf(1.3, size=(S, B, F)) + 2, V - 1).astype(np.int32)
    ids[:, :, :3] = np.arange(3, dtype=np.int32)
    wts = np.ones((S, B, F), np.float32)
    wts[:, :, :3] = rng.random((S, B, 3)).astype(np.float32)
    label = (rng.random((S, B, 1)) < 0.3).astype(np.float32)
    if keys is not None:
        ids = keys(ids)
    return ids, wts, label


def _wd_composition(train, loss_net, struct):
    """What the reference's TrainStepWrap did with the parameters: which optimizer owns which, with which hyper-parameters."""
    return {"optimizer_w": type(train.optimizer_w).__name__, "optimizer_d": type(train.optimizer_d).__name__,
            "weights_w": sorted(k for k, p in struct.items() if any(p is q for q in train.weights_w)),
            "weights_d": sorted(k for k, p in struct.items() if any(p is q for q in train.weights_d)),
            "sens": float(train.sens), "lr_d": train.optimizer_d.get_lr(), "lr_w": train.optimizer_w.get_lr(),
            "eps_d": train.optimizer_d.eps, "l1_w": train.optimizer_w.l1, "l2_w": train.optimizer_w.l2,
            "initial_accum_w": train.optimizer_w.initial_accum, "loss_scale_d": train.optimizer_d.loss_scale,
            "no_l2loss": bool(loss_net.no_l2loss), "l2_coef": float(loss_net.l2_coef)}


def wide_deep_case(name, S=3, mixed=False, key_dtype=np.int32, **mode):
    wd = _ref_module("wide_deep", "wide_and_deep")
    cfg = types.SimpleNamespace(batch_size=64, field_size=9, emb_dim=8, vocab_size=3000, vocab_cache_size=0,
                                deep_layer_dim=[32, 16, 16, 8], deep_layer_act="relu", keep_prob=1.0, dropout_flag=False,
                                use_mixed_precision=bool(mixed), parameter_server=0, sparse=False, dynamic_embedding=False,
                                weight_bias_init=["normal", "normal"], emb_init="normal", init_args=[-0.01, 0.01], l2_coef=8e-5,
                                full_batch=False, field_slice=False)
    for k, v in mode.items():
        setattr(cfg, k, v)
    mindspore.set_seed(1000)
    net = wd.WideDeepModel(cfg)
    loss_net = wd.NetWithLossClass(net, cfg)
    train = wd.TrainStepWrap(loss_net, parameter_server=bool(cfg.parameter_server), sparse=cfg.sparse,
                             dynamic_embedding=cfg.dynamic_embedding)
    evaln = wd.PredictWithSigmoid(net)
    train.set_train()
    out = {}
    struct = dict(net.parameters_and_names())
    dyn = bool(cfg.dynamic_embedding)
    comp = _wd_composition(train, loss_net, struct)
    # does the structural recogniser of mindrec_amd/lowering.py (GRAPH_MODE's compile step on an MI355X) see this -- the
    # REFERENCE's -- train cell for what it is?  Here (host tensors) it must get as far as the device check
    from mindrec_amd import lowering
    assert lowering.lower_train_step(train) is None
    comp["lowering_on_cpu"] = train._lowering_refused
    rng = np.random.default_rng(20240 + len(name))
    keys = (lambda i: (i.astype(np.int64) * 2654435761 + 17).astype(key_dtype) if key_dtype == np.int64 else i * 7 + 100) if dyn else None
    ids, wts, label = _batches(rng, S, cfg.batch_size, cfg.field_size, cfg.vocab_size, keys)
    for k, p in struct.items():
        if not dyn or "embedding_table" not in k:
            out["init/" + k] = _np(p)
    if dyn:
        out["deep_seed"], out["wide_seed"] = np.int64(net.deep_embeddinglookup.embedding_table.seed), np.int64(net.wide_embeddinglookup.embedding_table.seed)
        assert int(out["wide_seed"]) == int(out["deep_seed"]) + 1
    lw, ld = [], []
    for s in range(S):
        a, b = train(Tensor(ids[s]), Tensor(wts[s]), Tensor(label[s]))
        lw.append(float(_np(a)))
        ld.append(float(_np(b)))
    evaln.set_train(False)
    logits, probs, _ = evaln(Tensor(ids[S - 1]), Tensor(wts[S - 1]), Tensor(label[S - 1]))
    for k, p in struct.items():
        if dyn and "embedding_table" in k:
            kk, vv = p.get_data()
            order = np.argsort(_np(kk))
            out["final/" + k + "::keys"], out["final/" + k + "::values"] = _np(kk)[order], _np(vv)[order]
        else:
            out["final/" + k] = _np(p)
    for opt, slots in ((train.optimizer_d, ("moment1", "moment2")), (train.optimizer_w, ("accum", "linear"))):
        for k, p in struct.items():
            if any(p is q for q in opt.parameters) and not (dyn and "embedding_table" in k):
                for sl in slots:
                    out[f"state/{sl}/{k}"] = _np(opt._slot(p, sl, 0.0))
    out.update(ids=ids, wts=wts, label=label, loss_w=np.array(lw, np.float64), loss_d=np.array(ld, np.float64),
               eval_logits=_np(logits), eval_probs=_np(probs),
               cfg=np.array(json.dumps({k: v for k, v in vars(cfg).items()})), composition=np.array(json.dumps(comp)))
    _save(name, out)
    return comp


def deep_cross_case(name, S=3):
    dcn = _ref_module("deep_and_cross", "deep_and_cross")
    cfg = types.SimpleNamespace(batch_size=64, field_size=5, emb_dim=6, vocab_size=400, deep_layer_dim=[16, 8], cross_layer_num=6,
                                keep_prob=1.0)
    mindspore.set_seed(1000)
    net = dcn.DeepCrossModel(cfg)
    loss_net = dcn.NetWithLossClass(net)
    train = dcn.TrainStepWrap(loss_net)                       # lr 1e-4, eps 1e-8, loss_scale 1000 (deep_and_cross.py:336)
    evaln = dcn.PredictWithSigmoid(net)
    train.set_train()
    struct = dict(net.parameters_and_names())
    out = {"init/" + k: _np(p) for k, p in struct.items()}
    rng = np.random.default_rng(77)
    ids, wts, label = _batches(rng, S, cfg.batch_size, cfg.field_size, cfg.vocab_size)
    losses = [float(_np(train(Tensor(ids[s]), Tensor(wts[s]), Tensor(label[s])))) for s in range(S)]
    evaln.set_train(False)
    logits, probs, _ = evaln(Tensor(ids[S - 1]), Tensor(wts[S - 1]), Tensor(label[S - 1]))
    for k, p in struct.items():
        out["final/" + k] = _np(p)
    comp = {"optimizer": type(train.optimizer).__name__, "lr": train.optimizer.get_lr(), "eps": train.optimizer.eps,
            "loss_scale": train.optimizer.loss_scale, "sens": float(train.sens), "weights": list(struct)}
    from mindrec_amd import lowering
    assert lowering.lower_train_step(train) is None
    comp["lowering_on_cpu"] = train._lowering_refused
    out.update(ids=ids, wts=wts, label=label, loss=np.array(losses, np.float64), eval_logits=_np(logits), eval_probs=_np(probs),
               cfg=np.array(json.dumps(vars(cfg))), composition=np.array(json.dumps(comp)))
    _save(name, out)
    return comp


def deepfm_case(name, S=3, convert_dtype=False):
    """models/deepfm/src/deepfm.py: DeepFMModel + NetWithLossClass + TrainStepWrap + PredictWithSigmoid through ModelBuilder
    (:322-370), with the hyper-parameters of models/deepfm/default_config.yaml:27-33."""
    fm = _ref_module("deepfm", "deepfm")
    mc = types.SimpleNamespace(batch_size=64, data_field_size=7, data_vocab_size=500, data_emb_dim=8, deep_layer_args=[[32, 16, 16, 8], "relu"],
                               init_args=[-0.01, 0.01], weight_bias_init=["normal", "normal"], keep_prob=0.9, convert_dtype=bool(convert_dtype))
    tc = types.SimpleNamespace(l2_coef=8e-5, learning_rate=5e-4, epsilon=5e-8, loss_scale=1024.0)
    mindspore.set_seed(1000)
    np.random.seed(1000)                                      # (the model draws its initial values from numpy's global generator, :74,92)
    train, evaln = fm.ModelBuilder(mc, tc).get_train_eval_net()
    net = evaln.network
    train.set_train()
    struct = dict(net.parameters_and_names())
    out = {"init/" + k: _np(p) for k, p in struct.items()}
    rng = np.random.default_rng(4242)
    ids, wts, label = _batches(rng, S, mc.batch_size, mc.data_field_size, mc.data_vocab_size)
    losses = [float(_np(train(Tensor(ids[s]), Tensor(wts[s]), Tensor(label[s])))) for s in range(S)]
    evaln.set_train(False)
    logits, probs, _ = evaln(Tensor(ids[S - 1]), Tensor(wts[S - 1]), Tensor(label[S - 1]))
    for k, p in struct.items():
        out["final/" + k] = _np(p)
    opt = train.optimizer
    for k, p in struct.items():
        for sl in ("moment1", "moment2"):
            out[f"state/{sl}/{k}"] = _np(opt._slot(p, sl, 0.0))
    comp = {"optimizer": type(opt).__name__, "lr": opt.get_lr(), "eps": opt.eps, "loss_scale": opt.loss_scale, "sens": float(train.sens),
            "l2_coef": float(train.network.l2_coef), "weights": list(struct), "convert_dtype": bool(convert_dtype),
            "dropout_keep_prob_in_dense_layers": 1.0 - float(getattr(net.dense_layer_1.dropout, "p", 0.0))}
    from mindrec_amd import lowering
    assert lowering.lower_train_step(train) is None
    comp["lowering_on_cpu"] = train._lowering_refused
    out.update(ids=ids, wts=wts, label=label, loss=np.array(losses, np.float64), eval_logits=_np(logits), eval_probs=_np(probs),
               cfg=np.array(json.dumps({**vars(mc), **vars(tc)})), composition=np.array(json.dumps(comp)))
    _save(name, out)
    return comp


def train_eval_flow_case(name):
    """models/wide_deep/train_and_eval.py: `test_train_eval(config)` (:66-104) run AS IT IS -- its own `create_dataset` (src/datasets.py:
    226-271, the TFRecord reader: Schema, TFRecordDataset, batch(batch_size / 1000), map(padding function)), Model(train_net,
    eval_network, metrics={"auc": AUCMetric()}), EvalCallBack, LossCallBack, ModelCheckpoint, TimeMonitor, model.train(epochs, ds_train,
    ..., dataset_sink_mode=True), with its own src/callbacks.py and src/metrics.py.  Nothing of the script is replaced: the configuration
    is pointed at TFRecord files (`dataset_type: tfrecord`) that this repo's Criteo writer produced (mindrec_amd.criteo.write_tfrecords:
    1000 samples per row, as the reference's data preparation packs them).  A look-only wrapper around ModelBuilder.get_net records the
    initial parameters.  Recorded: what the callbacks wrote (loss.log, eval.log), the checkpoints' names."""
    import re
    import shutil
    import tempfile
    import mindspore.dataset as ds
    from mindrec_amd import criteo
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    wd_dir = os.path.join(REF, "models", "wide_deep")
    sys.path.insert(0, wd_dir)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    os.environ.setdefault("DEVICE_ID", "0")
    try:
        te = importlib.import_module("train_and_eval")
    finally:
        sys.argv = argv
        sys.path.remove(wd_dir)
    assert te.__file__.startswith(REF) and te.create_dataset.__module__ == "src.datasets"
    work = tempfile.mkdtemp(prefix="ref_flow_")
    cfg = te.cfg
    B, F, V, steps, n_eval, epochs = 1000, 39, 3000, 4, 2, 2              # (the reader's rows hold 1000 samples of 39 fields)
    for k, v in dict(batch_size=B, field_size=F, emb_dim=8, vocab_size=V, deep_layer_dim=[32, 16, 16, 8], epochs=epochs, sparse=False,
                     use_mixed_precision=False, dynamic_embedding=False, parameter_server=0, vocab_cache_size=0, dropout_flag=False,
                     dataset_type="tfrecord", data_path=os.path.join(work, "data"),
                     ckpt_path=os.path.join(work, "ckpt"), loss_file_name=os.path.join(work, "loss.log"),
                     eval_file_name=os.path.join(work, "eval.log")).items():
        setattr(cfg, k, v)
    rng = np.random.default_rng(31337)
    ids, wts, label = _batches(rng, steps + n_eval, B, F, V)
    ids[:, :, :13] = np.arange(13, dtype=np.int32)                        # Criteo's 13 dense fields: constant ids, the value as the weight
    wts[:, :, 3:13] = rng.random((steps + n_eval, B, 10)).astype(np.float32)
    label = (rng.random(label.shape) < 1.0 / (1.0 + np.exp(-(ids[..., 13:14] % 7 - 3.0)))).astype(np.float32)       # a learnable signal
    flat = lambda a, lo, hi: a[lo:hi].reshape((hi - lo) * B, -1)          # noqa: E731
    criteo.write_tfrecords(cfg.data_path, "train", flat(ids, 0, steps), flat(wts, 0, steps), flat(label, 0, steps), records_per_file=3)
    criteo.write_tfrecords(cfg.data_path, "test", flat(ids, steps, steps + n_eval), flat(wts, steps, steps + n_eval),
                           flat(label, steps, steps + n_eval), records_per_file=3)
    seen = {}
    build = te.ModelBuilder.get_net

    def spying_get_net(self, config):                         # (looks, does not touch: the initial parameters are part of the fixture)
        train_net, eval_net = build(self, config)
        net = eval_net.network
        struct = dict(net.parameters_and_names())
        seen["init"] = {"init/" + k: _np(p) for k, p in struct.items()}
        seen["comp"] = _wd_composition(train_net, train_net.network, struct)
        return train_net, eval_net

    te.ModelBuilder.get_net = spying_get_net
    mindspore.set_seed(1000)
    ds.config.set_seed(1000)
    # the order the reference's shuffling reader hands the training rows out in (its own dataset object, read once more)
    order = []
    for epoch in range(epochs):
        d = te.create_dataset(cfg.data_path, train_mode=True, batch_size=B, data_type=te.DataType.TFRECORD)
        for _ in range(epoch):
            d.reset()
        for bi, wi, li in d:
            order.append(int(np.flatnonzero((ids[:steps, 0] == np.asarray(bi)[0]).all(axis=1) & np.isclose(wts[:steps, 0], np.asarray(wi)[0]).all(axis=1))[0]))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        te.test_train_eval(cfg)
    finally:
        os.chdir(cwd)
    loss_lines = open(cfg.loss_file_name).read().strip().splitlines()
    eval_lines = [re.sub(r"^.*?== Rank", "== Rank", ln) for ln in open(cfg.eval_file_name).read().strip().splitlines()]
    eval_lines = [re.sub(r"eval_time: \d+s", "eval_time: Ns", ln) for ln in eval_lines]
    ckpts = sorted(f for f in os.listdir(cfg.ckpt_path) if f.endswith(".ckpt"))
    aucs = [float(re.search(r"dict_values\(\[([0-9.eE+-]+)\]\)", ln).group(1)) for ln in eval_lines]
    # ... and the reference's evaluation script on the last checkpoint: models/wide_deep/eval.py:test_eval(config) (:67-115) as it is --
    # load_checkpoint, load_param_into_net(eval_net), Model.eval with its EvalCallBack -- must see what training left
    import contextlib
    import io
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, wd_dir)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        ev = importlib.import_module("eval")
    finally:
        sys.argv = argv
        sys.path.remove(wd_dir)
    assert ev.__file__.startswith(REF)
    for k in ("batch_size", "field_size", "emb_dim", "vocab_size", "deep_layer_dim", "sparse", "use_mixed_precision", "dynamic_embedding",
              "parameter_server", "vocab_cache_size", "dropout_flag", "dataset_type", "data_path", "loss_file_name"):
        setattr(ev.cfg, k, getattr(cfg, k))
    ev.cfg.eval_file_name = os.path.join(work, "eval_py.log")
    ev.cfg.ckpt_path = os.path.join(cfg.ckpt_path, ckpts[-1])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ev.test_eval(ev.cfg)
    eval_py_auc = float(re.findall(r"auc: ([0-9.eE+-]+)", buf.getvalue())[-1])
    assert eval_py_auc == aucs[-1], (eval_py_auc, aucs)
    shutil.rmtree(work, ignore_errors=True)
    out = dict(ids=ids, wts=wts, label=label, n_train_steps=np.int64(steps), n_eval_steps=np.int64(n_eval), epochs=np.int64(epochs),
               train_order=np.array(order, np.int64),
               loss_log=np.array(json.dumps(loss_lines)), eval_log=np.array(json.dumps(eval_lines)), ckpts=np.array(json.dumps(ckpts)),
               auc=np.array(aucs, np.float64), eval_py_auc=np.float64(eval_py_auc), composition=np.array(json.dumps(seen["comp"])),
               cfg=np.array(json.dumps({k: getattr(cfg, k) for k in ("batch_size", "field_size", "emb_dim", "vocab_size", "deep_layer_dim", "epochs",
                                                                     "sparse", "use_mixed_precision", "l2_coef", "keep_prob", "dropout_flag",
                                                                     "dynamic_embedding", "vocab_cache_size", "parameter_server", "dataset_type")})))
    out.update(seen["init"])
    _save(name, out)
    return {"loss_log": loss_lines, "eval_log": eval_lines, "ckpts": ckpts, "auc": aucs, "train_order": order, "eval_py_auc": eval_py_auc}


def _dp_worker(rank, world, port, work, cfg_over):
    """One rank of models/wide_deep/train_and_eval_distribute.py, run as it is (init(), DATA_PARALLEL with gradients_mean, its sharded
    TFRecord reader, two DistributedGradReducers, its callbacks) under torch.distributed's gloo backend."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                      DEVICE_ID=str(rank))
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, "models", "wide_deep"))
    sys.argv = [sys.argv[0]]
    ted = importlib.import_module("train_and_eval_distribute")
    assert ted.__file__.startswith(REF)
    cfg = ted.cfg
    for k, v in cfg_over.items():
        setattr(cfg, k, v)
    cfg.loss_file_name, cfg.eval_file_name = os.path.join(work, f"loss{rank}.log"), os.path.join(work, f"eval{rank}.log")
    seen = {}
    build = ted.ModelBuilder.get_net

    def spying_get_net(self, config):
        train_net, eval_net = build(self, config)
        seen["net"], seen["train"] = eval_net.network, train_net
        seen["init"] = {k: _np(p) for k, p in seen["net"].parameters_and_names() if not hasattr(p, "get_data")}
        seen["seeds"] = {k: int(p.seed) for k, p in seen["net"].parameters_and_names() if hasattr(p, "get_data")}
        return train_net, eval_net

    ted.ModelBuilder.get_net = spying_get_net
    os.chdir(work)
    ted.train_wide_and_deep()
    import mindspore.dataset as ds
    ds.config.set_seed(ds.config.get_seed())
    from mindspore.communication.management import get_group_size, get_rank
    d = ted.create_dataset(cfg.data_path, train_mode=True, batch_size=cfg.batch_size, rank_id=get_rank(), rank_size=get_group_size(),
                           data_type=ted.DataType.TFRECORD)
    batches = [tuple(np.asarray(c) for c in row) for row in d]            # what this rank's reader handed out in its (one) epoch
    struct = dict(seen["net"].parameters_and_names())
    out = {f"rank{rank}/ids": np.stack([b[0] for b in batches]), f"rank{rank}/wts": np.stack([b[1] for b in batches]),
           f"rank{rank}/label": np.stack([b[2] for b in batches])}
    out.update({f"rank{rank}/init/{k}": v for k, v in seen["init"].items()})
    for k, p in struct.items():
        if hasattr(p, "get_data"):                            # a hash table: its (key, row) pairs, by key
            kk, vv = p.get_data()
            order = np.argsort(_np(kk))
            out[f"rank{rank}/final/{k}::keys"], out[f"rank{rank}/final/{k}::values"] = _np(kk)[order], _np(vv)[order]
            out[f"rank{rank}/seed/{k}"] = np.int64(seen["seeds"][k])
        else:
            out[f"rank{rank}/final/{k}"] = _np(p)
    comp = _wd_composition(seen["train"], seen["train"].network, struct)
    comp["reducer_flag"], comp["gradients_mean"], comp["degree"] = bool(seen["train"].reducer_flag), bool(seen["train"].grad_reducer_d.mean), int(seen["train"].grad_reducer_d.degree)
    np.savez(os.path.join(work, f"rank{rank}.npz"), composition=np.array(json.dumps(comp)), **out)


def dp_flow_case(name, world=2, **mode):
    import re
    import shutil
    import tempfile
    import torch.multiprocessing as mp
    from mindrec_amd import criteo
    work = tempfile.mkdtemp(prefix="ref_dp_")
    B, F, V, steps, n_eval = 1000, 39, 3000, 2, 2
    rng = np.random.default_rng(777)
    ids, wts, label = _batches(rng, world * steps + n_eval, B, F, V)
    ids[:, :, :13] = np.arange(13, dtype=np.int32)
    if mode.get("dynamic_embedding"):
        ids = ids * 7 + 100                                               # keys of a hash table: any integers
    flat = lambda a, lo, hi: a[lo:hi].reshape((hi - lo) * B, -1)          # noqa: E731
    nt = world * steps
    criteo.write_tfrecords(os.path.join(work, "data"), "train", flat(ids, 0, nt), flat(wts, 0, nt), flat(label, 0, nt), records_per_file=3)
    criteo.write_tfrecords(os.path.join(work, "data"), "test", flat(ids, nt, nt + n_eval), flat(wts, nt, nt + n_eval), flat(label, nt, nt + n_eval))
    over = dict(batch_size=B, field_size=F, emb_dim=8, vocab_size=V, deep_layer_dim=[32, 16, 16, 8], epochs=1, sparse=False, use_mixed_precision=False,
                dynamic_embedding=False, parameter_server=0, vocab_cache_size=0, dropout_flag=False, dataset_type="tfrecord", device_target="CPU",
                data_path=os.path.join(work, "data"), ckpt_path=os.path.join(work, "ckpt"))
    over.update(mode)
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cwd = os.getcwd()
    try:
        mp.spawn(_dp_worker, args=(world, port, work, over), nprocs=world, join=True)
    finally:
        os.chdir(cwd)
    out, logs = {}, {}
    for r in range(world):
        with np.load(os.path.join(work, f"rank{r}.npz")) as z:
            out.update({k: z[k] for k in z.files if k != "composition"})
            comp = json.loads(str(z["composition"]))
        logs[f"loss_log{r}"] = open(os.path.join(work, f"loss{r}.log")).read().strip().splitlines()
    for k in [k for k in out if k.startswith("rank0/init/") or k.startswith("rank0/final/") or k.startswith("rank0/seed/")]:
        k1 = k.replace("rank0/", "rank1/")
        if k.endswith("::keys"):
            continue
        if k.endswith("::values"):
            # hash tables: the replicas' key sets differ by what each rank's EVALUATION shard looked up (MapTensorGet inserts);
            # every key both hold -- all trained keys among them -- has the same row
            ka, kb = out[k.replace("::values", "::keys")], out[k1.replace("::values", "::keys")]
            common, ia, ib = np.intersect1d(ka, kb, return_indices=True)
            assert len(common) and np.array_equal(out[k][ia], out[k1][ib]), k
            continue
        assert np.array_equal(out[k], out[k1]), k                         # every replica starts from the same parameters and ends on the same
    for k in [k for k in list(out) if k.startswith("rank1/init/") or k.startswith("rank1/final/") or k.startswith("rank1/seed/")]:
        del out[k]
    ckpts = sorted(os.listdir(os.path.join(work, "ckpt", "ckpt_0"))) if os.path.isdir(os.path.join(work, "ckpt", "ckpt_0")) else []
    shutil.rmtree(work, ignore_errors=True)
    out.update(world=np.int64(world), steps=np.int64(steps), composition=np.array(json.dumps(comp)),
               logs=np.array(json.dumps(logs)), ckpts=np.array(json.dumps([c for c in ckpts if c.endswith(".ckpt")])),
               cfg=np.array(json.dumps({k: over[k] for k in ("batch_size", "field_size", "emb_dim", "vocab_size", "deep_layer_dim", "epochs", "sparse",
                                                             "use_mixed_precision", "dynamic_embedding", "vocab_cache_size", "parameter_server")})))
    _save(name, out)
    return {"composition": comp, **logs}


def hash_lookup_case(name):
    """HashEmbeddingLookup.construct alone (embedding.py:184-206): sparse True / False, int32 / int64 keys, max_norm."""
    from mindspore_rec import HashEmbeddingLookup
    out = {}
    rng = np.random.default_rng(5)
    variants = []
    for i, (kd, sparse, max_norm, D) in enumerate([(mindspore.int32, True, None, 8), (mindspore.int64, True, None, 16),
                                                   (mindspore.int32, False, None, 8), (mindspore.int64, True, 0.02, 8)]):
        mindspore.set_seed(300 + i)
        layer = HashEmbeddingLookup(embedding_size=D, key_dtype=kd, sparse=sparse, max_norm=max_norm)
        npd = np.int32 if kd == mindspore.int32 else np.int64
        raw = rng.integers(0, 60, size=(2, 11, 4))
        keys = (raw * 9 + 1000).astype(npd) if npd == np.int32 else (raw.astype(np.int64) * (2**33 + 5) - 2**40)
        ys = [_np(layer(Tensor(keys[c]))) for c in range(2)]
        k, v = layer.embedding_table.get_data()
        order = np.argsort(_np(k))
        out[f"v{i}/keys"], out[f"v{i}/out0"], out[f"v{i}/out1"] = keys, ys[0], ys[1]
        out[f"v{i}/table_keys"], out[f"v{i}/table_values"] = _np(k)[order], _np(v)[order]
        variants.append({"key_dtype": str(kd), "sparse": sparse, "max_norm": max_norm, "D": D, "seed": int(layer.embedding_table.seed)})
    out["variants"] = np.array(json.dumps(variants))
    _save(name, out)


def online_learning_ci():
    """ci/st/online_learning/test_online_learning.py:54-114 through the reference's RecModel (it asks for device_target "GPU":
    tensors are kept on the host by the test hook, the argument checks under test never touch a tensor)."""
    import importlib.util
    context._host_tensors = True
    try:
        spec = importlib.util.spec_from_file_location("ref_ci_online_learning", os.path.join(REF, "ci/st/online_learning/test_online_learning.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        ran = []
        for n in sorted(dir(m)):
            if n.startswith("test_"):
                getattr(m, n)()
                ran.append(n)
    finally:
        context._host_tensors = False
        context.set_context(device_target="CPU")
    return ran


def _save(name, arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KB, {len(arrays)} arrays")


if __name__ == "__main__":
    report = {"reference_ci_cases_passed": online_learning_ci()}
    report["ref_wd_dense"] = wide_deep_case("ref_wd_dense")                                               # the default: sparse False
    report["ref_wd_sparse"] = wide_deep_case("ref_wd_sparse", sparse=True, parameter_server=1)             # LazyAdam + FTRL on RowTensors
    report["ref_wd_dynamic"] = wide_deep_case("ref_wd_dynamic", sparse=True, dynamic_embedding=True)       # HashEmbeddingLookup x2
    report["ref_wd_mixed"] = wide_deep_case("ref_wd_mixed", mixed=True, sparse=True, parameter_server=1)   # fp16 DenseLayers
    report["ref_dcn"] = deep_cross_case("ref_dcn")
    report["ref_deepfm"] = deepfm_case("ref_deepfm")                                                       # fp32 DenseLayers
    report["ref_deepfm_mixed"] = deepfm_case("ref_deepfm_mixed", convert_dtype=True)                       # the default: fp16 DenseLayers
    hash_lookup_case("ref_hash_lookup")
    report["ref_train_eval_flow"] = train_eval_flow_case("ref_train_eval_flow")                            # train_and_eval.py's own flow
    report["ref_wd_dp2"] = dp_flow_case("ref_wd_dp2")                                                      # train_and_eval_distribute.py, 2 ranks, gloo
    report["ref_wd_dp2_dynamic"] = dp_flow_case("ref_wd_dp2_dynamic", dynamic_embedding=True, sparse=True)  # ... over hash tables: LazyAdam + FTRL on gathered row gradients
    with open(os.path.join(HERE, "ref_composition.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))
