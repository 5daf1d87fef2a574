"""TEST INFRASTRUCTURE: replaying tests/golden/ref_*.npz -- recorded from the REFERENCE's own Python by
tests/golden/make_ref_fixtures.py -- through the engines.  Shared by the CPU tests (the engines' host logic over the oracle op
set: does the product's composition equal the reference's?) and the GPU tests (the same over libmrec_hip.so)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, json.loads(str(z["cfg"])), json.loads(str(z["composition"]))


def row_rel(a, b):
    den = np.maximum(np.abs(b).max(axis=-1), 1e-30)
    return float((np.abs(a.astype(np.float64) - b).max(axis=-1) / den).max())


def wd_config(cfg, comp, **over):
    from mindrec_amd.wide_deep import WideDeepConfig
    kw = dict(vocab_size=cfg["vocab_size"], emb_dim=cfg["emb_dim"], field_size=cfg["field_size"], batch_size=cfg["batch_size"],
              deep_layer_dim=list(cfg["deep_layer_dim"]), sens=comp["sens"], adam_lr=comp["lr_d"], adam_eps=comp["eps_d"],
              ftrl_lr=comp["lr_w"], ftrl_l1=comp["l1_w"], ftrl_l2=comp["l2_w"], ftrl_initial_accum=comp["initial_accum_w"],
              mlp_dtype="fp16" if cfg["use_mixed_precision"] else "fp32", sparse=bool(cfg["sparse"]), l2_coef=comp["l2_coef"],
              dynamic_embedding=bool(cfg["dynamic_embedding"]),
              wide_b_optimizer="ftrl" if "wide_b" in comp["weights_w"] else "adam")
    kw.update(over)
    return WideDeepConfig(**kw)


def wd_load_init(eng, z, dynamic=False):
    n = len(eng.dims) - 1
    eng.load_dense_parameters([z[f"init/dense_layer_{i + 1}.weight"] for i in range(n)],
                              [z[f"init/dense_layer_{i + 1}.bias"] for i in range(n)], extra=z["init/wide_b"])
    if not dynamic:
        with torch.no_grad():
            eng.deep.copy_(torch.from_numpy(z["init/embedding_table"]))
            eng.wide.copy_(torch.from_numpy(z["init/wide_embeddinglookup.embedding_table"]))


def wd_replay(eng, z, dev):
    losses = []
    for s in range(z["ids"].shape[0]):
        ids, wts, label = (torch.from_numpy(z[k][s]).to(dev) for k in ("ids", "wts", "label"))
        losses.append(float(eng.train_step(ids, wts, label)))
    return np.array(losses)


def wd_dense_state(eng):
    n = len(eng.dims) - 1
    out = {}
    for i in range(n):
        out[f"dense_layer_{i + 1}.weight"] = eng.dense[2 * i].detach().cpu().numpy()
        out[f"dense_layer_{i + 1}.bias"] = eng.dense[2 * i + 1].detach().cpu().numpy()
    out["wide_b"] = eng.wide_b.detach().cpu().numpy()
    return out


# ---- Deep&Cross: engine.dense = [W1, b1, W2, b2, W3 [(h2 + X), 1], b3, cross_w [L, X], cross_b [L, X]] -------------------------
def dcn_load_init(eng, z):
    L = eng.cfg.cross_layer_num
    with torch.no_grad():
        eng.table.copy_(torch.from_numpy(z["init/deep_embeddinglookup.embedding_table"]))
        W1, b1, W2, b2, W3, b3, cw, cb = eng.dense
        for t, k in ((W1, "dense_layer_1.weight"), (b1, "dense_layer_1.bias"), (W2, "dense_layer_2.weight"), (b2, "dense_layer_2.bias"),
                     (W3, "dense_layer_3.weight"), (b3, "dense_layer_3.bias")):
            t.copy_(torch.from_numpy(z["init/" + k]).reshape(t.shape))
        for l in range(L):                           # CrossLayer.cross_weight / cross_bias are [X, 1] (deep_and_cross.py:126-127)
            cw[l].copy_(torch.from_numpy(z[f"init/cross_layer_{l + 1}.cross_weight"]).reshape(-1))
            cb[l].copy_(torch.from_numpy(z[f"init/cross_layer_{l + 1}.cross_bias"]).reshape(-1))


def dcn_state(eng):
    W1, b1, W2, b2, W3, b3, cw, cb = (t.detach().cpu().numpy() for t in eng.dense)
    out = {"deep_embeddinglookup.embedding_table": eng.table.detach().cpu().numpy(), "dense_layer_1.weight": W1, "dense_layer_1.bias": b1,
           "dense_layer_2.weight": W2, "dense_layer_2.bias": b2, "dense_layer_3.weight": W3, "dense_layer_3.bias": b3}
    for l in range(eng.cfg.cross_layer_num):
        out[f"cross_layer_{l + 1}.cross_weight"] = cw[l].reshape(-1, 1)
        out[f"cross_layer_{l + 1}.cross_bias"] = cb[l].reshape(-1, 1)
    return out


# ---- DeepFM: V_l2 [V, D], W_l2 [V, 1], DenseLayer x5 ---------------------------------------------------------------------------------
def deepfm_config(cfg, comp, **over):
    from mindrec_amd.deepfm import DeepFMConfig
    kw = dict(data_vocab_size=cfg["data_vocab_size"], data_emb_dim=cfg["data_emb_dim"], data_field_size=cfg["data_field_size"],
              batch_size=cfg["batch_size"], deep_layer_dims=list(cfg["deep_layer_args"][0]), l2_coef=comp["l2_coef"],
              learning_rate=comp["lr"], epsilon=comp["eps"], loss_scale=comp["loss_scale"],
              mlp_dtype="fp16" if comp["convert_dtype"] else "fp32", graphs="none")
    kw.update(over)
    return DeepFMConfig(**kw)


def deepfm_load_init(eng, z):
    n = len(eng.dims) - 1
    with torch.no_grad():
        eng.V_l2.copy_(torch.from_numpy(z["init/embedding_table"]))
        eng.W_l2.copy_(torch.from_numpy(z["init/fm_w"]))
    eng.load_dense_parameters([z[f"init/dense_layer_{i + 1}.weight"] for i in range(n)], [z[f"init/dense_layer_{i + 1}.bias"] for i in range(n)])


def deepfm_state(eng):
    n = len(eng.dims) - 1
    out = {"embedding_table": eng.V_l2.detach().cpu().numpy(), "fm_w": eng.W_l2.detach().cpu().numpy()}
    for i in range(n):
        out[f"dense_layer_{i + 1}.weight"] = eng.dense[2 * i].detach().cpu().numpy()
        out[f"dense_layer_{i + 1}.bias"] = eng.dense[2 * i + 1].detach().cpu().numpy()
    return out


# ---- models/wide_deep/train_and_eval.py's flow (ref_train_eval_flow.npz) through this repo's runner ---------------------------------
def run_train_eval_flow(eng, z, dev, work_dir):
    """test_train_eval(config) (train_and_eval.py:66-104) restated over the engine-level API: RecModel(WideDeepRunner(engine)) with
    LossCallBack + EvalCallBack + AUCMetric, `epochs` passes over the training batches.  Returns (loss.log lines, eval.log lines
    without their time stamps, AUC per epoch)."""
    import os
    import re
    from mindrec_amd.mindspore_rec.train.callback import Callback
    from mindrec_amd.mindspore_rec.train.rec_model import RecModel
    from mindrec_amd.wide_deep_run import AUCMetric, Config, EvalCallBack, LossCallBack, WideDeepRunner
    nt, ne, epochs = int(z["n_train_steps"]), int(z["n_eval_steps"]), int(z["epochs"])
    run_cfg = Config({"loss_file_name": os.path.join(work_dir, "loss.log"), "eval_file_name": os.path.join(work_dir, "eval.log"), "sparse": False})

    order = [int(v) for v in z["train_order"]] if "train_order" in z.files else list(range(nt)) * epochs

    class DS:
        """epoch e hands out the batches the reference's (shuffling) reader handed out in its epoch e"""

        def __init__(self, per_epoch):
            self.per_epoch, self.epoch = per_epoch, 0

        def get_dataset_size(self):
            return len(self.per_epoch[0])

        def __iter__(self):
            batches = self.per_epoch[min(self.epoch, len(self.per_epoch) - 1)]
            self.epoch += 1
            for s in batches:
                yield tuple(torch.from_numpy(z[k][s]).to(dev) for k in ("ids", "wts", "label"))

    class StopAfter(Callback):
        def epoch_end(self, run_context):
            if run_context.original_args().cur_epoch_num >= epochs:
                run_context.request_stop()

    metric = AUCMetric()
    net = WideDeepRunner(eng, metrics={"auc": metric})
    class Eval(DS):
        def __iter__(self):
            for s in self.per_epoch[0]:
                yield tuple(torch.from_numpy(z[k][s]).to(dev) for k in ("ids", "wts", "label"))

    ev = EvalCallBack(net, Eval([list(range(nt, nt + ne))]), metric, run_cfg)
    RecModel(net).online_train(DS([order[e * nt:(e + 1) * nt] for e in range(epochs)]), callbacks=[ev, LossCallBack(config=run_cfg), StopAfter()],
                               dataset_sink_mode=False)
    loss_lines = open(run_cfg.loss_file_name).read().strip().splitlines()
    eval_lines = [re.sub(r"eval_time: \d+s", "eval_time: Ns", re.sub(r"^.*?== Rank", "== Rank", ln))
                  for ln in open(run_cfg.eval_file_name).read().strip().splitlines()]
    aucs = [float(re.search(r"dict_values\(\[([0-9.eE+-]+)\]\)", ln).group(1)) for ln in eval_lines]
    return loss_lines, eval_lines, aucs


def check_train_eval_flow(z, got, loss_rtol=2e-6, auc_tol=3e-6):          # (AUC: a swapped pair of the 2000 eval samples moves it by ~1e-6)
    import json
    import re
    loss_lines, eval_lines, aucs = got
    ref_loss, ref_eval = json.loads(str(z["loss_log"])), json.loads(str(z["eval_log"]))
    assert len(loss_lines) == len(ref_loss) and len(eval_lines) == len(ref_eval)
    pat = re.compile(r"^epoch: (\d+), step: (\d+), wide_loss: ([0-9.eE+-]+), deep_loss: ([0-9.eE+-]+)$")
    for a, b in zip(loss_lines, ref_loss):
        ma, mb = pat.match(a), pat.match(b)
        assert ma and mb, (a, b)                                  # the reference's line format
        assert ma.group(1, 2) == mb.group(1, 2), (a, b)           # epoch / step-in-epoch numbering
        for i in (3, 4):
            assert abs(float(ma.group(i)) - float(mb.group(i))) <= loss_rtol * abs(float(mb.group(i))), (a, b)
            assert len(ma.group(i)) <= 12, a                      # printed as float32 (shortest repr), like the reference's numpy scalars
    strip = lambda ln: re.sub(r"dict_values\(\[[0-9.eE+-]+\]\)", "dict_values([AUC])", ln)         # noqa: E731
    assert [strip(x) for x in eval_lines] == [strip(x) for x in ref_eval]
    assert np.allclose(aucs, z["auc"], rtol=0, atol=auc_tol), (aucs, z["auc"])


def check_data_parallel_fixture(make_engine, dev):
    """ref_wd_dp2.npz (the reference's 2-process data-parallel run) against ONE engine fed the concatenation of the ranks' batches."""
    import json
    import re
    z, cfg, comp = load("ref_wd_dp2")
    assert comp["reducer_flag"] and comp["gradients_mean"] and comp["degree"] == 2 and comp["optimizer_d"] == "Adam"
    world, steps = int(z["world"]), int(z["steps"])
    eng = make_engine(wd_config(dict(cfg, batch_size=cfg["batch_size"] * world), comp))
    wd_load_init(eng, {k.replace("rank0/", ""): z[k] for k in z.files if k.startswith("rank0/init/")})
    logs = json.loads(str(z["logs"]))
    pat = re.compile(r"wide_loss: ([0-9.eE+-]+), deep_loss: ([0-9.eE+-]+)")
    for s in range(steps):
        ids, wts, label = (torch.from_numpy(np.concatenate([z[f"rank{r}/{k}"][s] for r in range(world)])).to(dev) for k in ("ids", "wts", "label"))
        loss = eng.train_step(ids, wts, label)
        lw, ld = float(loss), float(eng.deep_loss(loss))
        ref = [tuple(float(v) for v in pat.search(logs[f"loss_log{r}"][s]).groups()) for r in range(world)]
        assert abs(lw - np.mean([a for a, _ in ref])) <= 2e-6 * lw, (s, lw, ref)
        assert abs(ld - np.mean([b for _, b in ref])) <= 2e-6 * ld, (s, ld, ref)
    for k, v in wd_dense_state(eng).items():
        assert np.allclose(v, z["rank0/final/" + k], rtol=2e-4, atol=1e-7), k
    assert row_rel(eng.deep.cpu().numpy(), z["rank0/final/embedding_table"]) <= 1e-5
    assert np.allclose(eng.wide.cpu().numpy(), z["rank0/final/wide_embeddinglookup.embedding_table"], rtol=1e-4, atol=1e-8)
    assert json.loads(str(z["ckpts"])) == ["widedeep_train-1_2.ckpt"]            # rank 0 alone checkpoints (train_and_eval_distribute.py:108-110)


# ---- fixtures at CONFIGURATION size (tests/golden/make_ref_fixtures_cfgsize.py): inputs and initial parameters are too large to
# ---- commit, so both sides derive them from these functions; the fixture holds what the REFERENCE's code computed from them -------
def cfgsize_param(name, shape, sigma=0.01):
    """Initial value of the parameter `name`: the oracle's counter-based N(0, sigma^2) stream keyed by a CRC of the name."""
    import zlib
    from oracle import oracle as O
    shape = tuple(int(x) for x in shape)
    rows = shape[0]
    cols = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    return O.fill_normal(zlib.crc32(name.encode()) & 0x7FFFFFFF, rows, cols, sigma).reshape(shape)


def cfgsize_batches(seed, S, B, F, V):
    """Criteo-like batches: with 39 fields the first 13 are the constant ids 0..12 with fractional weights
    (datasets/criteo_1tb/process_data.py:138-147), the rest Zipf(1.05) ids with weight 1 (:149-162); labels Bernoulli(0.3)."""
    rng = np.random.default_rng(seed)
    nd = 13 if F == 39 else 0
    ids = (np.minimum(rng.zipf(1.05, size=(S, B, F)), V - nd - 1) + nd - 1 + 1).astype(np.int32)
    wts = np.ones((S, B, F), np.float32)
    if nd:
        ids[:, :, :nd] = np.arange(nd, dtype=np.int32)
        wts[:, :, :nd] = rng.random((S, B, nd)).astype(np.float32)
    label = (rng.random((S, B, 1)) < 0.3).astype(np.float32)
    return ids, wts, label


def cfgsize_summary(a, n=32):
    """What a fixture keeps of a large tensor: sum, sum of squares (float64) and n elements at fixed places."""
    a = np.asarray(a, np.float32).reshape(-1)
    idx = (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % a.size
    return np.concatenate([[a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum()], a[idx].astype(np.float64)])


def cfgsize_rows(ids, V, n=1024):
    """Row numbers a fixture keeps of a table: n of the rows the batches touched (the hottest included) and n / 4 they did not."""
    touched = np.unique(ids)
    rng = np.random.default_rng(99)
    t = np.unique(np.concatenate([touched[:16], rng.choice(touched, size=min(n, touched.size), replace=False)]))
    free = np.setdiff1d(rng.integers(0, V, size=n), touched)[: n // 4]
    return t.astype(np.int64), free.astype(np.int64)
