"""BUILD-CONTAINER-ONLY: the reference's own model code (unmodified, imported from /root/reference; see make_ref_fixtures.py for what
it runs on) at BASELINE's CONFIGURATION sizes -- where the engines dispatch to their large-shape kernels (three-part GEMMs, slab
counts, the fused tail, graph capture) and the small fixtures cannot see a composition error:

  ref_dcn_cfg2.npz   models/deep_and_cross/src/deep_and_cross.py:206-354 at configs[2]: batch 16384, 39 fields x 30, DenseLayers
                     1170-1024-1024, 6 cross layers, vocabulary 200 000, fp32, Adam(lr 1e-4, loss_scale 1000); 2 steps
  ref_wd_cfg1.npz    models/wide_deep/src/wide_and_deep.py:136-492 at the benchmarked shape: batch 16384, 39 fields, dim 80, the
                     1024-512-256-128 net in fp16 (use_mixed_precision), sparse=True (LazyAdam + FTRL), vocabulary 200 000; 2 steps

Inputs and initial parameters are NOT stored (100+ MB): both sides derive them from tests/_ref_fixtures.py (cfgsize_batches,
cfgsize_param -- the oracle's counter-based normal stream keyed by the parameter's name); the generator puts them into the
reference's model through Parameter.set_data.  Stored: per-step losses, ~1300 table rows, sum / sum of squares / 32 elements of every
dense parameter, 256 evaluation logits.  Each file < 1 MB.  Usage: python tests/golden/make_ref_fixtures_cfgsize.py"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_fixtures as G  # noqa: E402  (sets up compat/mindspore + the CPU kernel set + the reference on sys.path)
import _ref_fixtures as RF  # noqa: E402
from mindspore import Tensor  # noqa: E402


def _set_params(struct, sigma_of=lambda k: 0.01):
    for k, p in struct.items():
        p.set_data(Tensor(RF.cfgsize_param(k, tuple(p.shape), sigma_of(k))))


def dcn_cfg2(name="ref_dcn_cfg2", S=2):
    dcn = G._ref_module("deep_and_cross", "deep_and_cross")
    cfg = types.SimpleNamespace(batch_size=16384, field_size=39, emb_dim=30, vocab_size=200000, deep_layer_dim=[1024, 1024], cross_layer_num=6,
                                keep_prob=1.0)
    G.mindspore.set_seed(1000)
    net = dcn.DeepCrossModel(cfg)
    train = dcn.TrainStepWrap(dcn.NetWithLossClass(net))
    evaln = dcn.PredictWithSigmoid(net)
    train.set_train()
    struct = dict(net.parameters_and_names())
    _set_params(struct, lambda k: 0.05 if "embedding_table" in k else 0.01)
    ids, wts, label = RF.cfgsize_batches(4202, S, cfg.batch_size, cfg.field_size, cfg.vocab_size)
    losses = [float(G._np(train(Tensor(ids[s]), Tensor(wts[s]), Tensor(label[s])))) for s in range(S)]
    evaln.set_train(False)
    logits, _, _ = evaln(Tensor(ids[S - 1]), Tensor(wts[S - 1]), Tensor(label[S - 1]))
    out = {"loss": np.array(losses, np.float64), "eval_logits": G._np(logits).reshape(-1)[:256]}
    t, f = RF.cfgsize_rows(ids, cfg.vocab_size)
    table = G._np(struct["deep_embeddinglookup.embedding_table"])
    out["rows_touched"], out["rows_free"], out["table_touched"], out["table_free"] = t, f, table[t], table[f]
    for k, p in struct.items():
        out["sum/" + k] = RF.cfgsize_summary(G._np(p))
    comp = {"optimizer": type(train.optimizer).__name__, "lr": train.optimizer.get_lr(), "eps": train.optimizer.eps,
            "loss_scale": train.optimizer.loss_scale, "sens": float(train.sens), "weights": list(struct), "steps": S, "batch_seed": 4202,
            "shapes": {k: [int(x) for x in p.shape] for k, p in struct.items()}, "sigma": {k: (0.05 if "embedding_table" in k else 0.01) for k in struct}}
    out.update(cfg=np.array(json.dumps(vars(cfg))), composition=np.array(json.dumps(comp)))
    G._save(name, out)
    return comp


def wd_cfg1(name="ref_wd_cfg1", S=2):
    wd = G._ref_module("wide_deep", "wide_and_deep")
    cfg = types.SimpleNamespace(batch_size=16384, field_size=39, emb_dim=80, vocab_size=200000, vocab_cache_size=0,
                                deep_layer_dim=[1024, 512, 256, 128], deep_layer_act="relu", keep_prob=1.0, dropout_flag=False,
                                use_mixed_precision=True, parameter_server=1, sparse=True, dynamic_embedding=False,
                                weight_bias_init=["normal", "normal"], emb_init="normal", init_args=[-0.01, 0.01], l2_coef=8e-5,
                                full_batch=False, field_slice=False)
    G.mindspore.set_seed(1000)
    net = wd.WideDeepModel(cfg)
    loss_net = wd.NetWithLossClass(net, cfg)
    train = wd.TrainStepWrap(loss_net, parameter_server=True, sparse=True, dynamic_embedding=False)
    evaln = wd.PredictWithSigmoid(net)
    train.set_train()
    struct = dict(net.parameters_and_names())
    _set_params(struct)
    comp = G._wd_composition(train, loss_net, struct)
    ids, wts, label = RF.cfgsize_batches(4101, S, cfg.batch_size, cfg.field_size, cfg.vocab_size)
    lw, ld = [], []
    for s in range(S):
        a, b = train(Tensor(ids[s]), Tensor(wts[s]), Tensor(label[s]))
        lw.append(float(G._np(a)))
        ld.append(float(G._np(b)))
    evaln.set_train(False)
    logits, _, _ = evaln(Tensor(ids[S - 1]), Tensor(wts[S - 1]), Tensor(label[S - 1]))
    out = {"loss_w": np.array(lw, np.float64), "loss_d": np.array(ld, np.float64), "eval_logits": G._np(logits).reshape(-1)[:256]}
    t, f = RF.cfgsize_rows(ids, cfg.vocab_size)
    out["rows_touched"], out["rows_free"] = t, f
    for key, short in (("embedding_table", "deep"), ("wide_embeddinglookup.embedding_table", "wide")):
        tab = G._np(struct[key])
        out[short + "_touched"], out[short + "_free"] = tab[t], tab[f]
    for k, p in struct.items():
        if "embedding_table" not in k:
            out["sum/" + k] = RF.cfgsize_summary(G._np(p))
    comp.update(steps=S, batch_seed=4101, shapes={k: [int(x) for x in p.shape] for k, p in struct.items()}, sigma={k: 0.01 for k in struct})
    out.update(cfg=np.array(json.dumps({k: v for k, v in vars(cfg).items()})), composition=np.array(json.dumps(comp)))
    G._save(name, out)
    return comp


if __name__ == "__main__":
    rep = {"ref_dcn_cfg2": dcn_cfg2(), "ref_wd_cfg1": wd_cfg1()}
    with open(os.path.join(HERE, "ref_composition_cfgsize.json"), "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    print(json.dumps(rep, indent=1, sort_keys=True))
