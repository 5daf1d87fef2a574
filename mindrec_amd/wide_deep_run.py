"""Configuring and watching a Wide&Deep run the reference's way (host-side plumbing, no kernels):

  get_config            models/wide_deep/src/model_utils/config.py:42-127 -- a yaml file of up to three documents (the
                        configuration, help texts, allowed choices); every scalar key of the first becomes a `--key` option
                        typed by its default (booleans as Python literals), the command line overrides the file, the result is
                        an attribute namespace
  AUCMetric             models/wide_deep/src/metrics.py:23-52 -- collects (predict, label) pairs, roc_auc_score at eval()
  LossCallBack          models/wide_deep/src/callbacks.py:31-77 -- prints "===loss=== rank epoch step wide_loss deep_loss" every
                        step, appends to config.loss_file_name every per_print_times steps
  EvalCallBack          :79-131 -- at the end of every epoch runs model.eval(eval_dataset) and appends the metric values
                        and the evaluation time to config.eval_file_name
  WideDeepRunner        the train / eval networks of the scripts as two callables over a WideDeepEngine: train(ids, wts, label) ->
                        (wide_loss, deep_loss) as TrainStepWrap.construct returns them (wide_and_deep.py:472-492), eval(dataset)
                        -> {"auc": ...} as Model.eval with metrics={"auc": AUCMetric()} (train_and_eval.py:84-93)
"""
import argparse
import ast
import time

import numpy as np

from .mindspore_rec.train.callback import Callback


# ---- yaml + command line ---------------------------------------------------------------------------------------------
class Config:
    """Attribute namespace over a (nested) dictionary."""

    def __init__(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                v = Config(v)
            elif isinstance(v, (list, tuple)):
                v = [Config(x) if isinstance(x, dict) else x for x in v]
            setattr(self, k, v)

    def __repr__(self):
        return f"Config({self.__dict__!r})"


def parse_yaml(path):
    """(config, help texts, choices) of a configuration file holding one, two or three yaml documents."""
    import yaml
    with open(path, "r", encoding="utf-8") as f:
        try:
            docs = list(yaml.safe_load_all(f))
        except yaml.YAMLError as e:
            raise ValueError("Failed to parse yaml") from e
    if not 1 <= len(docs) <= 3:
        raise ValueError("At most 3 docs (config, description for help, choices) are supported in config yaml")
    docs += [{}] * (3 - len(docs))
    return docs[0] or {}, docs[1] or {}, docs[2] or {}


def get_config(config_path, argv=None):
    """The file's configuration with the command line laid over it.  argv: list of arguments (default: sys.argv[1:]);
    `--config_path FILE` on the command line replaces `config_path`."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config_path", type=str, default=config_path, help="Config file path")
    known, _ = pre.parse_known_args(argv)
    cfg, helper, choices = parse_yaml(known.config_path)
    ap = argparse.ArgumentParser(description="Wide&Deep configuration", parents=[pre])
    for key, default in cfg.items():
        if isinstance(default, (list, dict)):
            continue                        # structured values come from the file only
        kw = dict(default=default, choices=choices.get(key), help=helper.get(key, f"Please reference to {known.config_path}"))
        if isinstance(default, bool):
            ap.add_argument("--" + key, type=ast.literal_eval, **kw)
        elif default is None:
            ap.add_argument("--" + key, **kw)
        else:
            ap.add_argument("--" + key, type=type(default), **kw)
    args = ap.parse_args(argv)
    merged = dict(cfg)
    merged.update(vars(args))
    return Config(merged)


def engine_config(config, **overrides):
    """WideDeepConfig from the reference's configuration keys (models/wide_deep/default_config.yaml:14-44)."""
    from .wide_deep import WideDeepConfig
    get = lambda k, d: getattr(config, k, d)      # noqa: E731
    kw = dict(vocab_size=get("vocab_size", 200000), emb_dim=get("emb_dim", 80), field_size=get("field_size", 39),
              batch_size=get("batch_size", 16000), deep_layer_dim=list(get("deep_layer_dim", [1024, 512, 256, 128])),
              sparse=bool(get("sparse", False)), l2_coef=get("l2_coef", 8e-5), dropout_flag=bool(get("dropout_flag", False)),
              dynamic_embedding=bool(get("dynamic_embedding", False)), host_cache_rows=int(get("vocab_cache_size", 0) or 0),
              mlp_dtype="fp16" if bool(get("use_mixed_precision", True)) else "fp32")
    kw.update(overrides)
    return WideDeepConfig(**kw)


# ---- metric and callbacks ----------------------------------------------------------------------------------------------
def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    elif hasattr(x, "asnumpy"):
        x = x.asnumpy()
    return np.asarray(x)


class AUCMetric:
    """Area under the ROC curve over everything update() has seen since clear()."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.true_labels, self.pred_probs = [], []

    def update(self, *inputs):
        """inputs = (logits, predict, label), as PredictWithSigmoid returns them (wide_and_deep.py:495-518)."""
        self.pred_probs.extend(_np(inputs[1]).ravel().tolist())
        self.true_labels.extend(_np(inputs[2]).ravel().tolist())

    def eval(self):
        from sklearn.metrics import roc_auc_score
        if len(self.true_labels) != len(self.pred_probs):
            raise RuntimeError("true_labels.size is not equal to pred_probs.size()")
        auc = roc_auc_score(self.true_labels, self.pred_probs)
        print("====" * 20 + " auc_metric  end")
        print("====" * 20 + " auc: {}".format(auc))
        return auc


class LossCallBack(Callback):
    def __init__(self, config=None, per_print_times=1, rank_id=0):
        if not isinstance(per_print_times, int) or per_print_times < 0:
            raise ValueError("per_print_times must be in and >= 0.")
        self._per_print_times, self.config, self.rank_id = per_print_times, config, rank_id

    def step_end(self, run_context):
        p = run_context.original_args()
        out = p.get("net_outputs")
        if out is None:
            return
        wide_loss, deep_loss = (out if isinstance(out, (tuple, list)) else (out, out))[:2]
        # (as the reference prints them: the numpy scalars of `net_outputs[i].asnumpy()`, i.e. float32's shortest repr -- "0.68963903",
        # not the widened "0.6896390318870544")
        wide_loss, deep_loss = _np(wide_loss).reshape(()), _np(deep_loss).reshape(())
        step_in_epoch = (p.cur_step_num - 1) % p.batch_num + 1
        print("===loss===", self.rank_id, p.cur_epoch_num, step_in_epoch, wide_loss, deep_loss, flush=True)
        if self._per_print_times != 0 and p.cur_step_num % self._per_print_times == 0 and self.config is not None:
            line = "epoch: %s, step: %s, wide_loss: %s, deep_loss: %s" % (p.cur_epoch_num, step_in_epoch, wide_loss, deep_loss)
            with open(self.config.loss_file_name, "a+", encoding="utf-8") as f:
                f.write(line + "\n")
            print(line)


class EvalCallBack(Callback):
    def __init__(self, model, eval_dataset, auc_metric, config, print_per_step=1, rank_id=0):
        if not isinstance(print_per_step, int) or print_per_step < 0:
            raise ValueError("print_per_step must be int and >= 0.")
        self.print_per_step, self.model, self.eval_dataset, self.aucMetric = print_per_step, model, eval_dataset, auc_metric
        self.aucMetric.clear()
        self.eval_file_name, self.config, self.rank_id = config.eval_file_name, config, rank_id
        self.eval_values = []

    def epoch_end(self, run_context):
        self.aucMetric.clear()
        t0 = time.time()
        out = self.model.eval(self.eval_dataset)
        eval_time = int(time.time() - t0)
        stamp = time.strftime("%Y-%m-%d %H:%M%S", time.localtime())
        line = "{} == Rank: {} == EvalCallBack model.eval(): {}; eval_time: {}s".format(stamp, self.rank_id, out.values(), eval_time)
        print(line)
        self.eval_values = out.values()
        with open(self.eval_file_name, "a+", encoding="utf-8") as f:
            f.write(line + "\n")


class WideDeepRunner:
    """train(*batch) / eval(dataset) over a WideDeepEngine, with the reference's return conventions (module docstring)."""

    def __init__(self, engine, metrics=None):
        self.engine = engine
        self.metrics = dict(metrics or {})

    def __call__(self, ids, wts, label):
        loss = self.engine.train_step(ids, wts, label)
        return loss, self.engine.deep_loss(loss)

    def eval(self, dataset, dataset_sink_mode=False):
        for m in self.metrics.values():
            m.clear()
        for ids, wts, label in dataset:
            logit, prob = self.engine.predict(ids, wts)
            for m in self.metrics.values():
                m.update(logit, prob, label)
        # an epoch's end (EvalCallBack.epoch_end runs this): steps that dropped positions or overflowed the cache are not valid steps
        if getattr(self.engine, "_sharded", False):
            self.engine.check_shard_overflow()
        self.engine.check_cache()
        return {k: m.eval() for k, m in self.metrics.items()}

    def close(self):
        self.engine.close()
