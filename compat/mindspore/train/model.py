"""`mindspore.Model`: train / eval / predict loops and the private pieces `mindspore_rec.RecModel` (a subclass) calls
(mindspore_rec/train/rec_model.py:34-309).

GRAPH_MODE on an MI355X "compiles" a recognised train network: `mindrec_amd.lowering.lower_train_step` turns the
reference's Wide&Deep / Deep&Cross TrainStepWrap into the fused engine (one HIP graph per step, parameters shared); every
other network runs its `construct` primitive by primitive."""
import numpy as np
import torch

from .. import context
from ..common.tensor import Tensor
from ..nn.cell import Cell
from .callback import RunContext, _CallbackManager, _InternalCallbackParam


class DatasetHelper:
    """Iterates a dataset as tuples of device Tensors.  In sink mode one pass yields `sink_size` batches' worth of steps
    (sink_size == -1: the whole epoch)."""

    def __init__(self, dataset, dataset_sink_mode=True, sink_size=-1, epoch_num=1):
        self.dataset, self.sink, self.sink_size = dataset, bool(dataset_sink_mode), sink_size
        self._it = None

    def __iter__(self):
        dev = context._torch_device()
        if self.sink and self.sink_size and self.sink_size > 0:
            if self._it is None:
                self._it = iter(self.dataset)
            for _ in range(self.sink_size):
                try:
                    row = next(self._it)
                except StopIteration:
                    self._it = iter(self.dataset)
                    try:
                        row = next(self._it)
                    except StopIteration:
                        return
                yield tuple(_dev(x, dev) for x in row)
            return
        for row in self.dataset:
            yield tuple(_dev(x, dev) for x in row)

    def sink_size_(self):
        return self.sink_size


def _dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.as_subclass(Tensor) if x.device == dev else x.to(dev).as_subclass(Tensor)
    return Tensor(np.asarray(x), device=dev)


class Model:
    def __init__(self, network, loss_fn=None, optimizer=None, metrics=None, eval_network=None, eval_indexes=None,
                 amp_level="O0", boost_level="O0", **kwargs):
        if not isinstance(network, Cell):
            raise TypeError(f"For 'Model', the 'network' must be a Cell, but got {type(network).__name__}.")
        if amp_level not in ("O0", "O1", "O2", "O3", "auto"):
            raise ValueError(f"For 'Model', the 'amp_level' must be one of ['O0', 'O1', 'O2', 'O3', 'auto'], but got {amp_level}.")
        if amp_level not in ("O0",):
            raise NotImplementedError("automatic mixed precision levels are not provided: the in-scope models cast inside DenseLayer")
        self._network, self._loss_fn, self._optimizer = network, loss_fn, optimizer
        self._metrics, self._eval_indexes = metrics, eval_indexes
        self._amp_level, self._boost_level = amp_level, boost_level
        self._loss_scale_manager = kwargs.get("loss_scale_manager")
        self._parallel_mode = context.get_auto_parallel_context("parallel_mode")
        self._device_number = context.get_auto_parallel_context("device_num")
        self._parameter_broadcast = context.get_auto_parallel_context("parameter_broadcast")
        self._train_network = self._build_train_network()
        self._eval_network = eval_network
        if eval_network is None and metrics and loss_fn is not None:
            from ..nn.wrap.cell_wrapper import WithEvalCell
            self._eval_network = WithEvalCell(network, loss_fn)
            self._eval_indexes = [0, 1, 2]
        if eval_indexes is not None and (not isinstance(eval_indexes, list) or len(eval_indexes) != 3):
            raise ValueError("For 'Model', 'eval_indexes' must be a list of three ints or None.")
        self._metric_fns = self._get_metrics(metrics)
        self._predict_network = network
        self._dataset_owner = {}

    # ---- construction ------------------------------------------------------------------------------------------
    def _build_train_network(self):
        net = self._network
        if self._optimizer is not None:
            from ..nn.wrap.cell_wrapper import TrainOneStepCell, WithLossCell
            if self._loss_fn is not None:
                net = WithLossCell(net, self._loss_fn)
            net = TrainOneStepCell(net, self._optimizer)
        elif self._loss_fn is not None:
            from ..nn.wrap.cell_wrapper import WithLossCell
            net = WithLossCell(net, self._loss_fn)
        return net

    @staticmethod
    def _get_metrics(metrics):
        from ..nn.metrics import Metric, get_metric_fn
        if metrics is None:
            return {}
        if isinstance(metrics, dict):
            for k, v in metrics.items():
                if not isinstance(v, Metric):
                    raise TypeError(f"For 'Model', the value of metrics[{k!r}] must be a Metric, but got {type(v).__name__}.")
            return dict(metrics)
        if isinstance(metrics, (set, list, tuple)):
            return {n: get_metric_fn(n) for n in metrics}
        raise TypeError(f"For 'Model', the 'metrics' must be dict, set, list or None, but got {type(metrics).__name__}.")

    # ---- pieces RecModel.online_train calls (rec_model.py:152-190, 205-207, 288-296) -----------------------------
    @staticmethod
    def _check_methods_for_custom_callbacks(callbacks, current_mode):
        return None

    def _check_reuse_dataset(self, dataset):
        owner = getattr(dataset, "__model_hash__", None)
        if owner is not None and owner != hash(self):
            raise RuntimeError("The dataset object had been used in other model by model.train(...), "
                               "please create a new dataset.")

    @staticmethod
    def _check_network_mode(network, is_train):
        if network.training != is_train:
            network.set_train(is_train)
        return network

    def close(self):
        """Releases what a lowered train step holds on the device (captured steps of a sharded engine hold RCCL kernels: call this
        before destroy_process_group()).  The cells keep their parameters; a later call lowers again."""
        for net in (self._train_network, self._network):
            low = getattr(net, "__dict__", {}).get("_lowered")
            if low:
                eng = low.engine
                getattr(eng, "close", getattr(eng, "release_graphs", lambda: None))()

    # ---- a row-sharded lowered train step: the cell's tables are mirrors of the ranks' shards (mindrec_amd/lowering.py) ----------
    def sync_parameters(self):
        """A collective (every rank calls it): the train cell's full-size tables / MapParameters are refreshed from all ranks'
        shards.  Free when nothing was trained since the last call, and when the step is not a sharded lowering."""
        low = getattr(self._train_network, "__dict__", {}).get("_lowered")
        if low and getattr(low, "sharded", False):
            low.sync()

    @staticmethod
    def _sync_plan(callbacks):
        """More than one rank: every rank learns the periods at which SOME rank's ModelCheckpoint will read the parameters (the
        reference checkpoints on rank 0 alone, train_and_eval_distribute.py:108-112), so that all ranks refresh the mirrors
        together right before -- a checkpoint is a one-rank read, the refresh a collective.  -> [[period, last saved step], ...]"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        from .callback import ModelCheckpoint
        cbs = callbacks if isinstance(callbacks, (list, tuple)) else ([callbacks] if callbacks is not None else [])
        mine = []
        for cb in cbs:
            if isinstance(cb, ModelCheckpoint):
                mine.append((int(cb._config.save_checkpoint_steps or 1), int(cb._last_saved_step)))       # (seconds-based: every step)
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, mine)
        return [list(x) for x in sorted({tuple(x) for r in every for x in r})]

    @staticmethod
    def _sync_if_due(net, p, plan, force=False):
        low = getattr(net, "__dict__", {}).get("_lowered")
        if not (plan is not None and low and getattr(low, "sharded", False)):
            return
        due = force
        for st in plan:
            if p.cur_step_num >= st[1] + st[0]:
                st[1], due = p.cur_step_num, True
        if due:
            low.sync()

    @staticmethod
    def _lowered(network, batch=None):
        """GRAPH_MODE's compile step (mindspore/_lower.py): the fused engine behind a recognised train cell, or None."""
        from .._lower import lowered
        return lowered(network, batch)

    def _exec_preprocess(self, is_train, dataset, dataset_sink_mode, sink_size=-1, epoch_num=1, dataset_helper=None):
        if dataset_sink_mode and not is_train:
            dataset.__loop_size__ = 1
        if dataset_helper is None:
            dataset_helper = DatasetHelper(dataset, dataset_sink_mode, sink_size, epoch_num)
        network = self._train_network if is_train else self._eval_network
        if dataset_sink_mode:
            dataset.__model_hash__ = hash(self)
        return dataset_helper, network

    # ---- train -------------------------------------------------------------------------------------------------
    def train(self, epoch, train_dataset, callbacks=None, dataset_sink_mode=False, sink_size=-1, initial_epoch=0):
        from .. import _checkparam as V
        V.check_bool(dataset_sink_mode)
        if isinstance(epoch, bool) or not isinstance(epoch, int) or epoch <= 0:
            raise ValueError(f"For 'Model.train', the 'epoch' must be int and must > 0, but got {epoch!r}.")
        if sink_size != -1:
            V.check_positive_int(sink_size, "sink_size")
        if context.get_context("device_target") == "CPU":
            dataset_sink_mode = False
        ds_size = train_dataset.get_dataset_size()
        if ds_size == 0:
            raise ValueError("There is no valid data in dataset, please check dataset file firstly.")
        p = _InternalCallbackParam()
        p.train_network, p.epoch_num = self._train_network, epoch
        p.batch_num = (sink_size if dataset_sink_mode and sink_size > 0 else ds_size)
        p.mode, p.loss_fn, p.optimizer, p.parallel_mode, p.device_number = "train", self._loss_fn, self._optimizer, self._parallel_mode, self._device_number
        p.train_dataset, p.list_callback, p.train_dataset_element = train_dataset, None, None
        p.cur_epoch_num, p.cur_step_num, p.dataset_sink_mode = initial_epoch, 0, dataset_sink_mode
        sync_plan = self._sync_plan(callbacks)
        with _CallbackManager(callbacks) as cbs:
            self._check_reuse_dataset(train_dataset)
            rc = RunContext(p)
            cbs.on_train_begin(rc)
            helper, net = self._exec_preprocess(True, train_dataset, dataset_sink_mode, -1, epoch)
            for e in range(initial_epoch, epoch):
                p.cur_epoch_num = e + 1
                cbs.on_train_epoch_begin(rc)
                if dataset_sink_mode:
                    # the device loop: the steps of one sink run back to back, callbacks see the sink's last outputs
                    net = self._check_network_mode(net, True)
                    steps = sink_size if sink_size > 0 else ds_size
                    cbs.on_train_step_begin(rc)
                    self._run_sink(net, helper, steps, p)
                    p.cur_step_num += steps
                    self._sync_if_due(net, p, sync_plan)
                    cbs.on_train_step_end(rc)
                else:
                    for batch in helper:
                        p.cur_step_num += 1
                        p.train_dataset_element = batch
                        cbs.on_train_step_begin(rc)
                        net = self._check_network_mode(net, True)
                        p.net_outputs = self._run_step(net, batch)
                        self._sync_if_due(net, p, sync_plan)
                        cbs.on_train_step_end(rc)
                        if rc.get_stop_requested():
                            break
                    if hasattr(train_dataset, "reset"):
                        train_dataset.reset()
                cbs.on_train_epoch_end(rc)
                if rc.get_stop_requested():
                    break
            self._sync_if_due(net, p, sync_plan, force=True)          # (ModelCheckpoint saves once more at the end)
            cbs.on_train_end(rc)

    def _run_step(self, net, batch):
        low = self._lowered(net, batch)
        return low(*batch) if low is not None else net(*batch)

    def _run_sink(self, net, helper, steps, p):
        it = iter(DatasetHelper(helper.dataset, True, steps))
        first = next(it, None)
        if first is None:
            return
        low = self._lowered(net, first)
        import itertools
        it = itertools.chain([first], it)
        if low is not None and hasattr(low, "run_sink"):
            p.net_outputs = low.run_sink(list(it))
            return
        for batch in it:
            p.net_outputs = low(*batch) if low is not None else net(*batch)

    # ---- eval / predict ----------------------------------------------------------------------------------------
    def eval(self, valid_dataset, callbacks=None, dataset_sink_mode=False):
        if not self._metric_fns:
            raise ValueError("For Model.eval, the model argument 'metrics' can not be None or empty, you should set the argument "
                             "'metrics' for model.")
        if self._eval_network is None:
            raise ValueError("For Model.eval, the 'eval_network' (or 'loss_fn' to build one) must be given.")
        net = self._eval_network
        net.set_train(False)
        for m in self._metric_fns.values():
            m.clear()
        dev = context._torch_device()
        with torch.no_grad():
            for batch in valid_dataset:
                outs = net(*[_dev(x, dev) for x in batch])
                outs = outs if isinstance(outs, (tuple, list)) else (outs,)
                for m in self._metric_fns.values():
                    idx = getattr(m, "indexes", None)
                    m.update(*([outs[i] for i in idx] if idx else outs))
        if hasattr(valid_dataset, "reset"):
            valid_dataset.reset()
        return {k: m.eval() for k, m in self._metric_fns.items()}

    def predict(self, *predict_data):
        self._predict_network.set_train(False)
        dev = context._torch_device()
        with torch.no_grad():
            return self._predict_network(*[_dev(x, dev) for x in predict_data])

    @property
    def train_network(self):
        return self._train_network

    @property
    def predict_network(self):
        return self._predict_network

    @property
    def eval_network(self):
        return self._eval_network
