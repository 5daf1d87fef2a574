"""`mindspore_rec.RecModel` -- mirrors mindspore_rec/train/rec_model.py:34-309: same constructor,
`online_train(train_dataset, callbacks=None, dataset_sink_mode=True, sink_size=1)`, same argument
checks in the same order with the same messages, same callback sequence and counters.

The train network is any callable taking one batch (`net(*batch)`) that performs forward, backward
and the optimizer applies itself (loss_fn / optimizer None, as every W&D / DCN script of the
reference does with its TrainStepWrap cell).  A dataset is any iterable of tuples with
`get_dataset_size()` and (optionally) `reset()`.

One deliberate superset: the loops stop when a callback calls `run_context.request_stop()`
(MindSpore's own Model.train honours it; the reference's online loops run until killed).
"""
import logging
import sys

from ... import _validator as Validator
from ... import context, nn
from .callback import RunContext, _CallbackManager, _InternalCallbackParam

logger = logging.getLogger("mindspore_rec")


class RecModel:
    def __init__(self, network, loss_fn=None, optimizer=None, metrics=None, eval_network=None, eval_indexes=None,
                 amp_level="O0", boost_level="O0"):
        if loss_fn is not None or optimizer is not None:
            raise NotImplementedError("RecModel here drives self-contained train networks (loss_fn=None, optimizer=None), "
                                      "the only form the reference's scripts use")
        if amp_level not in ("O0", "O2", "O3", "auto"):
            raise ValueError(f"For 'Model', the 'amp_level' must be one of ['O0', 'O2', 'O3', 'auto'], but got {amp_level}.")
        self._network = network
        self._train_network = network
        self._eval_network = eval_network
        self._metrics = metrics
        self._eval_indexes = eval_indexes
        self._parallel_mode = "stand_alone"
        self._device_number = 1
        self._parameter_broadcast = False
        self._loss_scale_manager = None
        self._datasets_bound = set()

    # ---- pieces of mindspore.Model that online_train calls ----------------------------------------
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
        if hasattr(network, "train") and getattr(network, "training", is_train) != is_train:
            network.train(is_train)
        return network

    def online_train(self, train_dataset, callbacks=None, dataset_sink_mode=True, sink_size=1):
        Validator.check_bool(dataset_sink_mode)                                         # rec_model.py:152
        if isinstance(self._train_network, nn.GraphCell) and dataset_sink_mode:         # :153-156
            raise ValueError("Dataset sink mode is currently not supported when training with a GraphCell.")
        if callbacks:
            self._check_methods_for_custom_callbacks(callbacks, "train")
        cb_params = _InternalCallbackParam()
        cb_params.train_network = self._train_network
        if dataset_sink_mode:
            cb_params.batch_num = sink_size                                             # :166-169
        else:
            cb_params.batch_num = train_dataset.get_dataset_size()                      # :170-171
        with _CallbackManager(callbacks) as list_callback:
            self._check_reuse_dataset(train_dataset)
            if not dataset_sink_mode:
                self._online_train_dataset_not_sink(train_dataset, list_callback, cb_params)
            elif context.get_context("device_target") == "CPU":                         # :179-186
                logger.info("The CPU doesn't support dataset sink mode currently,"
                            "so the training process will be performed with dataset not sink.")
                self._online_train_dataset_not_sink(train_dataset, list_callback, cb_params)
            else:
                self._online_train_dataset_sink(train_dataset, list_callback, cb_params, sink_size)

    def _online_train_dataset_not_sink(self, train_dataset, callbacks=None, cb_params=None):
        """Feed mode (rec_model.py:192-249): batches go straight to the network; the dataset is reset each epoch."""
        self._stream_epochs(train_dataset, callbacks, cb_params, sink=False, step_inc=1)

    def _online_train_dataset_sink(self, train_dataset, callbacks=None, cb_params=None, sink_size=1):
        """Sink mode (rec_model.py:251-309): sink_size must be a positive int and, for now, exactly 1."""
        sink_size = Validator.check_positive_int(sink_size)                             # :267
        if sink_size != 1:                                                              # :268-271
            raise ValueError(f"The sink_size parameter only support value of 1 currently, but got: {sink_size}")
        train_dataset.__model_hash__ = hash(self)            # sink mode binds the dataset to this model
        self._stream_epochs(train_dataset, callbacks, cb_params, sink=True, step_inc=sink_size)

    def _stream_epochs(self, dataset, cbs, p, sink, step_inc):
        """The unbounded epoch/step loop both modes share.  Callback order and counter updates follow the
        reference: train_begin; per epoch: epoch_begin, per batch (cur_step_num += step_inc; step_begin; run;
        net_outputs; step_end), [feed mode: dataset.reset()], epoch_end; train_end."""
        p.cur_epoch_num = 0 if not sink else p.get("cur_epoch_num", 0)
        p.cur_step_num = 0
        p.dataset_sink_mode = sink
        ctx = RunContext(p)
        cbs.on_train_begin(ctx)
        epoch = 0
        while epoch < sys.maxsize and not ctx.get_stop_requested():
            epoch += 1
            p.cur_epoch_num = epoch
            cbs.on_train_epoch_begin(ctx)
            if sink:
                p.train_network = self._train_network
            for batch in dataset:
                p.cur_step_num += step_inc
                cbs.on_train_step_begin(ctx)
                net = self._check_network_mode(self._train_network, True)
                p.net_outputs = net(*batch)
                cbs.on_train_step_end(ctx)
                if ctx.get_stop_requested():
                    break
            if not sink and hasattr(dataset, "reset"):
                dataset.reset()
            cbs.on_train_epoch_end(ctx)
        cbs.on_train_end(ctx)
