"""`mindspore_rec.RecModel` over this repo's `mindspore.Model` -- the interface of mindspore_rec/train/rec_model.py:34-309:
`online_train(train_dataset, callbacks=None, dataset_sink_mode=True, sink_size=1)` with the reference's argument checks in
the reference's order (the three messages its CI pins, ci/st/online_learning/test_online_learning.py:72,93,114), the same
callback sequence and counters.  Both of the reference's loops (:192-249 feed, :251-309 sink) are one loop here; a
recognised train step runs as its lowered engine.  One superset: `run_context.request_stop()` ends the stream (the
reference runs until killed)."""
import sys

from mindspore import Model
from mindspore import _checkparam as Validator
from mindspore import context
from mindspore import log as logger
from mindspore import nn
from mindspore.parallel._utils import _device_number_check
from mindspore.train.callback import RunContext, _CallbackManager, _InternalCallbackParam


class RecModel(Model):
    def online_train(self, train_dataset, callbacks=None, dataset_sink_mode=True, sink_size=1):
        Validator.check_bool(dataset_sink_mode)
        if isinstance(self._train_network, nn.GraphCell) and dataset_sink_mode:
            raise ValueError("Dataset sink mode is currently not supported when training with a GraphCell.")
        _device_number_check(self._parallel_mode, self._device_number)
        if callbacks:
            self._check_methods_for_custom_callbacks(callbacks, "train")
        if self._parameter_broadcast:
            self._train_network.set_broadcast_flag()
        p = _InternalCallbackParam()
        p.train_network = self._train_network
        p.batch_num = sink_size if dataset_sink_mode else train_dataset.get_dataset_size()
        with _CallbackManager(callbacks) as cbs:
            self._check_reuse_dataset(train_dataset)
            sink = bool(dataset_sink_mode)
            if sink and context.get_context("device_target") == "CPU":
                logger.info("The CPU doesn't support dataset sink mode currently,"
                            "so the training process will be performed with dataset not sink.")
                sink = False
            if sink:
                sink_size = Validator.check_positive_int(sink_size)
                if sink_size != 1:
                    raise ValueError(f"The sink_size parameter only support value of 1 currently, but got: {sink_size}")
            self._stream(train_dataset, cbs, p, sink, sink_size if sink else 1)

    def _stream(self, dataset, cbs, p, sink, step_inc):
        p.cur_epoch_num, p.cur_step_num, p.dataset_sink_mode = 0, 0, sink
        rc = RunContext(p)
        sync_plan = self._sync_plan(list(getattr(cbs, "_callbacks", [])))
        cbs.on_train_begin(rc)
        # sink mode: an "epoch" of the callback protocol is one sink of `step_inc` batches (the reference re-enters its dataset
        # helper once per epoch, rec_model.py:281-303 -- checkpoint names `<prefix>-<epoch>_<step>` count sinks); feed mode: one
        # pass over the dataset, then reset()
        helper, net = self._exec_preprocess(is_train=True, dataset=dataset, dataset_sink_mode=sink, sink_size=step_inc if sink else -1,
                                            epoch_num=-1)
        for epoch in range(1, sys.maxsize):
            p.cur_epoch_num = epoch
            cbs.on_train_epoch_begin(rc)
            p.train_network = net
            for batch in helper:
                p.cur_step_num += 1
                cbs.on_train_step_begin(rc)
                net = self._check_network_mode(net, True)
                p.net_outputs = self._run_step(net, batch)
                self._sync_if_due(net, p, sync_plan)
                cbs.on_train_step_end(rc)
                if rc.get_stop_requested():
                    break
            if not sink:
                dataset.reset()
            cbs.on_train_epoch_end(rc)
            if rc.get_stop_requested():
                break
        self._sync_if_due(net, p, sync_plan, force=True)
        cbs.on_train_end(rc)
