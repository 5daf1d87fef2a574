"""`mindspore.train.callback`: the Callback protocol (shared with mindrec_amd.mindspore_rec), TimeMonitor, LossMonitor,
and the periodic checkpoint callback the reference's scripts install
(models/wide_deep/train_and_eval.py:96-97; examples/online_learning/online_train.py:81-82)."""
import os
import time

from mindrec_amd.mindspore_rec.train.callback import (Callback, RunContext, _CallbackManager,  # noqa: F401
                                                      _InternalCallbackParam)


class TimeMonitor(Callback):
    """TimeMonitor(data_size=None): prints `Train epoch time: ... ms, per step time: ... ms` at the end of every epoch
    (benchmarks/README.md:56 derives throughput from that line); per-step times are kept in `step_ms`."""

    def __init__(self, data_size=None):
        self.data_size = data_size
        self.step_ms = []
        self._e0 = self._t0 = None
        self._steps_at_epoch = 0

    def epoch_begin(self, run_context):
        self._e0 = time.perf_counter()
        self._steps_at_epoch = run_context.original_args().get("cur_step_num", 0)

    def step_begin(self, run_context):
        self._t0 = time.perf_counter()

    def step_end(self, run_context):
        if self._t0 is not None:
            self.step_ms.append((time.perf_counter() - self._t0) * 1e3)

    def epoch_end(self, run_context):
        if self._e0 is None:
            return
        p = run_context.original_args()
        ms = (time.perf_counter() - self._e0) * 1e3
        steps = self.data_size or max(1, p.get("cur_step_num", 0) - self._steps_at_epoch) or p.get("batch_num", 1)
        print(f"Train epoch time: {ms:5.3f} ms, per step time: {ms / steps:5.3f} ms", flush=True)


class LossMonitor(Callback):
    def __init__(self, per_print_times=1):
        if not isinstance(per_print_times, int) or per_print_times < 0:
            raise ValueError(f"For 'LossMonitor', the argument 'per_print_times' must be int and >= 0, but got {per_print_times}")
        self._per = per_print_times
        self.losses = []

    def step_end(self, run_context):
        import numpy as np
        p = run_context.original_args()
        out = p.net_outputs
        if isinstance(out, (tuple, list)):
            out = out[0]
        loss = float(np.mean(out.asnumpy() if hasattr(out, "asnumpy") else out))
        self.losses.append(loss)
        if self._per and p.cur_step_num % self._per == 0:
            print(f"epoch: {p.cur_epoch_num} step: {p.cur_step_num}, loss is {loss}", flush=True)


class CheckpointConfig:
    """CheckpointConfig(save_checkpoint_steps=1, save_checkpoint_seconds=0, keep_checkpoint_max=5,
    keep_checkpoint_per_n_minutes=0, integrated_save=True, async_save=False, saved_network=None, append_info=None, ...)."""

    def __init__(self, save_checkpoint_steps=1, save_checkpoint_seconds=0, keep_checkpoint_max=5, keep_checkpoint_per_n_minutes=0,
                 integrated_save=True, async_save=False, saved_network=None, append_info=None, enc_key=None, enc_mode="AES-GCM",
                 exception_save=False, **kw):
        for name, v in (("save_checkpoint_steps", save_checkpoint_steps), ("save_checkpoint_seconds", save_checkpoint_seconds),
                        ("keep_checkpoint_max", keep_checkpoint_max), ("keep_checkpoint_per_n_minutes", keep_checkpoint_per_n_minutes)):
            if v is not None and (isinstance(v, bool) or not isinstance(v, int) or v < 0):
                raise ValueError(f"For 'CheckpointConfig', the '{name}' must be int and >= 0, but got {v!r}.")
        if not save_checkpoint_steps and not save_checkpoint_seconds and not keep_checkpoint_max and not keep_checkpoint_per_n_minutes:
            raise ValueError("For 'CheckpointConfig', the input arguments 'save_checkpoint_steps', 'save_checkpoint_seconds', "
                             "'keep_checkpoint_max' and 'keep_checkpoint_per_n_minutes' can't be all None or 0.")
        if enc_key is not None:
            raise NotImplementedError("checkpoint encryption is not provided")
        self.save_checkpoint_steps = save_checkpoint_steps or None
        self.save_checkpoint_seconds = save_checkpoint_seconds or None
        if self.save_checkpoint_steps:
            self.save_checkpoint_seconds = None
        self.keep_checkpoint_max = keep_checkpoint_max or None
        self.keep_checkpoint_per_n_minutes = None if self.keep_checkpoint_max else (keep_checkpoint_per_n_minutes or None)
        if not self.keep_checkpoint_max and not self.keep_checkpoint_per_n_minutes:
            self.keep_checkpoint_max = 1
        self.integrated_save, self.async_save = bool(integrated_save), bool(async_save)
        self.saved_network, self.append_dict = saved_network, dict(append_info[0]) if append_info and isinstance(append_info[0], dict) else None


class ModelCheckpoint(Callback):
    """ModelCheckpoint(prefix='CKP', directory=None, config=None): saves `cb_params.train_network` (or
    config.saved_network) as `<directory>/<prefix>-<epoch>_<step in epoch>.ckpt` every `save_checkpoint_steps` steps (or
    `save_checkpoint_seconds`), keeps the newest `keep_checkpoint_max`, and saves once more at train end."""

    def __init__(self, prefix="CKP", directory=None, config=None):
        if not isinstance(prefix, str) or "/" in prefix:
            raise ValueError(f"For 'ModelCheckpoint', the argument 'prefix' must be a string without '/', but got {prefix!r}.")
        self._prefix = prefix
        self._directory = os.path.realpath(directory) if directory else os.getcwd()
        if config is not None and not isinstance(config, CheckpointConfig):
            raise TypeError(f"For 'ModelCheckpoint', the type of argument 'config' should be 'CheckpointConfig', but got {type(config)}.")
        self._config = config or CheckpointConfig()
        self._files = []
        self._last_t = time.time()
        self._last_saved_step = 0
        self._latest = ""

    @property
    def latest_ckpt_file_name(self):
        return self._latest

    def step_end(self, run_context):
        self._maybe_save(run_context.original_args(), force=False)

    def end(self, run_context):
        self._maybe_save(run_context.original_args(), force=True)

    def _due(self, p):
        c = self._config
        if c.save_checkpoint_steps:
            return p.cur_step_num >= self._last_saved_step + c.save_checkpoint_steps
        if c.save_checkpoint_seconds:
            return time.time() - self._last_t >= c.save_checkpoint_seconds
        return False

    def _maybe_save(self, p, force):
        if p.cur_step_num == self._last_saved_step or not (force or self._due(p)):
            return
        from .serialization import save_checkpoint
        os.makedirs(self._directory, exist_ok=True)
        batch_num = p.get("batch_num") or 1
        step_in_epoch = (p.cur_step_num - 1) % batch_num + 1
        name = os.path.join(self._directory, f"{self._prefix}-{p.cur_epoch_num}_{step_in_epoch}.ckpt")
        net = self._config.saved_network or p.train_network
        append = dict(self._config.append_dict or {})
        append.update(epoch_num=p.cur_epoch_num, step_num=p.cur_step_num)
        save_checkpoint(net, name, self._config.integrated_save, self._config.async_save, append)
        self._last_saved_step, self._last_t, self._latest = p.cur_step_num, time.time(), name
        if name in self._files:
            self._files.remove(name)
        self._files.append(name)
        while self._config.keep_checkpoint_max and len(self._files) > self._config.keep_checkpoint_max:
            old = self._files.pop(0)
            if os.path.exists(old):
                os.remove(old)
