"""Callback protocol subset that RecModel.online_train touches (rec_model.py:160-249;
SURVEY.md Appendix A.9): Callback, RunContext, _InternalCallbackParam, _CallbackManager."""


class _InternalCallbackParam(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


class RunContext:
    def __init__(self, original_args):
        if not isinstance(original_args, dict):
            raise TypeError(f"The argument 'original_args' of RunContext should be dict type, but got {type(original_args)}.")
        self._original_args = original_args
        self._stop_requested = False

    def original_args(self):
        return self._original_args

    def request_stop(self):
        self._stop_requested = True

    def get_stop_requested(self):
        return self._stop_requested


class Callback:
    def on_train_begin(self, run_context): self.begin(run_context)
    def on_train_epoch_begin(self, run_context): self.epoch_begin(run_context)
    def on_train_step_begin(self, run_context): self.step_begin(run_context)
    def on_train_step_end(self, run_context): self.step_end(run_context)
    def on_train_epoch_end(self, run_context): self.epoch_end(run_context)
    def on_train_end(self, run_context): self.end(run_context)
    # MindSpore's older method names, still dispatched to (models/wide_deep/src/callbacks.py:51,105)
    def begin(self, run_context): pass
    def epoch_begin(self, run_context): pass
    def step_begin(self, run_context): pass
    def step_end(self, run_context): pass
    def epoch_end(self, run_context): pass
    def end(self, run_context): pass
    def __enter__(self): return self
    def __exit__(self, *err): pass


class _CallbackManager(Callback):
    def __init__(self, callbacks):
        if callbacks is None:
            callbacks = []
        elif isinstance(callbacks, Callback):
            callbacks = [callbacks]
        for cb in callbacks:
            if not isinstance(cb, Callback):
                raise TypeError("When the 'callbacks' is a list, the elements in 'callbacks' must be Callback functions.")
        self._callbacks = list(callbacks)

    def __enter__(self): return self
    def __exit__(self, *err): return False

    def on_train_begin(self, rc): [cb.on_train_begin(rc) for cb in self._callbacks]
    def on_train_epoch_begin(self, rc): [cb.on_train_epoch_begin(rc) for cb in self._callbacks]
    def on_train_step_begin(self, rc): [cb.on_train_step_begin(rc) for cb in self._callbacks]
    def on_train_step_end(self, rc): [cb.on_train_step_end(rc) for cb in self._callbacks]
    def on_train_epoch_end(self, rc): [cb.on_train_epoch_end(rc) for cb in self._callbacks]
    def on_train_end(self, rc): [cb.on_train_end(rc) for cb in self._callbacks]


class TimeMonitor(Callback):
    """Per-step wall time, as the reference derives its throughput numbers (benchmarks/README.md:56)."""

    def __init__(self, data_size=None):
        import time
        self._time = time
        self.data_size = data_size
        self.step_ms = []

    def step_begin(self, run_context):
        self._t0 = self._time.perf_counter()

    def step_end(self, run_context):
        self.step_ms.append((self._time.perf_counter() - self._t0) * 1e3)
