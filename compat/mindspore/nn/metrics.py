"""`mindspore.nn.metrics`: the Metric protocol (models/wide_deep/src/metrics.py:23-52 subclasses it) and Loss."""
import numpy as np


class Metric:
    def __init__(self):
        self._indexes = None

    def clear(self):
        raise NotImplementedError

    def update(self, *inputs):
        raise NotImplementedError

    def eval(self):
        raise NotImplementedError

    def set_indexes(self, indexes):
        if not isinstance(indexes, list) or not all(isinstance(i, int) and i >= 0 for i in indexes):
            raise ValueError(f"For 'set_indexes', the argument 'indexes' must be a list of non-negative ints, but got {indexes}.")
        self._indexes = indexes
        return self

    @property
    def indexes(self):
        return self._indexes

    @staticmethod
    def _convert_data(data):
        if hasattr(data, "asnumpy"):
            return data.asnumpy()
        if hasattr(data, "detach"):
            return data.detach().cpu().numpy()
        return np.asarray(data)


class Loss(Metric):
    def __init__(self):
        super().__init__()
        self.clear()

    def clear(self):
        self._sum, self._n = 0.0, 0

    def update(self, *inputs):
        self._sum += float(np.mean(self._convert_data(inputs[0])))
        self._n += 1

    def eval(self):
        if self._n == 0:
            raise RuntimeError("The 'Loss' metric has seen no data.")
        return self._sum / self._n


def get_metric_fn(name, *a, **k):
    if name == "loss":
        return Loss()
    raise KeyError(f"metric {name!r} is not provided; pass a Metric object")
