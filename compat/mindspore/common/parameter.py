"""`mindspore.Parameter` / `ParameterTuple`.  A Parameter is a leaf Tensor that requires grad and has a name;
a table looked up through SparseGatherV2 / EmbeddingLookup(sparse=True) additionally collects RowTensor gradients
 instead of a dense [V, D] one."""
import itertools

import torch

from .tensor import Tensor

_key = itertools.count(1)


def _get_unique_parameter_key():
    """A process-unique integer (mindspore_rec/ops/embedding.py:168)."""
    return next(_key)


class Parameter(Tensor):
    """Parameter(default_input, name=None, requires_grad=True, layerwise_parallel=False, parallel_optimizer=True)."""

    @staticmethod
    def __new__(cls, default_input, name=None, requires_grad=True, layerwise_parallel=False, parallel_optimizer=True):
        from .initializer import Initializer
        if isinstance(default_input, Initializer):
            data = default_input.to_tensor().detach().as_subclass(torch.Tensor)
        elif isinstance(default_input, torch.Tensor):
            data = default_input.detach().as_subclass(torch.Tensor)
        else:
            data = Tensor(default_input).as_subclass(torch.Tensor)
        p = torch.Tensor._make_subclass(cls, data, bool(requires_grad) and data.is_floating_point())
        return p

    def __init__(self, default_input, name=None, requires_grad=True, layerwise_parallel=False, parallel_optimizer=True):
        self.name = name if name is not None else "Parameter"
        self._trainable = bool(requires_grad)
        self._row_hook = None        # 0-dim leaf that gives autograd an edge into a sparse lookup's backward
        self._row_grads = []         # (ids, per-position gradients) filed by that backward
        self.key = None
        self.cache_enable = False
        self.sliced = False
        self.is_init = True

    @property
    def name(self):                      # (torch.Tensor.name is a read-only slot of the named-tensor API: shadow it)
        return self.__dict__.get("_ms_name", "Parameter")

    @name.setter
    def name(self, value):
        self.__dict__["_ms_name"] = value

    @property
    def trainable(self):
        return self._trainable

    def row_hook(self):
        if self._row_hook is None or self._row_hook.device != self.device:
            self._row_hook = torch.zeros((), device=self.device, requires_grad=True)
        return self._row_hook

    def set_data(self, data, slice_shape=False):
        with torch.no_grad():
            self.as_subclass(torch.Tensor).copy_(torch.as_tensor(data, device=self.device).reshape(self.shape))
        return self

    def init_data(self, layout=None, set_sliced=False):
        return self

    def clone(self, init="same"):
        from .initializer import initializer
        if init == "same":
            q = Parameter(self.detach().clone(), name=self.name, requires_grad=self._trainable)
        else:
            q = Parameter(initializer(init, self.shape, self.dtype), name=self.name, requires_grad=self._trainable)
        return q

    def __repr__(self):
        return f"Parameter (name={self.name}, shape={tuple(self.shape)}, dtype={self.dtype}, requires_grad={self._trainable})"

    def __deepcopy__(self, memo):
        return self.clone()

    def __reduce_ex__(self, proto):
        return (_rebuild_param, (self.asnumpy(), self.name, self._trainable))


def _rebuild_param(arr, name, rg):
    return Parameter(Tensor(arr, device="cpu"), name=name, requires_grad=rg)


class ParameterTuple(tuple):
    """A tuple of Parameters (and MapParameters) -- what TrainStepWrap hands its optimizers
    (models/wide_deep/src/wide_and_deep.py:412-413)."""

    def __new__(cls, iterable):
        from ..experimental import MapParameter
        data = tuple(iterable)
        names = set()
        for p in data:
            if not isinstance(p, (Parameter, MapParameter)):
                raise TypeError(f"ParameterTuple input should be 'Parameter' collection, but got a {type(p)}.")
            if p.name in names:
                raise ValueError(f"The value {p.name!r} (parameter name) already exists in the ParameterTuple.")
            names.add(p.name)
        return tuple.__new__(cls, data)

    def clone(self, prefix, init="same"):
        out = []
        for p in self:
            q = p.clone(init)
            q.name = prefix + "." + q.name
            out.append(q)
        return ParameterTuple(out)
