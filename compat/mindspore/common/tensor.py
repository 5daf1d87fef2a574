"""`mindspore.Tensor` as a torch.Tensor subclass: torch owns the device memory, nothing else.
Results of torch operators on Tensors stay Tensors (torch's default subclass propagation), so `.asnumpy()`
-- what the reference's callbacks call on the step's outputs (models/wide_deep/src/callbacks.py:58-60) -- is
always there."""
import numpy as np
import torch

from .. import context as _context


def _device():
    return _context._torch_device()


class Tensor(torch.Tensor):
    """Tensor(input_data=None, dtype=None, shape=None, init=None)."""

    @staticmethod
    def __new__(cls, input_data=None, dtype=None, shape=None, init=None, device=None):
        if init is not None:
            data = init._materialize(tuple(shape), dtype or torch.float32, device or _device())
        elif isinstance(input_data, torch.Tensor):
            data = input_data.detach()
            if dtype is not None and data.dtype != dtype:
                data = data.to(dtype)
            if device is not None:
                data = data.to(device)
        else:
            arr = np.asarray(input_data)
            if dtype is None and arr.dtype == np.float64 and not isinstance(input_data, np.ndarray):
                arr = arr.astype(np.float32)          # python floats become float32, as in MindSpore
            data = torch.from_numpy(np.ascontiguousarray(arr)).to(device or _device())
            if dtype is not None:
                data = data.to(dtype)
        return torch.Tensor._make_subclass(cls, data, False)

    def __init__(self, *a, **kw):
        pass

    def asnumpy(self):
        return self.detach().cpu().numpy()

    def set_dtype(self, dtype):
        return self.to(dtype)

    @property
    def size_(self):
        return self.numel()

    def __repr__(self):
        return "Tensor(" + torch.Tensor.__repr__(self.detach().as_subclass(torch.Tensor)) + ")"

    def __reduce_ex__(self, proto):
        return (_rebuild, (self.asnumpy(), type(self).__name__))


def _rebuild(arr, _cls):
    return Tensor(arr, device="cpu")


def as_tensor(x):
    """torch.Tensor (any subclass) -> Tensor view, without a copy."""
    if isinstance(x, Tensor):
        return x
    return x.as_subclass(Tensor)
