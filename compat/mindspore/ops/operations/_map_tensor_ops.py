"""`mindspore.ops.operations._map_tensor_ops`: MapTensorGet / Put / Erase on a MapParameter
(mindspore_rec/ops/embedding.py:149,193,199; README.md:160-205).  The table is the kernel set's device hash table
(csrc/mrec_hash.hip)."""
import torch

from ...common.tensor import as_tensor
from .. import _grad
from ..primitive import Primitive


def _check(mp, keys, who):
    from ...experimental import MapParameter
    if not isinstance(mp, MapParameter):
        raise TypeError(f"For '{who}', the first input must be a MapParameter, but got {type(mp).__name__}.")
    if keys.dtype != mp.key_dtype:
        raise TypeError(f"For '{who}', the key dtype must be {mp.key_dtype}, but got {keys.dtype}.")


class MapTensorGet(Primitive):
    """MapTensorGet(insert_default_value=True)(map, key_tensor) -> values [n, *value_shape]; a missing key is inserted
    with a row drawn from `default_value`."""

    def __init__(self, insert_default_value=True):
        super().__init__()
        self.insert_default_value = bool(insert_default_value)

    def __call__(self, map_parameter, key_tensor):
        _check(map_parameter, key_tensor, "MapTensorGet")
        mp = map_parameter
        if torch.is_grad_enabled() and mp.requires_grad:
            out = _grad.MapGet.apply(mp.row_hook(), mp, key_tensor, self.insert_default_value)
        else:
            out = mp._store.get(key_tensor.as_subclass(torch.Tensor).reshape(-1), self.insert_default_value)
        return as_tensor(out.reshape(tuple(key_tensor.shape) + mp.value_shape))


class MapTensorPut(Primitive):
    def __call__(self, map_parameter, key_tensor, value_tensor):
        _check(map_parameter, key_tensor, "MapTensorPut")
        map_parameter.put(key_tensor, value_tensor)
        return map_parameter


class MapTensorErase(Primitive):
    def __call__(self, map_parameter, key_tensor):
        _check(map_parameter, key_tensor, "MapTensorErase")
        map_parameter.erase(key_tensor)
        return map_parameter
