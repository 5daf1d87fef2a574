"""`mindspore.experimental.MapParameter` (mindspore_rec/ops/embedding.py:27,136-146; README.md:160-205).

The table itself lives in the kernel set's store -- on an MI355X `mindrec_amd.experimental.MapParameter`: a device
key -> row-number index over dense row storage in HBM (csrc/mrec_hash.hip), optimizer slots as further row tables."""
import sys

import torch

from .. import context as _context
from .._kernels import K
from ..common.tensor import Tensor, as_tensor

MAX_SIZE = sys.maxsize


class MapParameter:
    """MapParameter(key_dtype=int32, value_dtype=float32, value_shape=1, key_tensor=None, value_tensor=None,
    default_value='normal', permit_filter_value=1, evict_filter_value=MAX_SIZE, name=None, requires_grad=True).
    MI355X extra: `capacity` rows are reserved in HBM up front (keyword only)."""

    def __init__(self, key_dtype=torch.int32, value_dtype=torch.float32, value_shape=1, key_tensor=None, value_tensor=None,
                 default_value="normal", permit_filter_value=1, evict_filter_value=MAX_SIZE, name=None, requires_grad=True,
                 *, capacity=None, seed=None):
        if isinstance(value_shape, int):
            value_shape = (value_shape,)
        self.key_dtype, self.value_dtype, self.value_shape = key_dtype, value_dtype, tuple(int(s) for s in value_shape)
        self.default_value = default_value
        self.permit_filter_value, self.evict_filter_value = permit_filter_value, evict_filter_value
        self.name = name if name is not None else "Parameter"
        self.requires_grad = bool(requires_grad)
        self.unique = False
        self.cache_enable = False
        self.key = None
        self.device = _context._torch_device()
        self._row_hook, self._row_grads = None, []
        kw = {} if capacity is None else {"capacity": int(capacity)}
        if seed is None:
            from ..common import initializer as _init
            seed = _init._next_seed()        # default rows are a function of (seed, key, column): consecutive tables get consecutive seeds
        kw["seed"] = self.seed = int(seed)
        self._store = K().MapStore(key_dtype=key_dtype, value_dtype=value_dtype, value_shape=self.value_shape,
                                   default_value=default_value, permit_filter_value=permit_filter_value,
                                   evict_filter_value=evict_filter_value, name=self.name, device=self.device, **kw)
        if key_tensor is not None:
            self.put(key_tensor, value_tensor)

    # (autograd edge + trainable flag, as on Parameter)
    @property
    def trainable(self):
        return self.requires_grad

    def row_hook(self):
        if self._row_hook is None:
            self._row_hook = torch.zeros((), device=self.device, requires_grad=True)
        return self._row_hook

    @property
    def shape(self):
        return (len(self),) + self.value_shape

    @property
    def dtype(self):
        return self.value_dtype

    def _k(self, keys):
        if not isinstance(keys, torch.Tensor):
            keys = Tensor(keys, self.key_dtype)
        if keys.dtype != self.key_dtype:
            raise TypeError(f"For 'MapParameter', the key dtype must be {self.key_dtype}, but got {keys.dtype}.")
        return keys.as_subclass(torch.Tensor).to(self.device).reshape(-1)

    def get(self, key_tensor, insert_default_value=True):
        keys = self._k(key_tensor)
        out = self._store.get(keys, bool(insert_default_value))
        return as_tensor(out.reshape(tuple(key_tensor.shape if isinstance(key_tensor, torch.Tensor) else keys.shape) + self.value_shape))

    def put(self, key_tensor, value_tensor):
        keys = self._k(key_tensor)
        vals = value_tensor if isinstance(value_tensor, torch.Tensor) else Tensor(value_tensor, torch.float32)
        self._store.put(keys, vals.as_subclass(torch.Tensor).to(self.device, torch.float32).reshape((keys.numel(),) + self.value_shape))
        return self

    def erase(self, key_tensor):
        self._store.erase(self._k(key_tensor))
        return self

    def __getitem__(self, key_tensor):
        return self.get(key_tensor, True)

    def __setitem__(self, key_tensor, value_tensor):
        self.put(key_tensor, value_tensor)

    def __len__(self):
        return self._store.size()

    def get_keys(self):
        return as_tensor(self._store.export()[0])

    def get_values(self):
        return as_tensor(self._store.export()[1])

    def get_data(self):
        k, v = self._store.export()
        return as_tensor(k), as_tensor(v)

    def export_data(self, incremental=False):
        return tuple(as_tensor(t) for t in self._store.export_data(bool(incremental)))

    def import_data(self, data):
        self._store.import_data(tuple(t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t for t in data))

    def evict(self):
        return self._store.evict()

    def clone(self, init="zeros"):
        return MapParameter(self.key_dtype, self.value_dtype, self.value_shape, default_value=init,
                            permit_filter_value=self.permit_filter_value, evict_filter_value=self.evict_filter_value, name=self.name,
                            requires_grad=self.requires_grad)

    def __repr__(self):
        return (f"MapParameter (name={self.name}, key_dtype={self.key_dtype}, value_shape={self.value_shape}, "
                f"size={len(self)})")
