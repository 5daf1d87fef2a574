"""`mindspore.common.initializer`: `initializer(init, shape, dtype)` and the few Initializer classes the in-scope models
name (models/wide_deep/src/wide_and_deep.py:36-71).  'normal' = N(0, 0.01) [EXT, SURVEY A.3].  MindSpore's generator
cannot be restated; the stream here is this repo's counter-based one (mindrec_amd/csrc/mrec_rng.h): value = f(seed, row,
column), the same on the HIP kernel set and in the CPU oracle, seeded from `set_seed` and a per-call counter."""
import numbers

import numpy as np
import torch

from .. import context as _context
from .._kernels import K
from .tensor import Tensor

_state = {"seed": 0, "calls": 0}


def _set_global_seed(seed):
    _state["seed"], _state["calls"] = int(seed), 0


def _next_seed():
    _state["calls"] += 1
    return (_state["seed"] * 1_000_003 + _state["calls"]) & 0x7FFFFFFF


class Initializer:
    def __init__(self, **kw):
        self._kw = kw
        self.shape, self.dtype, self.seed = None, torch.float32, None

    def _bind(self, shape, dtype):
        import copy
        o = copy.copy(self)
        o.shape, o.dtype, o.seed = tuple(int(s) for s in shape), dtype, _next_seed()
        return o

    def _materialize(self, shape, dtype, device):
        return self._fill(tuple(shape), device).to(dtype)

    def to_tensor(self):
        return Tensor(self._materialize(self.shape, self.dtype, _context._torch_device()))

    init_data = to_tensor


class Zero(Initializer):
    def _fill(self, shape, device):
        return torch.zeros(shape, dtype=torch.float32, device=device)


class One(Initializer):
    def _fill(self, shape, device):
        return torch.ones(shape, dtype=torch.float32, device=device)


class Constant(Initializer):
    def __init__(self, value):
        super().__init__(value=value)
        self.value = float(value)

    def _fill(self, shape, device):
        return torch.full(shape, self.value, dtype=torch.float32, device=device)


class Normal(Initializer):
    """Normal(sigma=0.01, mean=0.0)."""

    def __init__(self, sigma=0.01, mean=0.0):
        super().__init__(sigma=sigma, mean=mean)
        self.sigma, self.mean = float(sigma), float(mean)

    def _fill(self, shape, device):
        rows = shape[0] if len(shape) > 1 else 1
        cols = int(np.prod(shape[1:])) if len(shape) > 1 else int(np.prod(shape)) if shape else 1
        t = torch.empty((rows, cols), dtype=torch.float32, device=device)
        K().fill_normal_(t, self.seed if self.seed is not None else _next_seed(), self.sigma)
        t = t.reshape(shape)
        return t + self.mean if self.mean else t


class Uniform(Initializer):
    """Uniform(scale=0.07): U(-scale, scale).  Host-generated (numpy, seeded) -- off the hot path: the in-scope configs
    initialise with 'normal' (default_config.yaml:39-41)."""

    def __init__(self, scale=0.07):
        super().__init__(scale=scale)
        self.scale = float(scale)

    def _fill(self, shape, device):
        rng = np.random.default_rng(self.seed if self.seed is not None else _next_seed())
        return torch.from_numpy(rng.uniform(-self.scale, self.scale, shape).astype(np.float32)).to(device)


_ALIASES = {"zeros": Zero, "zero": Zero, "ones": One, "one": One, "normal": Normal, "uniform": Uniform}


def initializer(init, shape=None, dtype=torch.float32):
    """-> an Initializer bound to (shape, dtype); `Parameter(initializer(...))` materialises it."""
    if shape is None:
        shape = ()
    if isinstance(shape, numbers.Integral):
        shape = (shape,)
    if isinstance(init, torch.Tensor):
        if tuple(init.shape) != tuple(shape):
            raise ValueError(f"For 'initializer', the shape of the 'init' {tuple(init.shape)} must equal 'shape' {tuple(shape)}.")
        c = Constant(0.0)._bind(shape, dtype)
        c._fill = lambda s, device, _t=init: _t.detach().to(device, torch.float32).clone()
        return c
    if isinstance(init, str):
        if init.lower() not in _ALIASES:
            raise ValueError(f"For 'initializer', the class corresponding to '{init}' was not found.")
        init = _ALIASES[init.lower()]()
    elif isinstance(init, numbers.Number):
        init = Constant(init)
    elif not isinstance(init, Initializer):
        raise TypeError(f"For 'initializer', the type of the 'init' argument should be 'Tensor', 'number', 'string' "
                        f"or 'initializer', but got {type(init)}.")
    return init._bind(shape, dtype)
