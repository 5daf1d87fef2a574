"""`mindspore.nn.layer.basic`: Dropout, ClipByNorm, Dense (wide_and_deep.py:22,98,117-118; embedding.py:29,202-205)."""
import itertools

import torch

from ..._kernels import K
from ...common.initializer import initializer
from ...common.parameter import Parameter
from ...common.tensor import as_tensor
from ...ops import operations as P
from ..cell import Cell

_layer_ids = itertools.count()


def _dropout(x, keep_prob, owner):
    """x * mask / keep_prob with the kernel set's counter-based mask: a function of (seed, call number, layer, row, column)
    (mindrec_amd/csrc/mrec_dropout.h; MindSpore's own generator cannot be restated)."""
    from ...common import initializer as _init
    if not hasattr(owner, "_drop_layer"):
        owner._drop_layer, owner._drop_calls = next(_layer_ids), 0
    owner._drop_calls += 1
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x.reshape(1, -1)
    mask = K().dropout_mask(x2.shape[0], x2.shape[1], float(keep_prob), _init._state["seed"], owner._drop_calls, owner._drop_layer, x.device)
    return as_tensor(x * (mask.reshape(x.shape).to(x.dtype) * (1.0 / float(keep_prob))))


class Dropout(Cell):
    """Dropout(keep_prob=0.5, p=None): training mode zeroes an element with probability p = 1 - keep_prob and scales the
    rest by 1 / keep_prob."""

    def __init__(self, keep_prob=0.5, p=None, dtype=torch.float32):
        super().__init__()
        if p is not None:
            if not 0.0 <= p < 1.0:
                raise ValueError(f"For 'Dropout', the 'p' must be a number in range [0.0, 1.0), but got {p}.")
            keep_prob = 1.0 - p
        elif not 0.0 < keep_prob <= 1.0:
            raise ValueError(f"For 'Dropout', the 'keep_prob' must be a number in range (0.0, 1.0], but got {keep_prob}.")
        self.keep_prob = float(keep_prob)
        self.p = 1.0 - self.keep_prob

    def construct(self, x):
        if not self.training or self.keep_prob >= 1.0:
            return x
        return _dropout(x, self.keep_prob, self)


class ClipByNorm(Cell):
    """ClipByNorm(axis=None)(x, clip_norm): x * clip_norm / max(||x||_2 over axis, clip_norm) [EXT]."""

    def __init__(self, axis=None):
        super().__init__()
        self.axis = () if axis is None else ((axis,) if isinstance(axis, int) else tuple(axis))

    def construct(self, x, clip_norm):
        cn = clip_norm if isinstance(clip_norm, torch.Tensor) else torch.as_tensor(clip_norm, dtype=x.dtype, device=x.device)
        dims = self.axis if self.axis else tuple(range(x.dim()))
        n = torch.sqrt((x * x).sum(dim=dims, keepdim=True))
        return as_tensor(x * cn / torch.maximum(n, cn.to(x.device)))


class Dense(Cell):
    """Dense(in_channels, out_channels, weight_init='normal', bias_init='zeros', has_bias=True, activation=None):
    weight [out, in], y = x . weight^T + bias."""

    def __init__(self, in_channels, out_channels, weight_init="normal", bias_init="zeros", has_bias=True, activation=None):
        super().__init__()
        self.weight = Parameter(initializer(weight_init, [out_channels, in_channels]), name="weight")
        self.bias = Parameter(initializer(bias_init, [out_channels]), name="bias") if has_bias else None
        self.has_bias = bool(has_bias)
        self.matmul, self.bias_add = P.MatMul(transpose_b=True), P.BiasAdd()
        self.activation = {None: None, "relu": P.ReLU(), "sigmoid": P.Sigmoid(), "tanh": P.Tanh()}[activation] if not isinstance(activation, Cell) else activation

    def construct(self, x):
        y = self.matmul(x, self.weight)
        if self.has_bias:
            y = self.bias_add(y, self.bias)
        return self.activation(y) if self.activation is not None else y


class MatMul(Cell):
    """nn.MatMul(transpose_x1=False, transpose_x2=False): broadcasting batched matmul (CrossLayer, deep_and_cross.py:130,146)."""

    def __init__(self, transpose_x1=False, transpose_x2=False):
        super().__init__()
        self.op = P.BatchMatMul(transpose_x1, transpose_x2)

    def construct(self, x1, x2):
        return self.op(x1, x2)
