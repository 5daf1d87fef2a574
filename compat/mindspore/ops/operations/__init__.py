"""`mindspore.ops.operations` (imported as `P`): the primitives the in-scope models instantiate.

Two kinds.  (1) Hot-path primitives -- Unique, Gather / SparseGatherV2 / EmbeddingLookup, MapTensorGet, MatMul and the
optimizer updates behind nn.optim -- run on the kernel set (`_kernels.K()`: libmrec_hip.so; no CPU path).  (2) Shape and
broadcasting glue (Reshape, ExpandDims, Concat, Cast, Mul, ReduceSum, ...) are operations on the device tensors' container
(torch), as are the scalar-per-sample ends (sigmoid cross entropy): they exist so that the reference's `construct` bodies
execute as written.  A recognised train step never runs them one by one -- it is lowered to the fused engine
(mindrec_amd/lowering.py)."""
import numbers

import torch

from ..._kernels import K
from ...common.tensor import Tensor, as_tensor
from .. import _grad
from ..primitive import Primitive
from . import _map_tensor_ops  # noqa: F401


def _t(x, like=None):
    """Tensor / python scalar -> torch tensor on the right device."""
    if isinstance(x, torch.Tensor):
        return x
    dev = like.device if isinstance(like, torch.Tensor) else None
    dt = like.dtype if isinstance(like, torch.Tensor) and isinstance(x, float) else None
    return torch.as_tensor(x, device=dev, dtype=dt)


def _w(x):
    return as_tensor(x) if isinstance(x, torch.Tensor) else x


class _Elementwise(Primitive):
    _fn = None

    def __call__(self, *xs):
        like = next((x for x in xs if isinstance(x, torch.Tensor)), None)
        return _w(type(self)._fn(*[_t(x, like) for x in xs]))


class Mul(_Elementwise):
    _fn = staticmethod(torch.mul)


class Add(_Elementwise):
    _fn = staticmethod(torch.add)


TensorAdd = Add


class Sub(_Elementwise):
    _fn = staticmethod(torch.sub)


class RealDiv(_Elementwise):
    _fn = staticmethod(torch.true_divide)


Div = RealDiv


class Maximum(_Elementwise):
    _fn = staticmethod(torch.maximum)


class Minimum(_Elementwise):
    _fn = staticmethod(torch.minimum)


class Pow(_Elementwise):
    _fn = staticmethod(torch.pow)


class Square(_Elementwise):
    _fn = staticmethod(torch.square)


class Sqrt(_Elementwise):
    _fn = staticmethod(torch.sqrt)


class Rsqrt(_Elementwise):
    _fn = staticmethod(torch.rsqrt)


class Exp(_Elementwise):
    _fn = staticmethod(torch.exp)


class Log(_Elementwise):
    _fn = staticmethod(torch.log)


class Neg(_Elementwise):
    _fn = staticmethod(torch.neg)


class Abs(_Elementwise):
    _fn = staticmethod(torch.abs)


class ReLU(_Elementwise):
    _fn = staticmethod(torch.relu)


class Sigmoid(_Elementwise):
    _fn = staticmethod(torch.sigmoid)


class Tanh(_Elementwise):
    _fn = staticmethod(torch.tanh)


class ZerosLike(_Elementwise):
    _fn = staticmethod(torch.zeros_like)


class OnesLike(_Elementwise):
    _fn = staticmethod(torch.ones_like)


class Cast(Primitive):
    def __call__(self, x, dtype):
        return _w(_t(x).to(dtype))


class BiasAdd(Primitive):
    """BiasAdd(data_format='NCHW'): x [N, C, ...] + bias [C]."""

    def __init__(self, data_format="NCHW"):
        super().__init__()
        self.data_format = data_format

    def __call__(self, x, bias):
        if bias.dim() != 1 or x.dim() < 2 or bias.shape[0] != x.shape[1]:
            raise ValueError(f"For 'BiasAdd', bias must be [C] with C = x.shape[1], but got x {tuple(x.shape)}, bias {tuple(bias.shape)}.")
        if x.dtype != bias.dtype:
            raise TypeError(f"For 'BiasAdd', x and bias must have the same dtype, but got {x.dtype} and {bias.dtype}.")
        return _w(x + bias.reshape((1, -1) + (1,) * (x.dim() - 2)))


class MatMul(Primitive):
    """MatMul(transpose_a=False, transpose_b=False) on 2-D operands: the DenseLayer contraction
    (models/wide_deep/src/wide_and_deep.py:96,124,130) -- on the matrix cores through the kernel set."""

    def __init__(self, transpose_a=False, transpose_b=False):
        super().__init__()
        self.transpose_a, self.transpose_b = bool(transpose_a), bool(transpose_b)

    def __call__(self, a, b):
        if a.dim() != 2 or b.dim() != 2:
            raise ValueError(f"For 'MatMul', both inputs must be 2-D, but got {tuple(a.shape)} and {tuple(b.shape)}.")
        if a.dtype != b.dtype:
            raise TypeError(f"For 'MatMul', the inputs must have the same dtype, but got {a.dtype} and {b.dtype}.")
        ka = a.shape[0] if self.transpose_a else a.shape[1]
        kb = b.shape[1] if self.transpose_b else b.shape[0]
        if ka != kb:
            raise ValueError(f"For 'MatMul', the reduction dimensions differ: {tuple(a.shape)} x {tuple(b.shape)} "
                             f"(transpose_a={self.transpose_a}, transpose_b={self.transpose_b}).")
        return _w(_grad.MatMul2D.apply(a, b, self.transpose_a, self.transpose_b))


class BatchMatMul(Primitive):
    """Batched / broadcasting matmul: in scope only CrossLayer's rank-1 scale x0 [B, D, 1] . (x_l^T w) [B, 1, 1]
    (models/deep_and_cross/src/deep_and_cross.py:146) -- a reduction of length 1, i.e. a broadcast multiply, HBM-bound."""

    def __init__(self, transpose_a=False, transpose_b=False):
        super().__init__()
        self.transpose_a, self.transpose_b = bool(transpose_a), bool(transpose_b)

    def __call__(self, a, b):
        if self.transpose_a:
            a = a.transpose(-1, -2)
        if self.transpose_b:
            b = b.transpose(-1, -2)
        if a.dim() == 2 and b.dim() == 2:
            return _w(_grad.MatMul2D.apply(a, b, False, False))
        if a.shape[-1] == 1:
            return _w(a * b)                     # [.., M, 1] . [.., 1, N]: outer product by broadcasting
        # a true batched contraction has no call site in the in-scope models and no hand-written kernel here: refuse rather than
        # hand it to a library GEMM without saying so
        from mindrec_amd.wide_deep_mlp import UnsupportedNet
        raise UnsupportedNet(f"BatchMatMul of {tuple(a.shape)} x {tuple(b.shape)}: only 2-D operands (the MFMA MatMul kernels) and "
                             f"reductions of length 1 (CrossLayer's rank-1 scale, deep_and_cross.py:146) have a kernel")


class ReduceSum(Primitive):
    def __init__(self, keep_dims=False, skip_mode=False):
        super().__init__()
        self.keep_dims = bool(keep_dims)

    _fn = staticmethod(torch.sum)

    def __call__(self, x, axis=()):
        if axis == () or axis is None:
            r = type(self)._fn(x)
            return _w(r.reshape((1,) * x.dim()) if self.keep_dims else r)
        return _w(type(self)._fn(x, dim=axis, keepdim=self.keep_dims))


class ReduceMean(ReduceSum):
    _fn = staticmethod(torch.mean)


class Reshape(Primitive):
    def __call__(self, x, shape):
        return _w(x.reshape(tuple(int(s) for s in shape)))


class ExpandDims(Primitive):
    def __call__(self, x, axis):
        return _w(x.unsqueeze(int(axis)))


class Squeeze(Primitive):
    def __init__(self, axis=()):
        super().__init__()
        self.axis = axis

    def __call__(self, x):
        if self.axis == () or self.axis is None:
            return _w(x.squeeze())
        axes = (self.axis,) if isinstance(self.axis, numbers.Integral) else tuple(self.axis)
        for a in sorted((a % x.dim() for a in axes), reverse=True):
            if x.shape[a] != 1:
                raise ValueError(f"For 'Squeeze', the dimension {a} of the input must be 1, but got {x.shape[a]}.")
            x = x.squeeze(a)
        return _w(x)


class Tile(Primitive):
    """Tile(x, multiples): x repeated multiples[i] times along axis i (models/deepfm/src/deepfm.py:204 instantiates one)."""

    def __call__(self, x, multiples):
        m = tuple(int(v) for v in multiples)
        if len(m) < x.dim():
            raise ValueError(f"For 'Tile', the length of 'multiples' must be >= the rank of the input ({x.dim()}), but got {len(m)}.")
        return _w(x.repeat(m))


class Transpose(Primitive):
    def __call__(self, x, perm):
        return _w(x.permute(tuple(perm)))


class Concat(Primitive):
    def __init__(self, axis=0):
        super().__init__()
        self.axis = int(axis)

    def __call__(self, xs):
        return _w(torch.cat(tuple(xs), dim=self.axis))


class Shape(Primitive):
    def __call__(self, x):
        return tuple(x.shape)


TensorShape = DynamicShape = Shape


class DType(Primitive):
    def __call__(self, x):
        return x.dtype


class Rank(Primitive):
    def __call__(self, x):
        return x.dim()


class Size(Primitive):
    def __call__(self, x):
        return x.numel()


class Fill(Primitive):
    """Fill(dtype, shape, value) -- `sens` of a train step (wide_and_deep.py:479-480)."""

    def __call__(self, dtype, shape, value):
        from ... import context
        return Tensor(torch.full(tuple(shape), value, dtype=dtype, device=context._torch_device()))


class OneHot(Primitive):
    def __init__(self, axis=-1):
        super().__init__()
        self.axis = axis

    def __call__(self, indices, depth, on_value, off_value):
        oh = torch.nn.functional.one_hot(indices.long(), int(depth)).to(on_value.dtype)
        return _w(oh * on_value + (1 - oh) * off_value)


class Depend(Primitive):
    def __call__(self, value, expr):
        return value


class StopGradient(Primitive):
    def __call__(self, x):
        return _w(x.detach())


class Assign(Primitive):
    def __call__(self, variable, value):
        with torch.no_grad():
            variable.as_subclass(torch.Tensor).copy_(_t(value, variable))
        return variable


class AssignAdd(Primitive):
    def __call__(self, variable, value):
        with torch.no_grad():
            variable.as_subclass(torch.Tensor).add_(_t(value, variable))
        return variable


class SigmoidCrossEntropyWithLogits(Primitive):
    """loss = max(x, 0) - x * z + log(1 + exp(-|x|)), elementwise (wide_and_deep.py:342,354)."""

    def __call__(self, logits, label):
        if logits.shape != label.shape:
            raise ValueError(f"For 'SigmoidCrossEntropyWithLogits', logits and label must have the same shape, "
                             f"but got {tuple(logits.shape)} and {tuple(label.shape)}.")
        return _w(torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-torch.abs(logits))))


# ---- the embedding path: kernel-set primitives ------------------------------------------------------------------------
def _ids_ok(ids, who):
    if ids.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"For '{who}', the indices must be int32 or int64, but got {ids.dtype}.")


class Unique(Primitive):
    """(y, idx) with y[idx[i]] == x[i], y in first-occurrence order, idx int32 (embedding.py:153,192)."""

    def __call__(self, x):
        if x.dim() != 1:
            raise ValueError(f"For 'Unique', the input must be 1-D, but got {tuple(x.shape)}.")
        _ids_ok(x, "Unique")
        y, idx = K().unique(x.as_subclass(torch.Tensor))
        return _w(y), _w(idx)


class Gather(Primitive):
    """Gather(params, indices, axis): row gather for axis 0 with the dense UnsortedSegmentSum bprop
    (deep_and_cross.py:199; embedding.py:150,194)."""

    def __call__(self, params, indices, axis=0):
        _ids_ok(indices, "Gather")
        if axis != 0:
            return _w(torch.index_select(params, int(axis), indices.reshape(-1).long()).reshape(
                params.shape[:axis] + tuple(indices.shape) + params.shape[axis + 1:]))
        if params.dim() == 1:
            return _w(_grad.GatherDense.apply(params.reshape(-1, 1), indices).reshape(tuple(indices.shape)))
        tail = tuple(params.shape[1:])
        out = _grad.GatherDense.apply(params.reshape(params.shape[0], -1), indices)
        return _w(out.reshape(tuple(indices.shape) + tail))


GatherV2 = Gather


class SparseGatherV2(Primitive):
    """Gather whose bprop is a RowTensor when `params` is a Parameter."""

    def __call__(self, params, indices, axis=0):
        from ...common.parameter import Parameter
        _ids_ok(indices, "SparseGatherV2")
        if axis != 0 or params.dim() != 2:
            raise ValueError("For 'SparseGatherV2', only axis 0 of a 2-D table is supported.")
        if isinstance(params, Parameter) and torch.is_grad_enabled() and params.requires_grad:
            out = _grad.GatherRowsSparse.apply(params.row_hook(), params, indices)
        else:
            out = _grad.GatherDense.apply(params, indices)
        return _w(out.reshape(tuple(indices.shape) + (params.shape[1],)))


class EmbeddingLookup(Primitive):
    """EmbeddingLookup(params, indices, offset): rows of `params` at indices - offset, zeros outside the table
    (SURVEY A.2); RowTensor bprop like SparseGatherV2."""

    def __call__(self, params, indices, offset=0):
        if offset:
            indices = indices - int(offset)
        return SparseGatherV2.__call__(self, params, indices, 0)


# (names the wider mindspore.ops namespace has and in-scope scripts import without calling)
class Dropout(Primitive):
    def __init__(self, keep_prob=0.5, Seed0=0, Seed1=0):
        super().__init__()
        self.keep_prob = float(keep_prob)

    def __call__(self, x):
        from ...nn.layer.basic import _dropout
        return _dropout(x, self.keep_prob, self)


class L2Normalize(Primitive):
    def __init__(self, axis=0, epsilon=1e-4):
        super().__init__()
        self.axis, self.epsilon = axis, float(epsilon)

    def __call__(self, x):
        n = torch.sqrt(torch.clamp((x * x).sum(dim=self.axis, keepdim=True), min=self.epsilon))
        return _w(x / n)
