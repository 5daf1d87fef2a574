"""`mindspore.ops.functional` (imported as `F`): function forms of a few primitives."""
from . import operations as P

_reshape, _cast, _shape, _dtype = P.Reshape(), P.Cast(), P.Shape(), P.DType()


def rank(x):
    return x.dim()


def depend(value, expr):
    """Orders `expr` before `value` in MindSpore's graph; eager execution already ran it (wide_and_deep.py:490-492)."""
    return value


def shape(x):
    return tuple(x.shape)


def dtype(x):
    return x.dtype


def reshape(x, shp):
    return _reshape(x, shp)


def cast(x, t):
    return _cast(x, t)


def stop_gradient(x):
    return P.StopGradient()(x)


def tensor_mul(a, b):
    return P.Mul()(a, b)


def tensor_add(a, b):
    return P.Add()(a, b)


def partial(fn, *args):
    import functools
    return functools.partial(fn, *args)
