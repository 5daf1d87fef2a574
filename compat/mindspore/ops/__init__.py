"""`mindspore.ops`."""
from . import composite, functional, operations  # noqa: F401
from .composite import GradOperation, HyperMap, clip_by_value, tensor_dot  # noqa: F401
from .functional import cast, depend, dtype, rank, reshape, shape, stop_gradient  # noqa: F401
from .operations import *  # noqa: F401,F403
from .operations import (Abs, Add, Assign, AssignAdd, BatchMatMul, BiasAdd, Cast, Concat, Depend, Div, Dropout, DType,  # noqa: F401
                         EmbeddingLookup, Exp, ExpandDims, Fill, Gather, GatherV2, L2Normalize, Log, MatMul, Maximum, Minimum, Mul,
                         Neg, OneHot, OnesLike, Pow, Rank, RealDiv, ReduceMean, ReduceSum, ReLU, Reshape, Rsqrt, Shape,
                         Sigmoid, SigmoidCrossEntropyWithLogits, Size, SparseGatherV2, Sqrt, Square, Squeeze, StopGradient, Sub,
                         Tanh, TensorAdd, Transpose, Unique, ZerosLike)
from .primitive import Primitive, PrimitiveWithInfer, constexpr, prim_attr_register  # noqa: F401
