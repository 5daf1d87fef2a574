"""`mindspore.nn`."""
from . import layer, metrics, optim, wrap  # noqa: F401
from .cell import Cell, GraphCell  # noqa: F401
from .layer import ClipByNorm, Dense, Dropout, EmbeddingLookup, MatMul  # noqa: F401
from .metrics import Loss, Metric  # noqa: F401
from .optim import FTRL, Adam, LazyAdam, Optimizer  # noqa: F401
from .wrap import DistributedGradReducer, TrainOneStepCell, VirtualDatasetCellTriple, WithEvalCell, WithLossCell  # noqa: F401

from ..ops import operations as _P


class ReLU(Cell):
    def construct(self, x):
        return _P.ReLU()(x)


class Sigmoid(Cell):
    def construct(self, x):
        return _P.Sigmoid()(x)


class Tanh(Cell):
    def construct(self, x):
        return _P.Tanh()(x)
