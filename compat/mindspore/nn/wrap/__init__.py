"""`mindspore.nn.wrap`."""
from . import cell_wrapper, grad_reducer  # noqa: F401
from .cell_wrapper import TrainOneStepCell, VirtualDatasetCellTriple, WithEvalCell, WithLossCell  # noqa: F401
from .grad_reducer import DistributedGradReducer  # noqa: F401
