"""`mindspore.nn.wrap.cell_wrapper`: WithLossCell, TrainOneStepCell, VirtualDatasetCellTriple."""
import torch

from ...ops import composite as C
from ...ops import operations as P
from ..cell import Cell


class WithLossCell(Cell):
    def __init__(self, backbone, loss_fn):
        super().__init__(auto_prefix=False)
        self._backbone, self._loss_fn = backbone, loss_fn

    def construct(self, data, label):
        return self._loss_fn(self._backbone(data), label)

    @property
    def backbone_network(self):
        return self._backbone


class WithEvalCell(Cell):
    def __init__(self, network, loss_fn, add_cast_fp32=False):
        super().__init__(auto_prefix=False)
        self._network, self._loss_fn = network, loss_fn

    def construct(self, data, label):
        out = self._network(data)
        return self._loss_fn(out, label), out, label


class TrainOneStepCell(Cell):
    """TrainOneStepCell(network, optimizer, sens=1.0): loss -> gradients of the optimizer's parameters -> optimizer."""

    def __init__(self, network, optimizer, sens=1.0):
        super().__init__(auto_prefix=False)
        self.network = network
        self.network.set_grad()
        self.optimizer = optimizer
        self.weights = optimizer.parameters
        self.grad = C.GradOperation(get_by_list=True, sens_param=True)
        self.sens = float(sens)
        self.grad_reducer = None
        from ... import context
        if context.get_auto_parallel_context("parallel_mode") in (context.ParallelMode.DATA_PARALLEL, context.ParallelMode.HYBRID_PARALLEL):
            from .grad_reducer import DistributedGradReducer
            self.grad_reducer = DistributedGradReducer(self.weights, context.get_auto_parallel_context("gradients_mean"),
                                                       context.get_auto_parallel_context("device_num"))

    def construct(self, *inputs):
        loss = self.network(*inputs)
        sens = P.Fill()(loss.dtype, tuple(loss.shape), self.sens)
        grads = self.grad(self.network, self.weights)(*inputs, sens)
        if self.grad_reducer is not None:
            grads = self.grad_reducer(grads)
        self.optimizer(grads)
        return loss


class VirtualDatasetCellTriple(Cell):
    """Auto-parallel's virtual-dataset marker around a three-input network
    (models/wide_deep/train_and_eval_parameter_server_distribute.py:43-50): an identity here."""

    def __init__(self, backbone):
        super().__init__(auto_prefix=False)
        self._backbone = backbone

    def construct(self, a, b, c):
        return self._backbone(a, b, c)
