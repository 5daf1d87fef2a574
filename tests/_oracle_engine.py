"""TEST INFRASTRUCTURE ONLY: the engines with their op set replaced by the oracle's restatements (tests/_oracle_ops.py), their
dense nets by torch restatements (tests/_torch_net.py) and the CPU allowed -- the oracle-side engine the GPU engines are checked
against step for step, and the way the multi-rank host logic runs under gloo on a machine without a GPU.  The product classes
have no such switch: no CPU path, no library GEMM, no autograd."""
import _oracle_ops
from _torch_net import TorchDeepCrossMixin, TorchDeepFMMixin, TorchDenseNetMixin
from mindrec_amd.deep_cross import DeepCrossEngine
from mindrec_amd.deepfm import DeepFMEngine
from mindrec_amd.wide_deep import WideDeepEngine


class OracleWideDeepEngine(TorchDenseNetMixin, WideDeepEngine):
    _kernels = _oracle_ops
    _allow_cpu = True


class OracleDeepCrossEngine(TorchDeepCrossMixin, DeepCrossEngine):
    _kernels = _oracle_ops
    _allow_cpu = True


class OracleDeepFMEngine(TorchDeepFMMixin, DeepFMEngine):
    _kernels = _oracle_ops
    _allow_cpu = True


# GPU engines whose dense net is the torch restatement (embedding path on the HIP kernels): for nets the kernels do not cover
class TorchNetWideDeepEngine(TorchDenseNetMixin, WideDeepEngine):
    pass


class TorchNetDeepCrossEngine(TorchDeepCrossMixin, DeepCrossEngine):
    pass


class TorchNetDeepFMEngine(TorchDeepFMMixin, DeepFMEngine):
    pass
