"""TEST INFRASTRUCTURE ONLY: the engines with their op set replaced by the oracle's restatements (tests/_oracle_ops.py) and the
CPU allowed -- the oracle-side engine the GPU engines are checked against step for step, and the way the multi-rank host logic
runs under gloo on a machine without a GPU.  The product classes have no such switch."""
import _oracle_ops
from mindrec_amd.deep_cross import DeepCrossEngine
from mindrec_amd.deepfm import DeepFMEngine
from mindrec_amd.wide_deep import WideDeepEngine


class OracleWideDeepEngine(WideDeepEngine):
    _kernels = _oracle_ops
    _allow_cpu = True


class OracleDeepCrossEngine(DeepCrossEngine):
    _kernels = _oracle_ops
    _allow_cpu = True


class OracleDeepFMEngine(DeepFMEngine):
    _kernels = _oracle_ops
    _allow_cpu = True
