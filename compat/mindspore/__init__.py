"""`mindspore`-named surface over the MI355X-native engine (BASELINE north_star: "so the existing Wide&Deep and DCN
model scripts run unmodified"; SURVEY Appendix B lists the symbols).

This is NOT MindSpore and not a port of it: it is the slice of its Python API that the reference's in-scope code imports
(mindspore_rec/, models/wide_deep/, models/deep_and_cross/, ci/st/online_learning/), written from the public API's
documented behaviour, with torch tensors as the device container and every hot-path primitive on libmrec_hip.so
(`_kernels.py`).  Put this directory (`compat/`) on PYTHONPATH; `mindspore_rec` beside it is this repo's implementation of
the reference's own package.
"""
from . import _checkparam, common, communication, context, dataset, experimental, log, nn, ops, parallel, train  # noqa: F401
from .common import dtype as _dt
from .common import set_seed  # noqa: F401
from .common.dtype import *  # noqa: F401,F403
from .common.dtype import (bfloat16, bool_, float16, float32, float64, int8, int16, int32, int64, uint8)  # noqa: F401
from .common.parameter import Parameter, ParameterTuple  # noqa: F401
from .common.sparse_tensor import RowTensor  # noqa: F401
from .common.tensor import Tensor  # noqa: F401
from .context import GRAPH_MODE, PYNATIVE_MODE, ParallelMode, get_context, set_context  # noqa: F401
from .train.model import Model  # noqa: F401
from .train.serialization import load_checkpoint, load_param_into_net, save_checkpoint  # noqa: F401

dtype = _dt
__version__ = "2.0.0+mindrec_amd"
