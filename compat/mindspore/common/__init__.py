"""`mindspore.common`."""
from . import dtype, initializer  # noqa: F401
from .dtype import *  # noqa: F401,F403
from .parameter import Parameter, ParameterTuple  # noqa: F401
from .tensor import Tensor  # noqa: F401


def set_seed(seed):
    """`set_seed(1000)` (models/wide_deep/train_and_eval_distribute.py:72): seeds the initializers' counter-based stream,
    numpy's and torch's host generators."""
    import numpy as np
    import torch
    initializer._set_global_seed(seed)
    np.random.seed(int(seed))
    torch.manual_seed(int(seed))


def get_seed():
    return initializer._state["seed"]
