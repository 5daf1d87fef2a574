"""`mindspore.dataset.config`."""
import numpy as np

_seed = {"v": 0}


def set_seed(seed):
    _seed["v"] = int(seed)
    np.random.seed(int(seed))


def get_seed():
    return _seed["v"]


def set_num_parallel_workers(n):
    return None
