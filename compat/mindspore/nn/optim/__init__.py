"""`mindspore.nn.optim`."""
from .optimizer import FTRL, Adam, LazyAdam, Optimizer  # noqa: F401
