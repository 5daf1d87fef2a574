"""`mindspore.train`."""
from . import callback, serialization  # noqa: F401
from .callback import (Callback, CheckpointConfig, LossMonitor, ModelCheckpoint, RunContext, TimeMonitor)  # noqa: F401
from .model import Model  # noqa: F401
from .serialization import load_checkpoint, load_param_into_net, save_checkpoint  # noqa: F401
