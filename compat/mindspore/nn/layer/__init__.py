"""`mindspore.nn.layer`."""
from . import basic, embedding  # noqa: F401
from .basic import ClipByNorm, Dense, Dropout, MatMul  # noqa: F401
from .embedding import EmbeddingLookup  # noqa: F401
