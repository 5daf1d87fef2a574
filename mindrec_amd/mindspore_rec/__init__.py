"""Host-side mirror of the `mindspore_rec` package (mindspore_rec/__init__.py): the two public
classes of MindRec, over the MI355X kernels."""
from .ops import HashEmbeddingLookup
from .train import RecModel

__all__ = ["HashEmbeddingLookup", "RecModel"]
