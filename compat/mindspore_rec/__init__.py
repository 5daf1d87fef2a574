"""`mindspore_rec` (the reference's own package, mindspore_rec/__init__.py:18-21) on the MI355X engine."""
from mindspore_rec.ops import HashEmbeddingLookup
from mindspore_rec.train import RecModel

__all__ = ["RecModel", "HashEmbeddingLookup"]
