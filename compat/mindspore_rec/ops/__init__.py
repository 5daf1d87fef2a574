from mindspore_rec.ops.embedding import HashEmbeddingLookup

__all__ = ["HashEmbeddingLookup"]
