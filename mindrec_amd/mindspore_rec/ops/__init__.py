from .embedding import HashEmbeddingLookup

__all__ = ["HashEmbeddingLookup"]
