"""`mindspore_rec.HashEmbeddingLookup` over the `mindspore` surface of this repo -- the interface of
mindspore_rec/ops/embedding.py:47-206 (constructor arguments, attributes, the RuntimeError of :105-110, output shape and
gradient type), implemented as ONE MapTensorGet per call: the device index deduplicates the keys while it probes
(csrc/mrec_hash.hip ranks first occurrences in the probe chain), so the reference's Unique -> MapTensorGet -> Gather-back
(:189-195) and its non-unique branch (:196-200) are the same kernel sequence here and return the same values."""
import sys

import mindspore as ms
from mindspore import _checkparam as validator
from mindspore import context, nn
from mindspore.experimental import MapParameter
from mindspore.nn.layer.basic import ClipByNorm
from mindspore.ops.operations._map_tensor_ops import MapTensorGet


class HashEmbeddingLookup(nn.Cell):
    def __init__(self, embedding_size, key_dtype=ms.int32, param_init="normal", sparse=True, max_norm=None,
                 permit_filter_value=1, evict_filter_value=sys.maxsize, vocab_cache_size=0, capacity=None):
        super().__init__()
        validator.check_value_type("sparse", sparse, [bool], self.cls_name)
        vocab_cache_size = validator.check_non_negative_int(vocab_cache_size, "vocab_cache_size")
        ps_on = context.get_ps_context("enable_ps")
        cached = vocab_cache_size > 0
        if cached and not ps_on:
            raise RuntimeError(
                "The configuration of 'vocab_cache_size' is greater than 0 means enable embedding cache mode, "
                "this mode only support in parameter server training "
                "mode, please enable ps mode by 'context.set_ps_context(enable_ps=True)'")
        self.use_dense_tensor = bool(ps_on and cached and context.get_ps_context("ms_role") == "MS_WORKER")
        if self.use_dense_tensor:
            # worker side of the embedding-cache mode: ids arrive already mapped to cache slots, the device holds a plain table
            self.embedding_lookup = nn.EmbeddingLookup(vocab_size=vocab_cache_size, embedding_size=embedding_size, param_init=param_init,
                                                       target="DEVICE", max_norm=max_norm, sparse=sparse,
                                                       vocab_cache_size=vocab_cache_size)
            self.embedding_table = self.embedding_lookup.embedding_table
            return
        self.forward_unique = sparse
        self.embedding_size = validator.check_positive_int(embedding_size, "embedding_size", self.cls_name)
        kw = {} if capacity is None else {"capacity": capacity}
        self.embedding_table = MapParameter(key_dtype=key_dtype, value_dtype=ms.float32, value_shape=(embedding_size,),
                                            default_value=param_init, name="embedding_table",
                                            permit_filter_value=permit_filter_value, evict_filter_value=evict_filter_value, **kw)
        self.embedding_table.unique = self.forward_unique
        self.embedding_table.cache_enable = cached
        self.map_tensor_get = MapTensorGet(True)
        self.max_norm = None
        if max_norm is not None:
            self.max_norm = ms.Tensor(validator.check_positive_float(max_norm, "max_norm", self.cls_name), dtype=ms.float32)

    def construct(self, indices):
        if self.use_dense_tensor:
            return self.embedding_lookup(indices)
        out = self.map_tensor_get(self.embedding_table, indices)          # [*indices.shape, embedding_size]
        if self.max_norm is not None:
            out = ClipByNorm(tuple(range(indices.dim(), out.dim())))(out, self.max_norm)
        return out
