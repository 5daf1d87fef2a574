"""`mindspore_rec.HashEmbeddingLookup` on MI355X -- mirrors mindspore_rec/ops/embedding.py:47-206
(same constructor arguments, attributes, error and op order); arithmetic is libmrec_hip.so."""
import sys

import torch

from ... import _validator as validator
from ... import context, nn, ops
from ...experimental import MapParameter, RowGrad


class HashEmbeddingLookup(nn.Cell):
    """Dynamic-vocabulary embedding: indices are arbitrary int32/int64 keys of a MapParameter.

    construct(indices) = reshape -> Unique -> MapTensorGet(insert default) -> Gather back ->
    reshape (-> ClipByNorm)   [embedding.py:184-206].  sparse=False skips the Unique in the
    reference; here the index probe always deduplicates first (same result, fewer probes).
    """

    def __init__(self, embedding_size, key_dtype=torch.int32, param_init="normal", sparse=True, max_norm=None,
                 permit_filter_value=1, evict_filter_value=sys.maxsize, vocab_cache_size=0, capacity=1 << 20,
                 device="cuda:0"):
        super().__init__()
        validator.check_value_type("sparse", sparse, [bool], self.cls_name)
        vocab_cache_size = validator.check_non_negative_int(vocab_cache_size, "vocab_cache_size")
        enable_ps = context.get_ps_context("enable_ps")
        enable_cache = vocab_cache_size > 0
        if enable_cache and not enable_ps:          # embedding.py:105-110, message verbatim
            raise RuntimeError(
                "The configuration of 'vocab_cache_size' is greater than 0 means enable embedding cache mode, "
                "this mode only support in parameter server training "
                "mode, please enable ps mode by 'context.set_ps_context(enable_ps=True)'")
        self.use_dense_tensor = bool(enable_ps and enable_cache and context.get_ps_context("ms_role") == "MS_WORKER")
        if self.use_dense_tensor:                   # embedding.py:119-130: worker-side dense cache table
            self.embedding_lookup = nn.EmbeddingLookup(vocab_size=vocab_cache_size, embedding_size=embedding_size,
                                                       param_init=param_init, target="DEVICE", max_norm=max_norm,
                                                       sparse=sparse, vocab_cache_size=vocab_cache_size, device=device)
            self.embedding_table = self.embedding_lookup.embedding_table
            return
        self.forward_unique = sparse
        self.embedding_size = validator.check_positive_int(embedding_size, "embedding_size", self.cls_name)
        self.embedding_table = MapParameter(key_dtype=key_dtype, value_dtype=torch.float32,
                                            value_shape=(embedding_size,), default_value=param_init,
                                            name="embedding_table", permit_filter_value=permit_filter_value,
                                            evict_filter_value=evict_filter_value, capacity=capacity, device=device)
        self.embedding_table.unique = self.forward_unique
        self.max_norm = None
        if max_norm is not None:
            self.max_norm = validator.check_positive_float(max_norm, "max_norm", self.cls_name)
        self._hook = torch.nn.Parameter(torch.zeros((), device=device))

    def construct(self, indices):
        if self.use_dense_tensor:
            return self.embedding_lookup(indices)
        t = self.embedding_table
        shape = tuple(indices.shape) + (self.embedding_size,)
        flat = t._keys(indices)
        training = torch.is_grad_enabled() and t.requires_grad
        # a training lookup needs the Unique anyway (the optimizer's inverted index is built from it): the index is then
        # probed once per unique key; otherwise every position probes the index itself and no Unique is run
        d, rows_u, rows_pos = t.lookup_rows(flat, insert=True, dedup=ops.unique(flat) if training else None)
        out = ops.gather_rows(t.values, rows_pos).view(shape)
        if training:
            def plan_fn():
                plan = ops.group_by_inverse(d)
                plan.uniq_buf = t.admitted_rows(rows_u)     # groups -> table rows (un-admitted keys -> -1)
                return plan
            out = nn._RecordRowGrad.apply(out, self._hook, t, plan_fn)
        if self.max_norm is not None:
            out = nn.clip_by_norm(out, self.max_norm, axes=tuple(range(indices.dim(), out.dim())))
        return out
