"""`mindspore.nn.EmbeddingLookup` as the in-scope models build it (models/wide_deep/src/wide_and_deep.py:234-290;
SURVEY Appendix A.3).  One row gather on the kernel set; `sparse=True` makes the table's gradient a RowTensor (the
Unique -> SparseGatherV2 -> Gather-back chain of MindSpore is this gather plus the dedup the sparse optimizers' apply
kernel does anyway), `sparse=False` a dense [V, D] UnsortedSegmentSum."""
import torch

from ... import _checkparam as validator
from ...common.initializer import initializer
from ...common.parameter import Parameter
from ...ops import operations as P
from ..cell import Cell
from .basic import ClipByNorm


class EmbeddingLookup(Cell):
    BATCH_SLICE = "batch_slice"
    FIELD_SLICE = "field_slice"
    TABLE_ROW_SLICE = "table_row_slice"
    TABLE_COLUMN_SLICE = "table_column_slice"

    def __init__(self, vocab_size, embedding_size, param_init="normal", target="CPU", slice_mode="batch_slice",
                 manual_shapes=None, max_norm=None, sparse=True, vocab_cache_size=0, dtype=torch.float32):
        super().__init__()
        self.vocab_size = validator.check_positive_int(vocab_size, "vocab_size", self.cls_name)
        self.embedding_size = validator.check_positive_int(embedding_size, "embedding_size", self.cls_name)
        self.vocab_cache_size = validator.check_non_negative_int(vocab_cache_size, "vocab_cache_size", self.cls_name)
        validator.check_value_type("sparse", sparse, [bool], self.cls_name)
        if target not in ("CPU", "DEVICE"):
            raise ValueError(f"For '{self.cls_name}', the 'target' must be one of values in ('CPU', 'DEVICE'), but got {target}.")
        if slice_mode not in (self.BATCH_SLICE, self.FIELD_SLICE, self.TABLE_ROW_SLICE, self.TABLE_COLUMN_SLICE):
            raise ValueError(f"For '{self.cls_name}', the 'slice_mode' must be in "
                             f"['batch_slice', 'field_slice', 'table_row_slice', 'table_column_slice'], but got {slice_mode!r}.")
        if not sparse and target == "CPU":
            raise ValueError(f"For '{self.cls_name}', 'sparse' must be True when 'target' is \"CPU\", but got 'sparse': {sparse} "
                             f"and 'target': {target}")
        self.target, self.sparse, self.slice_mode, self.manual_shapes = target, sparse, slice_mode, manual_shapes
        self.cache_enable = self.vocab_cache_size > 0
        self.forward_unique = bool(sparse)
        self.embedding_table = Parameter(initializer(param_init, [self.vocab_size, self.embedding_size], dtype), name="embedding_table")
        self.max_norm = None if max_norm is None else validator.check_positive_float(max_norm, "max_norm", self.cls_name)
        self.gather = P.SparseGatherV2() if sparse else P.Gather()
        self.embeddinglookup = P.EmbeddingLookup()
        if self.max_norm is not None:
            self.clip = ClipByNorm(axis=None)

    def construct(self, indices):
        if self.target == "CPU":
            out = self.embeddinglookup(self.embedding_table, indices, 0)
        else:
            out = self.gather(self.embedding_table, indices, 0)
        if self.max_norm is not None:
            self.clip.axis = tuple(range(indices.dim(), out.dim()))
            out = self.clip(out, self.max_norm)
        return out
