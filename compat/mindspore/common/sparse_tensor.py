"""`mindspore.RowTensor` and the MapTensor-typed gradient: what the sparse lookups' bprops hand the optimizers."""


class RowTensor:
    """RowTensor(indices, values, dense_shape): rows `indices` of a [V, D] tensor hold `values`; duplicate indices add
    (the bprop of SparseGatherV2 / EmbeddingLookup, not yet deduplicated)."""

    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = indices, values, tuple(dense_shape)

    def __repr__(self):
        return f"RowTensor(indices={tuple(self.indices.shape)}, values={tuple(self.values.shape)}, dense_shape={self.dense_shape})"


class MapTensorGrad:
    """Gradient of MapTensorGet w.r.t. its MapParameter: (keys, value gradients), duplicates add."""

    def __init__(self, keys, values):
        self.keys, self.values = keys, values

    def __repr__(self):
        return f"MapTensorGrad(keys={tuple(self.keys.shape)}, values={tuple(self.values.shape)})"
