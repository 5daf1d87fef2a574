"""The slice of `mindspore.nn` on the hot path: Cell, EmbeddingLookup and the optimizers the
reference builds in TrainStepWrap (models/wide_deep/src/wide_and_deep.py:415-445,
models/deep_and_cross/src/deep_and_cross.py:342-344).  Names, argument order and defaults follow
MindSpore [EXT, SURVEY.md Appendix A.3-A.5]; arithmetic is libmrec_hip.so.
"""
import numpy as np
import torch

from . import _validator as validator
from . import ops
from .experimental import MapParameter, RowGrad


class Cell(torch.nn.Module):
    """nn.Cell: `construct` is the forward."""

    def forward(self, *a, **kw):
        return self.construct(*a, **kw)

    @property
    def cls_name(self):
        return type(self).__name__

    def set_train(self, mode=True):
        self.train(mode)
        return self

    def trainable_params(self):
        ps = [p for p in self.parameters() if p.requires_grad]
        for m in self.modules():
            t = getattr(m, "embedding_table", None)
            if isinstance(t, (MapParameter, TableParameter)) and t not in ps:
                ps.append(t)
        return ps


class GraphCell(Cell):
    """Marker for a compiled-graph network: online_train refuses it in sink mode (rec_model.py:153-156)."""


class TableParameter:
    """A dense [V, D] embedding table parameter.  Holds the weights (optionally as a view of a
    fused [V, 3D] row so Adam state sits beside them), optimizer slots and the sparse / dense
    gradients recorded by the lookup's backward."""

    def __init__(self, data, name="embedding_table", sparse=True):
        self.data = data
        self.name = name
        self.sparse = sparse
        self.requires_grad = True
        self.slots = {}
        self.sparse_grads = []
        self.grad = None

    @property
    def shape(self):
        return self.data.shape

    def add_slot(self, name, init=0.0):
        if name not in self.slots:
            self.slots[name] = {"table": torch.full_like(self.data, float(init)).contiguous(), "init": float(init)}
        return self.slots[name]["table"]


class _RecordRowGrad(torch.autograd.Function):
    """Identity on the gathered rows whose backward files the incoming gradient as a RowGrad on the
    parameter (the bprop of SparseGatherV2 / MapTensorGet: a RowTensor)."""

    @staticmethod
    def forward(ctx, out, hook, param, plan_fn):
        ctx.param, ctx.plan_fn = param, plan_fn
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        p = ctx.param
        p.sparse_grads.append(RowGrad(ctx.plan_fn(), g.reshape(-1, g.shape[-1]).contiguous()))
        return None, torch.zeros((), device=g.device), None, None


class EmbeddingLookup(Cell):
    """nn.EmbeddingLookup(vocab_size, embedding_size, param_init='normal', target='CPU',
    slice_mode='batch_slice', manual_shapes=None, max_norm=None, sparse=True, vocab_cache_size=0)
    as used at wide_and_deep.py:277-290.  sparse=True: Unique + SparseGatherV2 + Gather back with a
    RowTensor gradient; sparse=False: plain Gather with a dense [V, D] gradient."""
    BATCH_SLICE = "batch_slice"
    FIELD_SLICE = "field_slice"
    TABLE_ROW_SLICE = "table_row_slice"
    TABLE_COLUMN_SLICE = "table_column_slice"

    def __init__(self, vocab_size, embedding_size, param_init="normal", target="CPU", slice_mode="batch_slice",
                 manual_shapes=None, max_norm=None, sparse=True, vocab_cache_size=0, device="cuda:0", seed=0):
        super().__init__()
        self.vocab_size = validator.check_positive_int(vocab_size, "vocab_size", self.cls_name)
        self.embedding_size = validator.check_positive_int(embedding_size, "embedding_size", self.cls_name)
        self.vocab_cache_size = validator.check_non_negative_int(vocab_cache_size, "vocab_cache_size", self.cls_name)
        validator.check_value_type("sparse", sparse, [bool], self.cls_name)
        if target not in ("CPU", "DEVICE"):
            raise ValueError(f"For '{self.cls_name}', the 'target' must be one of values in ('CPU', 'DEVICE'), "
                             f"but got {target}.")
        if slice_mode not in (self.BATCH_SLICE, self.FIELD_SLICE, self.TABLE_ROW_SLICE, self.TABLE_COLUMN_SLICE):
            raise ValueError(f"For '{self.cls_name}', unknown 'slice_mode' {slice_mode}.")
        self.target, self.sparse, self.slice_mode = target, sparse, slice_mode
        self.max_norm = None if max_norm is None else validator.check_positive_float(max_norm, "max_norm", self.cls_name)
        dev = torch.device(device)
        table = torch.empty((self.vocab_size, self.embedding_size), dtype=torch.float32, device=dev)
        if isinstance(param_init, str):
            if param_init == "normal":
                ops.fill_normal_(table, seed, 0.01)
            elif param_init in ("zeros", "ones"):
                table.fill_(0.0 if param_init == "zeros" else 1.0)
            else:
                raise ValueError(f"For '{self.cls_name}', unsupported 'param_init' {param_init!r}.")
        elif torch.is_tensor(param_init):
            table.copy_(param_init)
        else:
            table.fill_(float(param_init))
        self.embedding_table = TableParameter(table, "embedding_table", sparse)
        self._hook = torch.nn.Parameter(torch.zeros((), device=dev))   # gives autograd an edge to the backward

    def construct(self, indices):
        t = self.embedding_table
        out = ops.gather_rows(t.data, indices)
        if torch.is_grad_enabled():
            ids = indices
            out = _RecordRowGrad.apply(out, self._hook, t, lambda: ops.sparse_plan(ids))
        if self.max_norm is not None:
            out = clip_by_norm(out, self.max_norm, axes=tuple(range(indices.dim(), out.dim())))
        return out


def clip_by_norm(x, clip_norm, axes):
    """nn.ClipByNorm [EXT]: x * clip_norm / max(l2norm(x), clip_norm) over `axes`."""
    n = torch.sqrt((x * x).sum(dim=axes, keepdim=True))
    return x * clip_norm / torch.maximum(n, torch.as_tensor(clip_norm, device=x.device, dtype=x.dtype))


# ---- optimizers ---------------------------------------------------------------------------------
class _Optimizer:
    def __init__(self, params, learning_rate, loss_scale, weight_decay):
        self.parameters = list(params)
        if not self.parameters:
            raise ValueError("Optimizer got an empty parameter list.")
        if weight_decay != 0:
            raise NotImplementedError("weight_decay != 0 is not on the reference's hot path (all call sites use 0)")
        self.learning_rate = float(learning_rate)
        self.loss_scale = float(loss_scale)
        self.reciprocal_scale = 1.0 / float(loss_scale)
        self.global_step = 0

    def _tables(self, p, names_inits):
        """(weights, state tables...) for a MapParameter / TableParameter / dense tensor."""
        if isinstance(p, MapParameter):
            return p.values, [p.add_slot(n, i) for n, i in names_inits]
        if isinstance(p, TableParameter):
            return p.data, [p.add_slot(n, i) for n, i in names_inits]
        st = self._dense_state.setdefault(id(p), [torch.full_like(p.data, float(i)) for _, i in names_inits])
        return p.data, st

    def __call__(self, gradients=None):
        return self.step(gradients)


class LazyAdam(_Optimizer):
    """nn.LazyAdam(params, learning_rate=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, use_locking=False,
    use_nesterov=False, weight_decay=0.0, loss_scale=1.0): Adam on dense gradients, per-touched-row
    Adam on RowTensor gradients (wide_and_deep.py:420-422)."""
    lazy = True

    def __init__(self, params, learning_rate=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, use_locking=False,
                 use_nesterov=False, weight_decay=0.0, loss_scale=1.0):
        super().__init__(params, learning_rate, loss_scale, weight_decay)
        if not 0.0 < beta1 < 1.0 or not 0.0 < beta2 < 1.0:
            raise ValueError("For 'LazyAdam', beta1 and beta2 must be in (0, 1).")
        if eps <= 0:
            raise ValueError("For 'LazyAdam', eps must be > 0.")
        self.beta1, self.beta2, self.eps = np.float32(beta1), np.float32(beta2), float(eps)
        self.use_nesterov = bool(use_nesterov)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)
        self._dense_state = {}

    def step(self, gradients=None):
        self.global_step += 1
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        kw = dict(lr=self.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=self.eps,
                  beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                  grad_scale=self.reciprocal_scale, use_nesterov=self.use_nesterov)
        for i, p in enumerate(self.parameters):
            w, (m, v) = self._tables(p, (("moment1", 0.0), ("moment2", 0.0)))
            sg = getattr(p, "sparse_grads", None)
            if sg:
                for rg in sg:
                    if self.lazy:
                        ops.sparse_lazy_adam_(w, m, v, rg.plan, rg.values, rg.row_scale, **kw)
                    else:
                        dense = _densify(rg, w)
                        ops.dense_adam_(w, m, v, dense, **kw)
                p.sparse_grads = []
                continue
            g = gradients[i] if gradients is not None else p.grad
            if g is None:
                continue
            ops.dense_adam_(w, m, v, g.contiguous(), **kw)
        return True


class Adam(LazyAdam):
    """nn.Adam: every element moves every step; RowTensor gradients are densified first
    (UnsortedSegmentSum into [V, D], the Gather bprop) (wide_and_deep.py:435-437)."""
    lazy = False


def _densify(rg, like):
    sums = ops.segment_sum(rg.plan, rg.values, rg.row_scale)
    dense = torch.zeros_like(like)
    ops.scatter_unique_rows_(dense, rg.plan, sums)
    return dense


class FTRL(_Optimizer):
    """nn.FTRL(params, initial_accum=0.1, learning_rate=0.001, lr_power=-0.5, l1=0.0, l2=0.0,
    use_locking=False, loss_scale=1.0, weight_decay=0.0) (wide_and_deep.py:423-430)."""

    def __init__(self, params, initial_accum=0.1, learning_rate=0.001, lr_power=-0.5, l1=0.0, l2=0.0,
                 use_locking=False, loss_scale=1.0, weight_decay=0.0):
        super().__init__(params, learning_rate, loss_scale, weight_decay)
        if initial_accum < 0 or l1 < 0 or l2 < 0 or lr_power > 0 or learning_rate <= 0:
            raise ValueError("For 'FTRL', need initial_accum >= 0, l1 >= 0, l2 >= 0, lr_power <= 0, learning_rate > 0.")
        self.initial_accum, self.lr_power, self.l1, self.l2 = float(initial_accum), float(lr_power), float(l1), float(l2)
        self._dense_state = {}

    def step(self, gradients=None):
        self.global_step += 1
        kw = dict(lr=self.learning_rate, l1=self.l1, l2=self.l2, lr_power=self.lr_power, grad_scale=self.reciprocal_scale)
        for i, p in enumerate(self.parameters):
            w, (acc, lin) = self._tables(p, (("accum", self.initial_accum), ("linear", 0.0)))
            sg = getattr(p, "sparse_grads", None)
            if sg:
                for rg in sg:
                    ops.sparse_ftrl_(w, acc, lin, rg.plan, rg.values, rg.row_scale, **kw)
                p.sparse_grads = []
                continue
            g = gradients[i] if gradients is not None else p.grad
            if g is None:
                continue
            ops.dense_ftrl_(w, acc, lin, g.contiguous(), **kw)
        return True
