"""Backward rules of the primitives that run on the kernel set (`torch.autograd.Function`s: torch keeps the tape, the
arithmetic on both sides of it is the kernel set's).  Bprop definitions follow SURVEY Appendix A.2 / A.6 [EXT]:

  Gather(params, idx, 0)            -> UnsortedSegmentSum(dout, idx, V): a dense [V, D] gradient
  SparseGatherV2 / EmbeddingLookup  -> RowTensor(indices = idx, values = dout), not deduplicated
  MapTensorGet                      -> MapTensor-typed gradient (keys, value gradients)
  MatMul                            -> dx = dy . w^T, dw = x^T . dy
"""
import torch

from .._kernels import K


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t


class GatherDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids):
        ctx.V, ctx.ids, ctx.needs = table.shape[0], ids, table.requires_grad
        return K().gather_rows(_plain(table).detach(), _plain(ids).reshape(-1))

    @staticmethod
    def backward(ctx, g):
        return K().gather_bwd_dense(ctx.V, _plain(ctx.ids).reshape(-1), _plain(g).contiguous()), None


class GatherRowsSparse(torch.autograd.Function):
    """out = table[ids]; the backward files (ids, dout) on the Parameter instead of building [V, D]."""

    @staticmethod
    def forward(ctx, hook, param, ids):
        ctx.param, ctx.ids = param, ids
        return K().gather_rows(_plain(param).detach(), _plain(ids).reshape(-1))

    @staticmethod
    def backward(ctx, g):
        ctx.param._row_grads.append((_plain(ctx.ids).reshape(-1), _plain(g).contiguous()))
        return torch.zeros((), device=g.device), None, None


class MapGet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, mp, keys, insert):
        ctx.mp, ctx.keys = mp, keys
        return mp._store.get(_plain(keys).reshape(-1), bool(insert))

    @staticmethod
    def backward(ctx, g):
        ctx.mp._row_grads.append((_plain(ctx.keys).reshape(-1), _plain(g).contiguous()))
        return torch.zeros((), device=g.device), None, None, None


class MatMul2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, ta, tb):
        ctx.save_for_backward(a, b)
        ctx.ta, ctx.tb = ta, tb
        return K().matmul(_plain(a).detach(), _plain(b).detach(), ta, tb)

    @staticmethod
    def backward(ctx, g):
        a, b = (_plain(t).detach() for t in ctx.saved_tensors)
        g = _plain(g).contiguous()
        ta, tb = ctx.ta, ctx.tb
        da = db = None
        mm = K().matmul
        if ctx.needs_input_grad[0]:
            # c = op(a) . op(b):  d op(a) = g . op(b)^T
            da = mm(b, g, tb, True) if ta else mm(g, b, False, not tb)
        if ctx.needs_input_grad[1]:
            db = mm(g, a, True, ta) if tb else mm(a, g, not ta, False)
        return da, db, None, None
