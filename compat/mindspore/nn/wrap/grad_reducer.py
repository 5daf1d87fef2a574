"""`mindspore.nn.wrap.grad_reducer.DistributedGradReducer` (wide_and_deep.py:458-470,487-489): sum over
the data-parallel ranks with RCCL (`torch.distributed`, backend "nccl"), divided by `degree` when `mean`.  Dense
gradients are flattened into ONE buffer per call -- xGMI ring collectives are per-link bound, so few large messages;
RowTensor gradients are all-gathered (indices and values), the optimizer's dedup adds the ranks' rows."""
import torch
import torch.distributed as dist

from ...common.sparse_tensor import MapTensorGrad, RowTensor
from ...common.tensor import as_tensor
from ..cell import Cell


def _staged(t):
    """A gloo group handed device tensors (ranks sharing one GPU): gloo moves host memory."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(t):
    if _staged(t):
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    else:
        dist.all_reduce(t)


def _all_gather(t, n):
    if _staged(t):
        c = t.cpu()
        outs = [torch.empty_like(c) for _ in range(n)]
        dist.all_gather(outs, c)
        return [o.to(t.device) for o in outs]
    outs = [torch.empty_like(t) for _ in range(n)]
    dist.all_gather(outs, t)
    return outs


class DistributedGradReducer(Cell):
    def __init__(self, parameters, mean=None, degree=None, fusion_type=1, group=None):
        super().__init__(auto_prefix=False)
        from ... import context
        self.mean = context.get_auto_parallel_context("gradients_mean") if mean is None else bool(mean)
        if degree is None:
            degree = dist.get_world_size() if dist.is_initialized() else 1
        if not isinstance(degree, int) or degree <= 0:
            raise ValueError(f"For 'DistributedGradReducer', the 'degree' must be a positive int, but got {degree!r}.")
        self.degree = degree
        self.__dict__["_parameters"] = tuple(parameters)

    def _gather_rows(self, idx, vals):
        n = dist.get_world_size()
        cnt = torch.tensor([idx.numel()], device=idx.device, dtype=torch.int64)
        cnts = _all_gather(cnt, n)
        cap = int(max(int(c) for c in cnts))
        pi = torch.full((cap,), -1, dtype=idx.dtype, device=idx.device)
        pv = torch.zeros((cap,) + tuple(vals.shape[1:]), dtype=vals.dtype, device=vals.device)
        pi[: idx.numel()], pv[: idx.numel()] = idx, vals
        gi, gv = _all_gather(pi, n), _all_gather(pv, n)
        keep = [slice(0, int(c)) for c in cnts]
        return torch.cat([a[s] for a, s in zip(gi, keep)]), torch.cat([a[s] for a, s in zip(gv, keep)])

    def construct(self, grads):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return grads
        scale = 1.0 / self.degree if self.mean else 1.0
        dense = [(i, g.as_subclass(torch.Tensor)) for i, g in enumerate(grads) if isinstance(g, torch.Tensor)]
        out = list(grads)
        if dense:
            flat = torch.cat([g.reshape(-1).to(torch.float32) for _, g in dense])
            _all_reduce(flat)
            if self.mean:
                flat *= scale
            off = 0
            for i, g in dense:
                out[i] = as_tensor(flat[off: off + g.numel()].reshape(g.shape).to(g.dtype))
                off += g.numel()
        for i, g in enumerate(grads):
            if isinstance(g, RowTensor):
                idx, vals = self._gather_rows(g.indices, g.values * scale if self.mean else g.values)
                out[i] = RowTensor(idx, vals, g.dense_shape)
            elif isinstance(g, MapTensorGrad):
                k, vals = self._gather_rows(g.keys, g.values * scale if self.mean else g.values)
                out[i] = MapTensorGrad(k, vals)
        return tuple(out)
