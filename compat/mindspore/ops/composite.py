"""`mindspore.ops.composite` (imported as `C`): GradOperation, HyperMap, tensor_dot
(models/wide_deep/src/wide_and_deep.py:446-448,481-486; models/deep_and_cross/src/deep_and_cross.py:145,345-346)."""
import torch

from ..common.sparse_tensor import MapTensorGrad, RowTensor
from ..common.tensor import Tensor, as_tensor


class HyperMap:
    """hyper_map(fn, *sequences): fn over the zipped leaves of (nested) tuples / lists."""

    def __init__(self, ops=None, reverse=False):
        self.ops = ops

    def __call__(self, *args):
        fn, seqs = (self.ops, args) if self.ops is not None else (args[0], args[1:])
        return self._map(fn, seqs)

    def _map(self, fn, seqs):
        if isinstance(seqs[0], (tuple, list)):
            n = len(seqs[0])
            if any(len(s) != n for s in seqs):
                raise ValueError("For 'HyperMap', all sequences must have the same length.")
            return tuple(self._map(fn, [s[i] for s in seqs]) for i in range(n))
        return fn(*seqs)


Map = HyperMap


class GradOperation:
    """GradOperation(get_all=False, get_by_list=False, sens_param=False).

    `grad_op(net, weights)(*inputs[, sens])` runs `net` once under torch's tape and differentiates it: dense gradients for
    ordinary Parameters, a RowTensor for a table read through a sparse lookup, a MapTensorGrad for a MapParameter.  With
    sens_param the last positional argument seeds the backward (the loss scale of TrainStepWrap)."""

    def __init__(self, get_all=False, get_by_list=False, sens_param=False):
        self.get_all, self.get_by_list, self.sens_param = bool(get_all), bool(get_by_list), bool(sens_param)

    def __call__(self, fn, weights=None):
        if self.get_by_list and weights is None:
            raise ValueError("For 'GradOperation', 'weights' must be given when get_by_list=True.")

        def grad_fn(*args):
            from ..common.parameter import Parameter
            from ..experimental import MapParameter
            args = list(args)
            sens = args.pop() if self.sens_param else None
            ws = list(weights) if self.get_by_list else []
            want_inputs = self.get_all or not self.get_by_list
            with torch.enable_grad():
                ins, call_args = [], []
                for a in args:
                    if want_inputs and isinstance(a, torch.Tensor) and a.is_floating_point():
                        a = as_tensor(a.detach().as_subclass(torch.Tensor).requires_grad_(True))
                        ins.append(a)
                    call_args.append(a)
                for w in ws:
                    w._row_grads = []
                out = fn(*call_args)
                outs = list(out) if isinstance(out, (tuple, list)) else [out]
                if sens is None:
                    seeds = [torch.ones_like(o) for o in outs]
                else:
                    sl = list(sens) if isinstance(sens, (tuple, list)) else [sens]
                    seeds = [s.as_subclass(torch.Tensor).to(o.dtype).reshape(o.shape) if isinstance(s, torch.Tensor)
                             else torch.full_like(o, float(s)) for s, o in zip(sl, outs)]
                leaves, slots = [], []
                for i, w in enumerate(ws):
                    if isinstance(w, Parameter) and w.requires_grad:
                        leaves.append(w)
                        slots.append(("dense", i))
                    hook = getattr(w, "_row_hook", None)
                    if hook is not None:
                        leaves.append(hook)
                        slots.append(("hook", i))
                n_w = len(leaves)
                leaves += ins
                live = [(o, s) for o, s in zip(outs, seeds) if o.requires_grad]
                if live and leaves:
                    gs = torch.autograd.grad([o for o, _ in live], leaves, [s for _, s in live], allow_unused=True)
                else:
                    gs = [None] * len(leaves)
            dense = {i: g for (kind, i), g in zip(slots, gs[:n_w]) if kind == "dense"}
            wgrads = []
            for i, w in enumerate(ws):
                rows, w._row_grads = w._row_grads, []
                if isinstance(w, MapParameter):
                    if rows:
                        wgrads.append(MapTensorGrad(torch.cat([k for k, _ in rows]), torch.cat([v for _, v in rows])))
                    else:
                        wgrads.append(MapTensorGrad(torch.empty(0, dtype=w.key_dtype, device=w.device),
                                                    torch.empty((0,) + w.value_shape, device=w.device)))
                    continue
                g = dense.get(i)
                if rows:
                    rt = RowTensor(torch.cat([k for k, _ in rows]), torch.cat([v for _, v in rows]), w.shape)
                    if g is not None and bool((g != 0).any()):
                        # the same table also took a dense gradient (e.g. an L2 term): hand the optimizer the dense sum
                        from .._kernels import K
                        g = g.as_subclass(torch.Tensor) + K().gather_bwd_dense(w.shape[0], rt.indices, rt.values)
                        wgrads.append(as_tensor(g))
                    else:
                        wgrads.append(rt)
                else:
                    wgrads.append(as_tensor(g) if g is not None else Tensor(torch.zeros_like(w.as_subclass(torch.Tensor))))
            igrads = tuple(as_tensor(g) if g is not None else None for g in gs[n_w:])
            if self.get_by_list and self.get_all:
                return igrads, tuple(wgrads)
            if self.get_by_list:
                return tuple(wgrads)
            if self.get_all:
                return igrads
            return igrads[0] if igrads else None

        return grad_fn


def tensor_dot(x1, x2, axes):
    """tensor_dot(x1, x2, axes): contraction over the given axes -- CrossLayer's x_l^T w, a reduction over D
    (deep_and_cross.py:145): a GEMV, HBM-bound."""
    if isinstance(axes, int):
        return as_tensor(torch.tensordot(x1, x2, dims=axes))
    a, b = axes
    a = [a] if isinstance(a, int) else list(a)
    b = [b] if isinstance(b, int) else list(b)
    return as_tensor(torch.tensordot(x1, x2, dims=(a, b)))


def clip_by_value(x, clip_value_min=None, clip_value_max=None):
    return as_tensor(torch.clamp(x, min=clip_value_min, max=clip_value_max))
