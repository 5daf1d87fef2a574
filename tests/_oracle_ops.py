"""CPU stand-in for mindrec_amd.ops built on the oracle.  TEST INFRASTRUCTURE ONLY: it exists so
the multi-rank host logic of WideDeepEngine (routing, all-to-all protocol, gradient averaging) can
run under gloo on a machine without a GPU, and so the GPU engine can be checked step for step.
The product never imports this module."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    return t.detach().numpy()


def fill_normal_(table, seed, sigma=0.01, row0=0, row_stride=1):
    V, D = table.shape
    rows = row0 + row_stride * np.arange(V, dtype=np.int64)
    _np(table)[...] = O.normal_rows(seed, rows, D, sigma)
    return table


class Plan:
    def __init__(self, ids):
        self.ids = _np(ids).reshape(-1).copy()
        self.n = self.ids.size


def sparse_plan(ids, skip_negative=False):
    """(negative ids need no special care here: the oracle's row updates skip rows outside the table)"""
    return Plan(ids)


def head_supported(K5):
    return False


def gather_rows(table, ids, row_scale=None):
    out = O.gather_rows(_np(table), _np(ids), _np(row_scale) if row_scale is not None else None)
    return torch.from_numpy(out)


def wide_sum(w, ids, wts, bias=None):
    return torch.from_numpy(O.wide_sum(_np(w), _np(ids), _np(wts), float(bias[0]) if bias is not None else 0.0))


def sparse_lazy_adam_(p, m, v, plan, g, row_scale=None, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9,
                      beta2_power=0.999, grad_scale=1.0, use_nesterov=False):
    O.sparse_lazy_adam(_np(p), _np(m), _np(v), plan.ids, _np(g).reshape(plan.n, -1),
                       _np(row_scale).reshape(-1) if row_scale is not None else None, lr=lr, b1=beta1, b2=beta2, eps=eps,
                       b1_pow=beta1_power, b2_pow=beta2_power, grad_scale=grad_scale, nesterov=use_nesterov)


def sparse_ftrl_(var, accum, linear, plan, g, row_scale=None, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5, grad_scale=1.0):
    O.sparse_ftrl(_np(var), _np(accum), _np(linear), plan.ids, _np(g).reshape(plan.n, -1),
                  _np(row_scale).reshape(-1) if row_scale is not None else None, lr=lr, l1=l1, l2=l2, lr_power=lr_power,
                  grad_scale=grad_scale)


def dense_adam_(p, m, v, g, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9, beta2_power=0.999,
                grad_scale=1.0, use_nesterov=False, ftrl1=None):
    """ftrl1 = (index, lr, l1, l2, lr_power): that element belongs to nn.FTRL (m word = accum, v word = linear)."""
    if ftrl1 is not None:
        i = int(ftrl1[0])
        keep = [a.reshape(-1)[i:i + 1].copy() for a in (_np(p), _np(m), _np(v))]
    O.dense_adam(_np(p), _np(m), _np(v), _np(g), lr=lr, b1=beta1, b2=beta2, eps=eps, b1_pow=beta1_power,
                 b2_pow=beta2_power, grad_scale=grad_scale, nesterov=use_nesterov)
    if ftrl1 is not None:
        O.dense_ftrl(keep[0], keep[1], keep[2], _np(g).reshape(-1)[i:i + 1].copy(), lr=ftrl1[1], l1=ftrl1[2], l2=ftrl1[3],
                     lr_power=ftrl1[4], grad_scale=grad_scale)
        for a, k in zip((_np(p), _np(m), _np(v)), keep):
            a.reshape(-1)[i] = k[0]


def dense_adam_l2_(p, m, v, g, l2_scaled, sumsq=None, accumulate=False, **kw):
    """g + fl(l2_scaled * p) (product rounded, then added), sum(p^2) of the old values in float64."""
    pp = _np(p)
    if sumsq is not None:
        ss = float((pp.astype(np.float64) ** 2).sum())
        sumsq[0] = (float(sumsq[0]) if accumulate else 0.0) + ss
    gg = (_np(g) + (pp * np.float32(l2_scaled)).astype(np.float32)).astype(np.float32)
    dense_adam_(p, m, v, torch.from_numpy(gg), **kw)


def dense_adam_rows_l2_(p, m, v, plan, sums, l2_scaled=0.0, sumsq=None, accumulate=False, step_state=None, **kw):
    """The restatement: the [V, D] gradient of the Gather -- zeros, the group sums scattered to their rows -- then dense_adam_l2_."""
    assert step_state is None
    g = torch.zeros_like(p)
    scatter_unique_rows_(g, plan, sums)
    dense_adam_l2_(p, m, v, g, l2_scaled, sumsq=sumsq, accumulate=accumulate, **kw)


def dense_ftrl_(var, accum, linear, g, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5, grad_scale=1.0):
    O.dense_ftrl(_np(var), _np(accum), _np(linear), _np(g), lr=lr, l1=l1, l2=l2, lr_power=lr_power, grad_scale=grad_scale)


# ---- fixed-capacity routing of a sharded step (oracle/oracle.py restatements; fp32 rows) ------------------------------------
def shard_capacity(n, n_shards, factor=1.25):
    return O.shard_capacity(n, n_shards, factor)


def shard_msg_words(D, act_dtype):
    assert act_dtype == torch.float32
    return D, D + 4


def shard_route_slots(ids, wts, n_shards, cap, hashed=False, overflow=None, rot=0, out=None):
    rid, rw, sop, pos, dropped = O.shard_route_slots(_np(ids), _np(wts) if wts is not None else None, n_shards, cap, hashed, rot)
    idt = _np(ids).dtype
    req = np.empty((n_shards * cap, 2), idt) if out is None else _np(out)
    req[:, 0] = rid.astype(idt)
    req[:, 1] = rw.view(np.int32).astype(idt)                       # the weight's bits in the entry's second word
    if overflow is not None:
        overflow += dropped
    return (torch.from_numpy(req) if out is None else out), torch.from_numpy(sop), torch.from_numpy(pos)


def shard_unpack_req(req):
    r = _np(req)
    return torch.from_numpy(r[:, 0].copy()), torch.from_numpy(r[:, 1].astype(np.int32).view(np.float32).copy())


def shard_unroute_slots(back, slot_of_pos, D, act_dtype, out=None):
    b, s = _np(back), _np(slot_of_pos)
    ok = s >= 0
    emb = np.zeros((s.size, D), np.float32)
    wp = np.zeros((s.size, 2), np.float32)
    emb[ok] = b[s[ok], :D]
    wp[ok, 0] = b[s[ok], D]
    return torch.from_numpy(emb), torch.from_numpy(wp)


def shard_route_grads(g, dlogit, F, pos_of_slot, out=None):
    gg, dl, p = _np(g), _np(dlogit), _np(pos_of_slot)
    D = gg.shape[1]
    if out is not None:
        msg = _np(out)
        msg[...] = 0
    else:
        msg = np.zeros((p.size, D + 4), np.float32)
    ok = p >= 0
    msg[ok, :D] = gg[p[ok]]
    msg[ok, D] = dl[p[ok] // F]
    return torch.from_numpy(msg) if out is None else out


def segment_sum(plan, g, row_scale=None, grad_scale=1.0):
    u, inv = O.unique(plan.ids)
    vals = _np(g).reshape(plan.n, -1)
    if row_scale is not None:
        vals = vals * _np(row_scale).reshape(-1, 1)
    vals = (vals * np.float32(grad_scale)).astype(np.float32)
    plan._uniq = u
    out = np.zeros((plan.n, vals.shape[1]), np.float32)
    out[: u.size] = O.segment_sum(vals, inv, u.size)
    return torch.from_numpy(out)


def scatter_unique_rows_(table, plan, vals):
    u = plan._uniq
    ok = (u >= 0) & (u < table.shape[0])
    _np(table)[u[ok]] = _np(vals)[: u.size][ok]


def cross_layers(x0, w, b):
    return torch.from_numpy(O.cross_layers(_np(x0), _np(w), _np(b)))


def cross_layers_bwd(x0, w, b, dy):
    return tuple(torch.from_numpy(a) for a in O.cross_layers_bwd(_np(x0), _np(w), _np(b), _np(dy)))


def fm_forward(vx):
    fm, cs = O.fm_forward(_np(vx))
    return torch.from_numpy(fm.astype(np.float32)), torch.from_numpy(cs)


def fm_backward_(g, vx, colsum, dout):
    _np(g)[...] = _np(g) + O.fm_backward(_np(vx), _np(colsum), _np(dout)).astype(np.float32)
    return g


def scatter_unique_rows_add_(table, plan, vals):
    u = plan._uniq
    ok = (u >= 0) & (u < table.shape[0])
    _np(table)[u[ok]] += _np(vals)[: u.size][ok]


class Dropout:
    """ops.Dropout for the CPU stand-in: the descriptor only (the step always by value)."""

    def __init__(self, keep_prob, seed, layer, step=0, row0=0, step_state=None):
        assert step_state is None
        self.keep_prob, self.seed, self.layer, self.step, self.row0 = float(keep_prob), int(seed), int(layer), int(step), int(row0)

    @property
    def scale(self):
        return float(np.float32(1.0) / np.float32(self.keep_prob))


def dropout_mask(M, W, drop, device):
    return torch.from_numpy(O.dropout_mask(M, W, drop.seed, drop.step, drop.layer, drop.keep_prob, drop.row0))
