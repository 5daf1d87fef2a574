"""TEST INFRASTRUCTURE ONLY: a CPU kernel set for `compat/mindspore` built on the oracle's restatements (oracle/oracle.py,
tests/_oracle_ops.py).  The build container has no GPU; `tests/golden/make_ref_fixtures.py` installs this set
(`mindspore._kernels._install`) so that the REFERENCE's own Python (models/wide_deep/src/wide_and_deep.py,
models/deep_and_cross/src/deep_and_cross.py, mindspore_rec/) can run there over host tensors and record fixtures, and the CPU
tests drive the compat package's host logic with it.  The product never imports this module; its kernel set is
compat/mindspore/_hip_kernels.py (libmrec_hip.so), which has no CPU path."""
import ctypes as C

import numpy as np
import torch

import _oracle_ops as OO
from oracle import oracle as O

fill_normal_ = OO.fill_normal_
sparse_plan = OO.sparse_plan
dense_adam_ = OO.dense_adam_
dense_ftrl_ = OO.dense_ftrl_


def _np(t):
    return t.detach().as_subclass(torch.Tensor).numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def sparse_lazy_adam_(p, m, v, plan, g, row_scale=None, **kw):
    OO.sparse_lazy_adam_(p, m, v, plan, g.to(torch.float32).contiguous(), row_scale, **kw)


def sparse_ftrl_(var, accum, linear, plan, g, row_scale=None, **kw):
    OO.sparse_ftrl_(var, accum, linear, plan, g.to(torch.float32).contiguous(), row_scale, **kw)


def unique(x):
    y, inv = O.unique(_np(x))
    return torch.from_numpy(y), torch.from_numpy(inv.astype(np.int32))


def gather_rows(table, ids):
    t = np.ascontiguousarray(_np(table), dtype=np.float32)
    return torch.from_numpy(O.gather_rows(t, _np(ids).reshape(-1))).to(table.dtype)


def gather_bwd_dense(V, ids, g):
    """UnsortedSegmentSum(g, ids, V): positions added in ascending order per id (the CPU kernel's order)."""
    ids = _np(ids).reshape(-1)
    vals = np.ascontiguousarray(_np(g), dtype=np.float32).reshape(ids.size, -1)
    dense = np.zeros((int(V), vals.shape[1]), np.float32)
    if ids.size:
        u, inv = O.unique(ids)
        sums = O.segment_sum(vals, inv, u.size)
        ok = (u >= 0) & (u < V)
        dense[u[ok]] = sums[ok]
    return torch.from_numpy(dense)


def matmul(a, b, ta=False, tb=False):
    """fp32: float32 BLAS (summation order is the library's: inside the 1e-5 the fp32 paths are held to); 16-bit operands:
    exact products, fp32 accumulation, one rounding."""
    a2, b2 = a.detach().as_subclass(torch.Tensor), b.detach().as_subclass(torch.Tensor)
    a2 = a2.t() if ta else a2
    b2 = b2.t() if tb else b2
    if a.dtype == torch.float32:
        return a2 @ b2
    return (a2.to(torch.float32) @ b2.to(torch.float32)).to(a.dtype)


def dropout_mask(M, W, keep_prob, seed, step, layer, device):
    return torch.from_numpy((O.dropout_mask(int(M), int(W), seed, step, layer, keep_prob) != 0).astype(np.float32))


class MapStore:
    """oracle.Map (key -> row number, default rows keyed by (seed, key)) + numpy optimizer slots with the same row numbering."""

    def __init__(self, key_dtype, value_dtype, value_shape, default_value, permit_filter_value, evict_filter_value, name, device,
                 capacity=1 << 16, seed=0):
        assert device.type == "cpu"
        self.D, self.capacity, self.seed = int(value_shape[0]), int(capacity), int(seed or 0)
        self.key_dtype = key_dtype
        if isinstance(default_value, str):
            sigma, fill = (0.01, None) if default_value == "normal" else (None, {"zeros": 0.0, "ones": 1.0}[default_value])
        else:
            sigma, fill = None, float(default_value)
        self.map = O.Map(self.D, self.capacity, seed=self.seed, sigma=sigma if sigma is not None else 0.01, fill=fill)
        self.sigma, self.fill = sigma, fill
        self.permit, self.evict_after = int(permit_filter_value), int(evict_filter_value)
        self.slots = {}
        self.hits = np.zeros(self.capacity, np.int64)
        self.last = np.zeros(self.capacity, np.int64)
        self.step = 0

    def _rows_view(self):
        pr = O.lib().mrec_o_map_rows_ptr
        pr.restype = C.POINTER(C.c_float)
        return np.ctypeslib.as_array(pr(self.map._h), shape=(self.capacity, self.D))

    def get(self, keys, insert):
        return torch.from_numpy(self.map.get(_np(keys).astype(np.int64), bool(insert)))

    def put(self, keys, vals):
        self.map.put(_np(keys).astype(np.int64), _np(vals))

    def erase(self, keys):
        self.map.erase(_np(keys).astype(np.int64))

    def size(self):
        return self.map.size()

    def export(self):
        k, v = self.map.export()
        return torch.from_numpy(k).to(self.key_dtype), torch.from_numpy(v)

    def export_data(self, incremental):
        if incremental:
            raise NotImplementedError("incremental export: the device store's job (tests/test_map_gpu.py)")
        k, v = self.export()
        return k, v, torch.zeros(k.numel(), dtype=torch.int32)

    def import_data(self, data):
        self.put(data[0], data[1])

    def clear(self):
        k, _ = self.map.export()
        if k.size:
            self.map.erase(k)

    def evict(self):
        return 0

    def _slot(self, name, init):
        if name not in self.slots:
            self.slots[name] = np.full((self.capacity, self.D), np.float32(init), np.float32)
            pend = self.__dict__.setdefault("_pending", {}).pop(name, None)
            if pend is not None:                   # restored before the optimizer created the slot: only it knows the untouched rows' value
                self.slots[name][pend[0]] = pend[1]
        return self.slots[name]

    def export_slots(self):
        k, _ = self.map.export()
        rows = self.map.find_or_insert(k, False)
        out = {n: t[rows].copy() for n, t in self.slots.items()}
        for n, (r, v) in self.__dict__.get("_pending", {}).items():
            t = np.zeros((self.capacity, self.D), np.float32)
            t[r] = v
            out[n] = t[rows].copy()
        return out

    def import_slots(self, keys, slots):
        rows = self.map.find_or_insert(_np(keys).astype(np.int64), True)
        for n, vals in slots.items():
            v = _np(vals).astype(np.float32).reshape(len(rows), self.D)
            if n in self.slots:
                self.slots[n][rows] = v
            else:
                self.__dict__.setdefault("_pending", {})[n] = (rows.copy(), v.copy())

    def _rows(self, keys):
        """Row numbers for an apply: one training step of the table; rows of keys not yet admitted -> -1 (skipped)."""
        k = _np(keys).astype(np.int64).reshape(-1)
        rows = self.map.find_or_insert(k, True).astype(np.int64)
        self.step += 1
        u = np.unique(rows)
        self.hits[u] += 1
        self.last[u] = self.step
        if self.permit > 1:
            rows = np.where(self.hits[rows] >= self.permit, rows, -1)
        return rows

    def apply_lazy_adam(self, keys, g, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale, use_nesterov):
        rows = self._rows(keys)
        O.sparse_lazy_adam(self._rows_view(), self._slot("moment1", 0.0), self._slot("moment2", 0.0), rows,
                           np.ascontiguousarray(_np(g), np.float32).reshape(rows.size, self.D), None, lr=lr, b1=beta1, b2=beta2, eps=eps,
                           b1_pow=beta1_power, b2_pow=beta2_power, grad_scale=grad_scale, nesterov=use_nesterov)

    def apply_ftrl(self, keys, g, initial_accum, lr, l1, l2, lr_power, grad_scale):
        rows = self._rows(keys)
        O.sparse_ftrl(self._rows_view(), self._slot("accum", initial_accum), self._slot("linear", 0.0), rows,
                      np.ascontiguousarray(_np(g), np.float32).reshape(rows.size, self.D), None, lr=lr, l1=l1, l2=l2, lr_power=lr_power,
                      grad_scale=grad_scale)
