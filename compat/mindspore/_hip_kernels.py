"""The product kernel set: every function is a thin adapter onto `mindrec_amd.ops` / `mindrec_amd.experimental`
(ctypes calls into libmrec_hip.so through the C ABI of include/mrec.h).  Device tensors only -- `mindrec_amd.ops`
refuses host tensors and raises ImportError when the HIP library is missing; there is no fallback below this file."""
import torch

from mindrec_amd import ops
from mindrec_amd.experimental import MapParameter as _DeviceMap

# ---- straight re-exports (same names and signatures as the engines use) ------------------------------------------------
fill_normal_ = ops.fill_normal_
sparse_plan = ops.sparse_plan
sparse_lazy_adam_ = ops.sparse_lazy_adam_
sparse_ftrl_ = ops.sparse_ftrl_
dense_adam_ = ops.dense_adam_
dense_ftrl_ = ops.dense_ftrl_


def unique(x):
    """ops.Unique: (y [U], idx int32 [n]), y in first-occurrence order.  The ONE host read of U is what gives `y` its
    data-dependent shape (the engines keep U on the device; a Python-visible Unique cannot)."""
    d = ops.unique(x)
    return d.uniq, d.inv


def gather_rows(table, ids):
    """out[i] = table[ids[i]], zeros for ids outside the table; `table` may be a strided [V, D] view."""
    if table.dim() != 2:
        raise TypeError("gather_rows: a [V, D] table")
    if table.stride(1) != 1:
        table = table.contiguous()
    if table.dtype != torch.float32:
        return ops.gather_rows(table.to(torch.float32), ids).to(table.dtype)
    return ops.gather_rows(table, ids)


def gather_bwd_dense(V, ids, g):
    """UnsortedSegmentSum(g, ids, V): the dense [V, D] gradient of Gather -- segment sums of the sorted-by-id positions (fixed
    order, no float atomics), scattered to their rows."""
    g = g.to(torch.float32).contiguous()
    dense = torch.zeros((int(V), g.shape[1]), dtype=torch.float32, device=g.device)
    if ids.numel() == 0:
        return dense
    plan = ops.sparse_plan(ids)
    ops.scatter_unique_rows_(dense, plan, ops.segment_sum(plan, g))
    return dense


def _mm32(a, b, ta, tb):
    if not ta and not tb:
        return ops.dense32_fwd(a, b, None, relu=False)
    if not ta and tb:
        return ops.dense32_bwd_input(a, b)                      # a [M, N] . b[K, N]^T
    if ta and not tb:
        M, K = a.shape
        N = b.shape[1]
        S = ops.dense32_bwd_weight_slabs(M, K, N)
        slabs = torch.empty((S, K, N), dtype=torch.float32, device=a.device)
        ops.dense32_bwd_weight(a, b, slabs)                     # a[M, K]^T . b [M, N]
        if S == 1:
            return slabs[0]
        return ops.sum_slabs(slabs, torch.empty((K, N), dtype=torch.float32, device=a.device))
    return _mm32(b, a, False, False).t().contiguous()


def matmul(a, b, ta=False, tb=False):
    """op(a) . op(b) on the matrix cores: fp32 on v_mfma_f32_32x32x2_f32 (exact fp32, csrc/mrec_gemm_f32.hip); fp16 / bf16
    operands are widened (their products are exact in fp32), accumulated in fp32 and rounded once -- what a 16-bit MFMA with
    an fp32 accumulator returns."""
    if a.dtype != b.dtype:
        raise TypeError(f"matmul: operand dtypes differ: {a.dtype} and {b.dtype}")
    if a.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError(f"matmul: unsupported dtype {a.dtype}")
    a2 = a if a.stride(-1) == 1 else a.contiguous()
    b2 = b if b.stride(-1) == 1 else b.contiguous()
    if a.dtype != torch.float32:
        return _mm32(a2.to(torch.float32), b2.to(torch.float32), ta, tb).to(a.dtype)
    return _mm32(a2, b2, ta, tb)


def dropout_mask(M, W, keep_prob, seed, step, layer, device):
    """{0, 1} mask [M, W] (fp32) of the counter-based Dropout (csrc/mrec_dropout.h)."""
    d = ops.Dropout(keep_prob, seed, layer, step=step)
    return (ops.dropout_mask(int(M), int(W), d, device) != 0).to(torch.float32)


class MapStore:
    """Storage + optimizer applies of one `mindspore.experimental.MapParameter`: a device key index over dense row tables
    (mindrec_amd/experimental.py, csrc/mrec_hash.hip).  A training step of the table is counted where the optimizer applies
    its gradient (admission / eviction thresholds are in training steps, README.md:182-183)."""

    def __init__(self, key_dtype, value_dtype, value_shape, default_value, permit_filter_value, evict_filter_value, name, device,
                 capacity=1 << 20, seed=None):
        self.m = _DeviceMap(key_dtype=key_dtype, value_dtype=value_dtype, value_shape=value_shape, default_value=default_value,
                            permit_filter_value=permit_filter_value, evict_filter_value=evict_filter_value, name=name,
                            capacity=capacity, device=device, seed=seed)

    def get(self, keys, insert):
        m = self.m
        _, _, rows = m.lookup_rows(keys, insert=insert, train=False)
        out = ops.gather_rows(m.values, rows)
        if not insert:
            m.index.fill_missing(keys, rows, out, m._sigma, m._fill, m.seed)
        return out

    def put(self, keys, vals):
        self.m.put(keys, vals)

    def erase(self, keys):
        if keys.numel():
            self.m.erase(keys)

    def size(self):
        return len(self.m)

    def export(self):
        return self.m.get_data()

    def export_data(self, incremental):
        return self.m.export_data(incremental)

    def import_data(self, data):
        self.m.import_data(data)

    def evict(self):
        return self.m.evict()

    def clear(self):
        k = self.m.get_keys()
        if k.numel():
            self.m.erase(k)

    def export_slots(self):
        k, r = self.m.index.export()
        return {n: v.cpu().numpy() for n, v in self.m.slot_rows(r).items()}

    def import_slots(self, keys, slots):
        _, _, rows = self.m.lookup_rows(keys.reshape(-1), insert=True, train=False)
        for n, vals in slots.items():
            self.m.import_slot(n, rows, vals)

    def _plan(self, keys):
        m = self.m
        d = ops.unique(keys)
        _, rows_u, _ = m.lookup_rows(keys, insert=True, dedup=d, train=True)
        plan = ops.group_by_inverse(d)
        plan.uniq_buf = m.admitted_rows(rows_u)          # groups -> table rows; keys not yet admitted -> -1 (skipped)
        return plan

    def apply_lazy_adam(self, keys, g, **kw):
        m = self.m
        mo, ve = m.add_slot("moment1", 0.0), m.add_slot("moment2", 0.0)
        ops.sparse_lazy_adam_(m.values, mo, ve, self._plan(keys), g.to(torch.float32), None, **kw)

    def apply_ftrl(self, keys, g, initial_accum=0.1, **kw):
        m = self.m
        acc, lin = m.add_slot("accum", initial_accum), m.add_slot("linear", 0.0)
        ops.sparse_ftrl_(m.values, acc, lin, self._plan(keys), g.to(torch.float32), None, **kw)
