"""Host-side wrappers: torch tensors in, libmrec_hip.so kernels on the current HIP stream.

Each function names the MindSpore primitive it stands in for and the reference call site
(paths relative to the mindspore-lab/mindrec checkout).  torch is the device-memory container
only: no arithmetic on the hot path is done by torch here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_WS = {}
_DT16 = {torch.bfloat16: "bf16", torch.float16: "f16"}
_WS_RETIRED = []      # outgrown workspaces stay referenced: a captured HIP graph may still hold their addresses


def _stream():
    """The current stream's raw handle.  Through torch's C entry points: torch.cuda.current_stream() builds a Python Stream object
    and re-checks device availability (an os.environ lookup) on every call -- tens of microseconds of host time per kernel
    launch, which a step issued kernel by kernel (a shard's: ~30 launches) cannot afford."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_PLAN_PRIMED = {}
_MAP_PRIMED = {}


def workspace(tag, nbytes, device):
    """Stream-ordered scratch, cached per (tag, device, stream) and grown geometrically."""
    key = (tag, str(device), torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _WS_RETIRED.append(t)
        t = torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mindrec_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")


def _suffix(ids):
    if ids.dtype == torch.int32:
        return "i32"
    if ids.dtype == torch.int64:
        return "i64"
    raise TypeError(f"ids must be int32 or int64, got {ids.dtype}")


def _table(t):
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError("table must be a float32 [V, D] tensor with unit column stride")
    return t.shape[0], t.shape[1], t.stride(0)


def fill_normal_(table, seed, sigma=0.01, row0=0, row_stride=1):
    """initializer('normal') written on the device (default_config.yaml:41; embedding.py:88).
    Row r of `table` gets the values of global row row0 + r*row_stride."""
    _need_cuda(table)
    V, D, ld = _table(table)
    _lib.call("mrec_fill_normal_f32", _ptr(table), V, D, ld, C.c_uint64(seed), row0, row_stride, sigma, _stream())
    return table


class Dedup:
    """Result of ops.Unique (embedding.py:153,192): uniq[:U] in first-occurrence order, inv[n]."""

    def __init__(self, ids_flat, uniq, inv, n_uniq_dev):
        self.ids = ids_flat
        self.uniq_buf = uniq
        self.inv = inv
        self.n_uniq_dev = n_uniq_dev
        self._U = None

    @property
    def n(self):
        return self.ids.numel()

    @property
    def U(self):
        """Number of unique ids (host sync on first use)."""
        if self._U is None:
            self._U = int(self.n_uniq_dev.item())
        return self._U

    @property
    def uniq(self):
        return self.uniq_buf[: self.U]


def unique(ids):
    _need_cuda(ids)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    dev = flat.device
    uniq = torch.empty(max(n, 1), dtype=flat.dtype, device=dev)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    n_uniq = torch.empty(1, dtype=torch.int64, device=dev)
    nb = _lib.query_bytes("mrec_dedup_workspace_bytes", n)
    ws = workspace("dedup", nb, dev)
    _lib.call(f"mrec_dedup_{sfx}", _ptr(flat), n, _ptr(uniq), _ptr(inv), _ptr(n_uniq), _ptr(ws), ws.numel(), _stream())
    return Dedup(flat, uniq, inv, n_uniq)


class SparsePlan(Dedup):
    """Dedup + inverted index of one step's ids: what the optimizer-side RowTensor dedup needs."""

    def __init__(self, d, sorted_pos, sorted_seg, seg_offsets, n_valid_dev=None):
        super().__init__(d.ids, d.uniq_buf, d.inv, d.n_uniq_dev)
        self._U = d._U
        self.sorted_pos = sorted_pos
        self.sorted_seg = sorted_seg
        self.seg_offsets = seg_offsets
        self.n_valid_dev = n_valid_dev      # device word: entries of the index proper (sparse_plan(skip_negative=True)); None: all n


def group_by_inverse(d):
    n = d.n
    dev = d.ids.device
    sorted_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    sorted_seg = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    seg_offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nb = _lib.query_bytes("mrec_group_workspace_bytes", n)
    ws = workspace("group", nb, dev)
    _lib.call("mrec_group_by_inverse", _ptr(d.inv), n, _ptr(sorted_pos), _ptr(sorted_seg), _ptr(seg_offsets), _ptr(ws),
              ws.numel(), _stream())
    return SparsePlan(d, sorted_pos, sorted_seg, seg_offsets)


def sparse_plan(ids, skip_negative=False):
    """Unique + inverted index of a step's ids in one library call (mrec_sparse_plan_*).  skip_negative: negative ids are
    padding (the unused slots of a shard's fixed-capacity request message) -- no group, no entry of the index proper, whose
    length then lives in plan.n_valid_dev."""
    _need_cuda(ids)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    dev = flat.device
    uniq = torch.empty(max(n, 1), dtype=flat.dtype, device=dev)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    n_uniq2 = torch.empty(2, dtype=torch.int64, device=dev)          # [U, entries of the index proper]
    n_uniq = n_uniq2[:1]
    sorted_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    sorted_seg = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    seg_offsets = torch.empty(n + 2, dtype=torch.int32, device=dev)
    nb = _lib.query_bytes("mrec_sparse_plan_workspace_bytes", n)
    # a workspace of its own per problem size: only plans of this n ever write it, and each of them hands the scratch table
    # and the scan's look-back words back clean -- so every call after the first skips the memsets (MREC_PLAN_WS_PRIMED)
    ws = workspace(f"plan:{n}:{sfx}", nb, dev)
    key = (ws.data_ptr(), n, sfx)
    primed = _PLAN_PRIMED.pop(key, False)          # (dropped while the call is in flight: an exception leaves it unprimed)
    _lib.call(f"mrec_sparse_plan_ex_{sfx}", _ptr(flat), n, _ptr(uniq), _ptr(inv), _ptr(n_uniq), _ptr(sorted_pos),
              _ptr(sorted_seg), _ptr(seg_offsets), _ptr(ws), ws.numel(), (1 if primed else 0) | (2 if skip_negative else 0), _stream())
    # a call issued under HIP-graph capture has not RUN: it primes nothing (it leaves a primed workspace primed -- every
    # replay of the captured chain hands the workspace back clean, as an eager call does)
    if primed or not torch.cuda.is_current_stream_capturing():
        _PLAN_PRIMED[key] = True
    return SparsePlan(Dedup(flat, uniq, inv, n_uniq), sorted_pos, sorted_seg, seg_offsets, n_uniq2[1:] if skip_negative else None)


def gather_rows(table, ids, row_scale=None, out=None, out_dtype=torch.float32):
    """ops.Gather / SparseGatherV2 / EmbeddingLookup (embedding.py:150,194; deep_and_cross.py:199),
    optionally fused with the mask multiply of wide_and_deep.py:303,308.  out_dtype=torch.bfloat16
    also fuses the half-precision cast in front of the MLP (wide_and_deep.py:122)."""
    _need_cuda(table, ids, row_scale)
    V, D, ld = _table(table)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
        if row_scale.dtype != torch.float32 or row_scale.numel() != n:
            raise TypeError("row_scale must be float32 with one value per id")
    if out_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError("gather_rows out_dtype must be float32, bfloat16 or float16")
    if out is None:
        out = torch.empty((n, D), dtype=out_dtype, device=table.device)
    fn = {torch.float32: "mrec_gather_rows_f32_", torch.bfloat16: "mrec_gather_rows_bf16_",
          torch.float16: "mrec_gather_rows_f16_"}[out.dtype]
    _lib.call(fn + sfx, _ptr(table), V, ld, D, _ptr(flat), n, _ptr(row_scale), _ptr(out), _stream())
    return out.view(tuple(ids.shape) + (D,))


def gather_rows_wide(table, ids, row_scale, wide_col, out=None, out_dtype=torch.bfloat16, packed_words=0, drop=None, step_state=None):
    """Both lookups of WideDeepModel.construct (wide_and_deep.py:300-302) in one pass over fused rows: returns
    (rows [.., D] in out_dtype, wide_prod [.., 2] with [.., 0] = table_row[wide_col] * row_scale and [.., 1] = 0).  `table` is the [V, D] view of the deep
    columns; wide_col is the column (relative to it, >= D) of the wide weight in the same rows.
    packed_words=W (>= D/2 + 2, multiple of 4): ONE float32 [n, W] result whose row is [D 16-bit values | product, 0 | pad] --
    a shard's answer message (one collective for both tables).  drop (Dropout; ids [B, F]): the looked-up rows are the
    [F * D] input of the DenseLayer the descriptor names and leave the kernel dropped out.  step_state (StepState): the kernel
    leaves its begin / end stamps there (StepState.lookup_ms)."""
    _need_cuda(table, ids, row_scale, out)
    V, D, ld = _table(table)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).contiguous()
        if row_scale.dtype != torch.float32 or row_scale.numel() != n:
            raise TypeError("row_scale must be float32 with one value per id")
    if out_dtype not in _DT16:
        raise TypeError("gather_rows_wide writes bfloat16 or float16 rows")
    kind = 1 if out_dtype == torch.bfloat16 else 2
    if packed_words:
        W = int(packed_words)
        if D % 2 or W < D // 2 + 2 or W % 4:
            raise ValueError("packed_words must be a multiple of 4 that holds D / 2 + 2 words")
        msg = torch.empty((max(n, 1), W), dtype=torch.float32, device=table.device)[:n]
        _lib.call("mrec_gather_rows_wide", _ptr(table), V, ld, D, _ptr(flat), flat.element_size(), n, _ptr(row_scale), _ptr(msg),
                  kind, 2 * W, int(wide_col), C.c_void_p(msg.data_ptr() + 2 * D), W, None, 0, _stream())
        return msg
    if out is None:
        out = torch.empty((n, D), dtype=out_dtype, device=table.device)
    wprod = torch.empty((max(n, 1), 2), dtype=torch.float32, device=table.device)[:n]      # (product, pad) pairs
    _lib.call("mrec_gather_rows_wide_ex", _ptr(table), V, ld, D, _ptr(flat), flat.element_size(), 1, n, _ptr(row_scale), 1, _ptr(out),
              1 if out.dtype == torch.bfloat16 else 2, D, int(wide_col), _ptr(wprod), 2, _drop_ref(drop),
              ids.shape[-1] if drop is not None else 0, 0, _ptr(step_state.buf) if step_state is not None else None, _stream())
    return out.view(tuple(ids.shape) + (D,)), wprod.view(tuple(ids.shape) + (2,))


def gather_rows_skip_(table, rows, out):
    """out[i, :] = table[rows[i], :] where rows[i] is a row of the table; rows of ids outside it (-1) are LEFT ALONE (the lookup in
    front has written them: KeyIndex.lookup(out=...))."""
    _need_cuda(table, rows, out)
    V, D, ld = _table(table)
    flat = rows.reshape(-1).contiguous()
    if flat.dtype != torch.int32 or out.dtype != torch.float32 or out.numel() != flat.numel() * D or not out.is_contiguous():
        raise TypeError("gather_rows_skip_: int32 rows, contiguous float32 [n, D] out")
    _lib.call("mrec_gather_rows_f32_skip_i32", _ptr(table), V, ld, D, _ptr(flat), flat.numel(), _ptr(out), _stream())
    return out


def gather_rows_pinned(host_table, ids):
    """Rows of a PINNED HOST table straight into HBM: the same gather kernel reads host memory over PCIe (pinned
    allocations are device-addressable; ~45 GB/s measured).  ids < 0 give zero rows without touching the host."""
    if host_table.is_cuda or not host_table.is_pinned():
        raise TypeError("gather_rows_pinned needs a pinned host table")
    _need_cuda(ids)
    V, D, ld = _table(host_table)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    out = torch.empty((flat.numel(), D), dtype=torch.float32, device=ids.device)
    _lib.call("mrec_gather_rows_f32_" + sfx, _ptr(host_table), V, ld, D, _ptr(flat), flat.numel(), None, _ptr(out), _stream())
    return out


def scatter_rows_pinned_(host_table, rows, vals):
    """host_table[rows[i], :] = vals[i, :] written by the device over PCIe (rows < 0 skipped); the host may read the
    table after the stream has been synchronised."""
    if host_table.is_cuda or not host_table.is_pinned():
        raise TypeError("scatter_rows_pinned_ needs a pinned host table")
    _need_cuda(rows, vals)
    V, D, ld = _table(host_table)
    n = rows.numel()
    vals = vals.reshape(n, D).contiguous()
    r32 = rows.to(torch.int32).contiguous()
    _lib.call("mrec_scatter_rows_f32", _ptr(host_table), ld, D, _ptr(r32), n, _ptr(vals), _stream())


def move_rows_(dst, dst_rows, src, src_rows, n_dev=None):
    """dst[dst_rows[i], :] = src[src_rows[i], :] for i < n_dev (device int64 word; None: every i); pairs with a negative row are
    skipped.  dst / src: float32 [*, W] device tensors or PINNED host tensors (device-addressable; moved over PCIe); the row
    lists are int64 device tensors of one length."""
    for t in (dst, src):
        if not t.is_cuda and not t.is_pinned():
            raise TypeError("move_rows_ needs device tensors or pinned host tensors")
    _need_cuda(dst_rows, src_rows, n_dev)
    if dst_rows.dtype != torch.int64 or src_rows.dtype != torch.int64 or dst_rows.numel() != src_rows.numel():
        raise TypeError("row lists must be int64 tensors of one length")
    _, W, ldd = _table(dst)
    _, Ws, lds = _table(src)
    if W != Ws:
        raise ValueError("row widths differ")
    _lib.call("mrec_move_rows_f32", _ptr(src), lds, _ptr(src_rows.contiguous()), _ptr(dst), ldd, _ptr(dst_rows.contiguous()),
              dst_rows.numel(), _ptr(n_dev), W, _stream())


def wide_sum(w, ids, wts, bias=None):
    """Wide branch of WideDeepModel.construct (wide_and_deep.py:300,303-306): [B]."""
    _need_cuda(w, ids, wts, bias)
    sfx = _suffix(ids)
    if ids.dim() != 2:
        raise ValueError("ids must be [B, F]")
    B, F = ids.shape
    if w.dtype != torch.float32 or w.dim() != 2 or w.shape[1] != 1:
        raise TypeError("w must be a float32 [V,1] table (any row stride)")
    V, ldw = w.shape[0], (w.stride(0) if w.shape[0] > 1 else 1)
    ids_c = ids.contiguous()
    wts_c = wts.contiguous()
    out = torch.empty(B, dtype=torch.float32, device=w.device)
    _lib.call(f"mrec_wide_sum_f32_{sfx}", _ptr(w), V, ldw, _ptr(ids_c), _ptr(wts_c), B, F, _ptr(bias), _ptr(out),
              _stream())
    return out


def _grads(plan, g, D, allow_bf16=False):
    if g.dtype != torch.float32 and not (allow_bf16 and g.dtype in (torch.bfloat16, torch.float16)):
        raise TypeError("row gradients must be float32" + (", bfloat16 or float16" if allow_bf16 else ""))
    g2 = g.reshape(plan.n, D)
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    return g2, (g2.stride(0) if plan.n > 1 else D)


def _row_scale(plan, row_scale):
    if row_scale is None:
        return None
    rs = row_scale.reshape(-1).contiguous()
    if rs.dtype != torch.float32 or rs.numel() != plan.n:
        raise TypeError("row_scale must be float32 with one value per id")
    return rs


def _apply_ws(plan, D, dev):
    nb = _lib.query_bytes("mrec_sparse_apply_workspace_bytes", plan.n, D)
    return workspace("apply", nb, dev)


def apply_window(D, aligned=True):
    """Sorted-index window of the sparse-apply kernels for width D (see include/mrec.h)."""
    return int(_lib.lib().mrec_sparse_apply_window(D, int(aligned)))


def segment_sum(plan, g, row_scale=None, grad_scale=1.0):
    """ops.UnsortedSegmentSum over the plan's groups: returns an [n, D] buffer whose first U rows
    are the per-unique-id sums (rows >= U are unspecified)."""
    _need_cuda(g, row_scale)
    D = g.shape[-1]
    g2, ldg = _grads(plan, g, D, allow_bf16=True)
    rs = _row_scale(plan, row_scale)
    out = torch.empty((max(plan.n, 1), D), dtype=torch.float32, device=g.device)
    ws = _apply_ws(plan, D, g.device)
    if g2.dtype == torch.float32:
        _lib.call("mrec_segment_sum_f32", _ptr(plan.sorted_pos), _ptr(plan.sorted_seg), _ptr(plan.seg_offsets), plan.n,
                  _ptr(g2), ldg, _ptr(rs), grad_scale, D, _ptr(out), _ptr(ws), ws.numel(), _stream())
    else:       # 16-bit row gradients (what the mixed-precision MLP backward produces): widened exactly, summed in fp32
        _lib.call("mrec_segment_sum_g16", _ptr(plan.sorted_pos), _ptr(plan.sorted_seg), _ptr(plan.seg_offsets), plan.n,
                  _ptr(g2), 1 if g2.dtype == torch.bfloat16 else 2, ldg, _ptr(rs), grad_scale, D, _ptr(out), _ptr(ws), ws.numel(), _stream())
    return out


def sparse_lazy_adam_(p, m, v, plan, g, row_scale=None, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9,
                      beta2_power=0.999, grad_scale=1.0, use_nesterov=False):
    """nn.LazyAdam on a RowTensor gradient (wide_and_deep.py:420-422): in place on p, m, v."""
    _need_cuda(p, m, v, g, row_scale)
    V, D, ld = _table(p)
    for t in (m, v):
        if _table(t) != (V, D, ld):
            raise ValueError("p, m, v must share shape and row stride")
    g2, ldg = _grads(plan, g, D, allow_bf16=True)
    rs = _row_scale(plan, row_scale)
    ws = _apply_ws(plan, D, p.device)
    sfx = _suffix(plan.uniq_buf)
    fn = {torch.float32: "mrec_sparse_lazy_adam_f32_", torch.bfloat16: "mrec_sparse_lazy_adam_bf16g_",
          torch.float16: "mrec_sparse_lazy_adam_f16g_"}[g2.dtype]
    _lib.call(fn + sfx, _ptr(p), _ptr(m), _ptr(v), V, ld, D, _ptr(plan.uniq_buf),
              _ptr(plan.sorted_pos), _ptr(plan.sorted_seg), _ptr(plan.seg_offsets), plan.n, _ptr(g2), ldg, _ptr(rs), lr,
              beta1, beta2, eps, beta1_power, beta2_power, grad_scale, int(use_nesterov), _ptr(ws), ws.numel(), _stream())


class ApplyFinish(C.Structure):      # mrec_apply_finish_t
    _fields_ = [("opaque", C.c_ubyte * 448)]


CONST_COLS_STATE_BYTES = 2592      # MREC_CONST_COLS_STATE_BYTES


def const_cols_state(device):
    """The device-side state of const_cols_detect (zeroed once, then owned by the library's calls)."""
    return torch.zeros(CONST_COLS_STATE_BYTES // 4, dtype=torch.int32, device=device)


def const_cols_detect(ids, V, state=None, min_count=None):
    """Which fields of the [B, F] batch `ids` are hot columns (mrec_const_cols_detect, include/mrec.h): ONE id -- the most frequent of
    the field's first 16 samples -- fills at least `min_count` of the field's B samples, is a row of a V-row table and occurs in no
    other field.  min_count = B (the default): constant columns, what the reference's Criteo pipeline gives the 13 dense features
    (process_data.py:138-147); smaller: also a field's dominant id (the bucket its rare categories fall into).  Returns the state
    tensor (const_cols_state) that holds the mask and the ids (const_cols_mask / const_cols_ids read them); hand it to
    sparse_lazy_adam_wide_(..., const_cols=(state, ids)).  None: more than 64 fields."""
    _need_cuda(ids)
    if ids.dim() != 2 or ids.dtype not in (torch.int32, torch.int64) or not ids.is_contiguous():
        raise TypeError("ids must be a contiguous [B, F] int32 / int64 tensor")
    B, F = ids.shape
    if F > 64 or B == 0:
        return None
    if state is None:
        state = const_cols_state(ids.device)
    _lib.call("mrec_const_cols_detect", _ptr(ids), ids.element_size(), B, F, int(V), int(B if min_count is None else max(1, min_count)),
              _ptr(state), _stream())
    return state


def const_cols_mask(state):
    """The constant columns of the last const_cols_detect over this state, as a Python int (bit f = field f); synchronises."""
    w = state[4:6].cpu().numpy().view(np.uint64)
    return int(w[0])


def const_cols_ids(state):
    """{field: hot id} of the last const_cols_detect over this state; synchronises."""
    a = state.cpu().numpy()
    m = int(a[4:6].view(np.uint64)[0])
    hid = a[8:8 + 128].view(np.int64)
    return {f: int(hid[f]) for f in range(64) if (m >> f) & 1}


def sparse_lazy_adam_wide_(p, m, v, plan, g, row_scale, gw, F, wide_col, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                           beta1_power=0.9, beta2_power=0.999, grad_scale=1.0, use_nesterov=False, ftrl_lr=5e-2, l1=1e-8, l2=1e-8,
                           lr_power=-0.5, step_state=None, defer=False, const_cols=None):
    """LazyAdam on the deep columns and FTRL on the wide record of the same fused rows in ONE pass (wide_and_deep.py:420-430):
    gw [n / F] is the wide branch's gradient per sample (the head's dlogit); position i contributes gw[i // F] * row_scale[i].
    step_state (StepState): the Adam step size comes from device memory (beta powers ignored) and the main kernel stamps its
    begin / end there.  const_cols = (const_cols_detect(ids, V), ids): the batch's constant columns are summed sample by sample
    by the same launch instead of through the index (a different, fixed order of additions for those rows)."""
    _need_cuda(p, m, v, g, row_scale, gw)
    V, D, ld = _table(p)
    for t in (m, v):
        if _table(t) != (V, D, ld):
            raise ValueError("p, m, v must share shape and row stride")
    g2, ldg = _grads(plan, g, D, allow_bf16=True)
    rs = _row_scale(plan, row_scale)
    if gw.dtype != torch.float32 or gw.numel() * F != plan.n:
        raise TypeError("gw must be float32 with one value per sample (n / F)")
    if gw.dim() == 2 and gw.shape[1] == 1:
        gws = gw.stride(0) if gw.shape[0] > 1 else 1            # a column of wider rows (a shard's gradient message)
    elif gw.is_contiguous():
        gws = 1
    else:
        raise TypeError("gw must be contiguous, or an [n, 1] column view")
    nb = _lib.query_bytes("mrec_sparse_apply_workspace_bytes", max(plan.n, 1), D + 4)
    ws = workspace("apply", nb, p.device)
    kind = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[g2.dtype]
    args = (_ptr(p), _ptr(m), _ptr(v), V, ld, D, _ptr(plan.uniq_buf), plan.uniq_buf.element_size(),
            _ptr(plan.sorted_pos), _ptr(plan.sorted_seg), _ptr(plan.seg_offsets), plan.n, _ptr(g2), kind, ldg, _ptr(rs), lr,
            beta1, beta2, eps, beta1_power, beta2_power, grad_scale, int(use_nesterov), _ptr(gw), int(gws), int(F), int(wide_col),
            ftrl_lr, l1, l2, lr_power, _ptr(ws), ws.numel(), _ptr(step_state.buf) if step_state is not None else None,
            _ptr(getattr(plan, "n_valid_dev", None)))
    keep_cc = None
    if const_cols is not None and const_cols[0] is not None:
        bad, ids = const_cols
        _need_cuda(bad, ids)
        if (ids.dim() != 2 or ids.shape[1] != F or ids.numel() != plan.n or not ids.is_contiguous()
                or ids.element_size() != plan.uniq_buf.element_size() or bad.dtype != torch.int32 or bad.numel() * 4 != CONST_COLS_STATE_BYTES):
            raise TypeError("const_cols: (the state of const_cols_detect, the contiguous [n / F, F] id batch of the plan, ids of the table's key width)")
        _lib.call("mrec_sparse_apply_next_const_cols", _ptr(bad), _ptr(ids), ids.element_size(), ids.shape[0])
        keep_cc = (bad, ids)
    if defer:
        # the finishing pass (runs of duplicates that cross windows of the sorted index) is handed back: dense_adam_slabs_(...,
        # finish=...) runs it as the first workgroups of the dense net's Adam launch.  The record points into `ws`, the plan and
        # the tables: run the finish before any of them is reused.
        fin = ApplyFinish()
        fin.keep = (ws, plan, p, g2, rs, gw, keep_cc)
        _lib.call("mrec_sparse_lazy_adam_wide_defer", *args, C.cast(C.pointer(fin), C.c_void_p), _stream())
        return fin
    _lib.call("mrec_sparse_lazy_adam_wide", *args, _stream())
    return None


def sparse_ftrl_(var, accum, linear, plan, g, row_scale=None, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5, grad_scale=1.0):
    """nn.FTRL sparse apply (wide_and_deep.py:423-430): in place on var, accum, linear."""
    _need_cuda(var, accum, linear, g, row_scale)
    V, D, ld = _table(var)
    for t in (accum, linear):
        if _table(t) != (V, D, ld):
            raise ValueError("var, accum, linear must share shape and row stride")
    g2, ldg = _grads(plan, g, D)
    rs = _row_scale(plan, row_scale)
    ws = _apply_ws(plan, D, var.device)
    sfx = _suffix(plan.uniq_buf)
    _lib.call(f"mrec_sparse_ftrl_f32_{sfx}", _ptr(var), _ptr(accum), _ptr(linear), V, ld, D, _ptr(plan.uniq_buf),
              _ptr(plan.sorted_pos), _ptr(plan.sorted_seg), _ptr(plan.seg_offsets), plan.n, _ptr(g2), ldg, _ptr(rs), lr, l1,
              l2, lr_power, grad_scale, _ptr(ws), ws.numel(), _stream())


def _flat_same(*ts):
    n = ts[0].numel()
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise TypeError("dense optimizer tensors must be contiguous float32 of equal size")
    return n


class _Ftrl1(C.Structure):         # mrec_ftrl1_t
    _fields_ = [("index", C.c_int64), ("lr", C.c_float), ("l1", C.c_float), ("l2", C.c_float), ("lr_power", C.c_float)]


def _ftrl1(one):
    """one = (index, lr, l1, l2, lr_power): the element of a dense buffer that belongs to FTRL (its m word = accum, its v word =
    linear) -- Wide&Deep's `wide_b` (wide_and_deep.py:407-411)."""
    idx, lr, l1, l2, lrp = one
    return _Ftrl1(int(idx), float(lr), float(l1), float(l2), float(lrp))


def dense_adam_(p, m, v, g, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9, beta2_power=0.999,
                grad_scale=1.0, use_nesterov=False, shadow_bf16=None, ftrl1=None):
    """nn.Adam over a whole tensor (wide_and_deep.py:435-437; deep_and_cross.py:342-344).  g may be
    bfloat16 (widened on load); shadow_bf16 (optional bf16 tensor of the same size) receives the updated
    parameters rounded to bf16, ready to be the next forward's GEMM operand.  ftrl1: see _ftrl1."""
    _need_cuda(p, m, v, g, shadow_bf16)
    n = _flat_same(p, m, v)
    if ftrl1 is not None:
        if shadow_bf16 is not None or g.dtype != torch.float32 or g.numel() != n or not g.is_contiguous():
            raise TypeError("dense_adam_(ftrl1=...): contiguous float32 gradient, no shadow")
        f = _ftrl1(ftrl1)
        _lib.call("mrec_dense_adam_one_ftrl_f32", _ptr(p), _ptr(m), _ptr(v), _ptr(g), n, lr, beta1, beta2, eps, beta1_power,
                  beta2_power, grad_scale, int(use_nesterov), C.cast(C.pointer(f), C.c_void_p), _stream())
        return
    if g.numel() != n or not g.is_contiguous() or g.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("dense gradient must be contiguous float32 or bfloat16 of the parameter's size")
    if shadow_bf16 is not None and (shadow_bf16.dtype != torch.bfloat16 or shadow_bf16.numel() != n
                                    or not shadow_bf16.is_contiguous()):
        raise TypeError("shadow_bf16 must be a contiguous bfloat16 tensor of the parameter's size")
    _lib.call("mrec_dense_adam_ex_f32", _ptr(p), _ptr(m), _ptr(v), _ptr(g), int(g.dtype == torch.bfloat16),
              _ptr(shadow_bf16), n, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale, int(use_nesterov),
              _stream())


def dense_adam_l2_(p, m, v, g, l2_scaled, sumsq=None, accumulate=False, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9,
                   beta2_power=0.999, grad_scale=1.0, use_nesterov=False):
    """nn.Adam over a whole table whose loss carries l2_coef * sum(p^2) / 2 (wide_and_deep.py:356-360; deepfm.py:252-259): `g` holds
    the scattered row-gradient sums only; the kernel adds l2_scaled * p (= l2_coef * sens * p) and leaves sum(p^2) of the values
    before the update in `sumsq` (float64 [1] device tensor; accumulate=True: added onto it) -- one pass instead of four."""
    _need_cuda(p, m, v, g, sumsq)
    n = _flat_same(p, m, v, g)
    if sumsq is not None and (sumsq.dtype != torch.float64 or sumsq.numel() != 1):
        raise TypeError("sumsq must be a float64 [1] device tensor")
    ws = workspace("adam_l2", _lib.query_bytes("mrec_dense_adam_l2_workspace_bytes", n), p.device)
    _lib.call("mrec_dense_adam_l2_f32", _ptr(p), _ptr(m), _ptr(v), _ptr(g), n, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale,
              int(use_nesterov), float(l2_scaled), _ptr(sumsq), int(bool(accumulate)), _ptr(ws), ws.numel(), _stream())


def dense_adam_rows_l2_(p, m, v, plan, sums, l2_scaled=0.0, sumsq=None, accumulate=False, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                        beta1_power=0.9, beta2_power=0.999, grad_scale=1.0, use_nesterov=False, step_state=None):
    """nn.Adam over a whole table [V, D] whose gradient is the bprop of a dense Gather: `sums` [>= U, D] = the row-gradient sums of the
    plan's groups (ops.segment_sum), every other row's gradient is zero (+ l2_scaled * p everywhere, sum(p^2) into `sumsq` as
    dense_adam_l2_).  One pass over p, m, v -- no [V, D] gradient is zeroed, scattered into and read back.  step_state: the step size
    from device memory (a captured step)."""
    _need_cuda(p, m, v, sums, sumsq)
    V, D, ld = _table(p)
    for t in (p, m, v):
        if _table(t) != (V, D, D) or t.dtype != torch.float32:
            raise TypeError("dense_adam_rows_l2_: p, m, v must be contiguous float32 [V, D] tables")
    uniq = plan.uniq_buf if plan.uniq_buf.dtype == torch.int32 else plan.uniq_buf.to(torch.int32)
    U = uniq.numel()
    if sums.dtype != torch.float32 or sums.dim() != 2 or sums.shape[1] != D or sums.shape[0] < U or not sums.is_contiguous():
        raise TypeError("dense_adam_rows_l2_: sums must be contiguous float32 [>= U, D]")
    if sumsq is not None and (sumsq.dtype != torch.float64 or sumsq.numel() != 1):
        raise TypeError("sumsq must be a float64 [1] device tensor")
    ws = workspace(f"adam_rows:{V}:{D}", _lib.query_bytes("mrec_dense_adam_rows_l2_workspace_bytes", V, D), p.device)
    _lib.call("mrec_dense_adam_rows_l2_f32", _ptr(p), _ptr(m), _ptr(v), V, D, _ptr(uniq), U, _ptr(plan.n_uniq_dev), _ptr(sums), lr, beta1,
              beta2, eps, beta1_power, beta2_power, grad_scale, int(use_nesterov), float(l2_scaled), _ptr(sumsq), int(bool(accumulate)),
              C.c_void_p(step_state.buf.data_ptr()) if step_state is not None else None, _ptr(ws), ws.numel(), _stream())


def dense_ftrl_(var, accum, linear, g, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5, grad_scale=1.0):
    """nn.FTRL over a whole tensor (wide_and_deep.py:438-445)."""
    _need_cuda(var, accum, linear, g)
    n = _flat_same(var, accum, linear, g)
    _lib.call("mrec_dense_ftrl_f32", _ptr(var), _ptr(accum), _ptr(linear), _ptr(g), n, lr, l1, l2, lr_power, grad_scale,
              _stream())


# ---- MapParameter key index ------------------------------------------------------------------
class KeyIndex:
    """Device key -> row-number index behind MapParameter (embedding.py:136-146)."""

    def __init__(self, capacity, device):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        nb = _lib.query_bytes("mrec_map_bytes", self.capacity)
        self._mem = torch.empty(nb + 256, dtype=torch.uint8, device=self.device)
        off = (-self._mem.data_ptr()) % 256
        self._base = self._mem.data_ptr() + off
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call("mrec_map_create", C.byref(h), C.c_void_p(self._base), nb, self.capacity, _stream())
        self._h = h
        cptr = _lib.lib().mrec_map_counters_dev(self._h)
        self._counters_off = cptr - self._mem.data_ptr()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:          # _lib is None during interpreter shutdown
            try:
                _lib.lib().mrec_map_destroy(h)
            except Exception:
                pass
            self._h = None

    def counters(self):
        """(rows handed out, live keys, dropped keys, free rows) -- host sync."""
        c = self._mem[self._counters_off: self._counters_off + 32].view(torch.int64)
        return tuple(int(x) for x in c.tolist())

    def counters_all(self):
        """All eight device counter words ([4] = tombstones, [6] = slot-array rebuilds) -- host sync."""
        c = self._mem[self._counters_off: self._counters_off + 64].view(torch.int64)
        return tuple(int(x) for x in c.tolist())

    def __len__(self):
        return self.counters()[1]

    def _view(self, ptr, nbytes, dtype):
        off = int(ptr) - self._mem.data_ptr()
        return self._mem[off: off + nbytes].view(dtype)

    def tracking(self):
        """(hits int32 [C], last_step int32 [C], dirty uint8 [C]) views of the index's per-row counters."""
        h, l, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.call("mrec_map_tracking_dev", self._h, C.byref(h), C.byref(l), C.byref(d))
        n = self.capacity
        return self._view(h.value, 4 * n, torch.int32), self._view(l.value, 4 * n, torch.int32), self._view(d.value, n, torch.uint8)

    def row_keys(self):
        """int64 [C] view: the key each row holds (valid for live rows)."""
        return self._view(_lib.lib().mrec_map_row_keys_dev(self._h), 8 * self.capacity, torch.int64)

    def _ws(self, n):
        nb = _lib.query_bytes("mrec_map_workspace_bytes", max(n, 1))
        return workspace("map", nb, self.device)

    def find_or_insert(self, keys_i64, insert=True, n_dev=None):
        """keys must be unique within the call.  Returns (rows int32[n], is_new uint8[n]).
        n_dev: optional device int64 word; only the first min(n, n_dev) keys are processed."""
        n = keys_i64.numel()
        rows = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n]
        is_new = torch.empty(max(n, 1), dtype=torch.uint8, device=self.device)[:n]
        ws = self._ws(n)
        _lib.call("mrec_map_find_or_insert", self._h, _ptr(keys_i64), n, _ptr(n_dev), int(insert), _ptr(rows), _ptr(is_new),
                  _ptr(ws), ws.numel(), _stream())
        return rows, is_new

    # ---- MapTensorGet as one chain (mrec_map_lookup) -----------------------------------------------------------
    class _MapTable(C.Structure):
        _fields_ = [("rows", C.c_void_p), ("ld", C.c_int64), ("D", C.c_int32), ("sigma", C.c_float), ("fill", C.c_float),
                    ("seed", C.c_uint64)]

    @classmethod
    def _tables(cls, tables):
        arr = (cls._MapTable * max(len(tables), 1))()
        for q, (t, sigma, fill, seed) in enumerate(tables):
            _need_cuda(t)
            V, D, ld = _table(t)
            arr[q] = cls._MapTable(t.data_ptr(), ld, D, -1.0 if sigma is None else float(sigma), float(fill or 0.0), int(seed))
        return arr

    def lookup(self, keys, insert=True, unique=False, train=False, step=0, permit=1, tables=(), n_dev=None, want_admitted=False,
               skip_pad=False, out=None, out_table=0):
        """Rows of `keys` (int32 / int64, any shape, duplicates allowed) in 3 launches (1 when not inserting): probe, rank and
        place the missing keys in order of first appearance, default rows of `tables` = [(tensor [C, D], sigma or None, fill,
        seed)] + admission.  Returns rows int32 [n] (and the admitted rows when want_admitted).  skip_pad: key -1 is a padding
        slot (row -1, never inserted)."""
        _need_cuda(keys)
        flat = keys.reshape(-1).contiguous()
        n = flat.numel()
        rows = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n]
        adm = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n] if want_admitted else None
        nb = _lib.query_bytes("mrec_map_lookup_workspace_bytes", max(n, 1))
        ws = workspace(f"maplookup:{n}", nb, self.device)           # one per problem size, so it stays primed
        key = (ws.data_ptr(), n)
        primed = _MAP_PRIMED.pop(key, False)
        flags = (1 if insert else 0) | (2 if unique else 0) | (4 if train else 0) | (8 if primed else 0) | (16 if skip_pad else 0)
        tabs = self._tables(tables)
        if out is not None:
            # the lookup's output rows of NEW keys are written by the kernel that generates their default rows; returns
            # (rows, rows for the gather behind this call: -1 where `out` is written already)
            if not insert or not tables or out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] != n or out.stride(1) != 1:
                raise TypeError("lookup(out=...): an inserting lookup with tables and a float32 [n, D] output")
            rows_g = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)[:n]
            _lib.call("mrec_map_lookup_out", self._h, _ptr(flat), flat.element_size(), n, _ptr(n_dev), flags, int(step), int(permit),
                      C.cast(tabs, C.c_void_p), len(tables), _ptr(rows), _ptr(adm), _ptr(out), out.stride(0), int(out_table), _ptr(rows_g),
                      _ptr(ws), ws.numel(), _stream())
        else:
            _lib.call("mrec_map_lookup", self._h, _ptr(flat), flat.element_size(), n, _ptr(n_dev), flags, int(step), int(permit),
                      C.cast(tabs, C.c_void_p), len(tables), _ptr(rows), _ptr(adm), _ptr(ws), ws.numel(), _stream())
        if primed or (insert and not torch.cuda.is_current_stream_capturing()):
            _MAP_PRIMED[key] = True        # (a probe-only call leaves the workspace untouched; a captured call has not run)
        if out is not None:
            return (rows, adm, rows_g) if want_admitted else (rows, rows_g)
        return (rows, adm) if want_admitted else rows

    def fill_missing(self, keys, rows, out, sigma, fill, seed):
        flat = keys.reshape(-1).contiguous()
        tab = self._MapTable(0, out.stride(0), out.shape[1], -1.0 if sigma is None else float(sigma), float(fill or 0.0), int(seed))
        _lib.call("mrec_map_fill_missing", _ptr(flat), flat.element_size(), _ptr(rows), flat.numel(), _ptr(out), out.stride(0),
                  C.byref(tab), _stream())

    def evict(self, step, threshold):
        """Device-side eviction; returns the device word holding the count (no host sync)."""
        n_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        ws = workspace("mapevict", 4 * ((self.capacity + 2047) // 2048) + 256, self.device)
        _lib.call("mrec_map_evict", self._h, int(step), int(threshold), _ptr(n_dev), _ptr(ws), ws.numel(), _stream())
        return n_dev

    def export_dirty(self, clear=True):
        """(keys int64, rows int32, status int32) of the rows modified and the keys erased since the last clearing call."""
        cap2 = 2 * self.capacity
        keys = torch.empty(cap2, dtype=torch.int64, device=self.device)
        rows = torch.empty(cap2, dtype=torch.int32, device=self.device)
        status = torch.empty(cap2, dtype=torch.int32, device=self.device)
        n_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        nb = _lib.query_bytes("mrec_map_workspace_bytes", self.capacity) + self.capacity + 512
        ws = workspace("mapexport", nb, self.device)
        _lib.call("mrec_map_export_dirty", self._h, _ptr(keys), _ptr(rows), _ptr(status), _ptr(n_dev), int(bool(clear)), _ptr(ws),
                  ws.numel(), _stream())
        n = int(n_dev.item())
        return keys[:n], rows[:n], status[:n]

    def mark_dirty(self, rows):
        _lib.call("mrec_map_mark_dirty", self._h, _ptr(rows), rows.numel(), _stream())

    def erase(self, keys_i64):
        n = keys_i64.numel()
        ws = self._ws(n)
        _lib.call("mrec_map_erase", self._h, _ptr(keys_i64), n, _ptr(ws), ws.numel(), _stream())

    def export(self):
        """Live (keys int64, rows int32) in row order (host sync for the count)."""
        keys = torch.empty(self.capacity, dtype=torch.int64, device=self.device)
        rows = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        n_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        ws = self._ws(self.capacity)
        _lib.call("mrec_map_export", self._h, _ptr(keys), _ptr(rows), _ptr(n_dev), _ptr(ws), ws.numel(), _stream())
        n = int(n_dev.item())
        return keys[:n], rows[:n]


def init_rows_(table, rows, keys_i64, is_new=None, seed=0, sigma=0.01, fill=None, n_dev=None):
    """Default-value rows of MapTensorGet(insert_default_value=True) (embedding.py:149)."""
    V, D, ld = _table(table)
    n = rows.numel()
    s = -1.0 if fill is not None else float(sigma)
    _lib.call("mrec_init_rows_f32", _ptr(table), ld, D, _ptr(rows), _ptr(keys_i64), _ptr(is_new), n, _ptr(n_dev),
              C.c_uint64(seed), s, float(fill or 0.0), _stream())


def copy3_(dsts, srcs):
    """Three contiguous device tensors copied in one launch (a step's inputs into its graph's static buffers); falls back to
    tensor.copy_ when a size is not a multiple of 16 bytes."""
    _need_cuda(*dsts, *srcs)
    nb = [d.numel() * d.element_size() for d in dsts]
    ok = all(d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.numel() == s_.numel() and n % 16 == 0
             and d.data_ptr() % 16 == 0 and s_.data_ptr() % 16 == 0 for d, s_, n in zip(dsts, srcs, nb))
    if not ok:
        for d, s_ in zip(dsts, srcs):
            d.copy_(s_)
        return
    _lib.call("mrec_copy3", _ptr(dsts[0]), _ptr(srcs[0]), nb[0], _ptr(dsts[1]), _ptr(srcs[1]), nb[1], _ptr(dsts[2]), _ptr(srcs[2]),
              nb[2], _stream())


def copy_many_(dsts, srcs):
    """Up to 24 contiguous device tensors copied in one launch (all the inputs of a sink of steps into its graph's static
    buffers); falls back to copy3_ / tensor.copy_ otherwise."""
    _need_cuda(*dsts, *srcs)
    n = len(dsts)
    nb = [d.numel() * d.element_size() for d in dsts]
    ok = n <= 24 and all(d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.numel() == s_.numel() and b % 16 == 0
                         and d.data_ptr() % 16 == 0 and s_.data_ptr() % 16 == 0 for d, s_, b in zip(dsts, srcs, nb))
    if not ok:
        for d, s_ in zip(dsts, srcs):
            d.copy_(s_)
        return
    da = (C.c_void_p * n)(*[d.data_ptr() for d in dsts])
    sa = (C.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
    ba = (C.c_int64 * n)(*nb)
    _lib.call("mrec_copy_many", n, C.cast(da, C.c_void_p), C.cast(sa, C.c_void_p), C.cast(ba, C.c_void_p), _stream())


def put_rows_last_(table, rows, vals, winner):
    """table[rows[i]] = vals[i] for the LAST position of every row (duplicates: a sequential upsert); winner: int32 [rows of
    table] scratch, all -1 before and after."""
    _need_cuda(table, rows, vals, winner)
    V, D, ld = _table(table)
    vals = vals.contiguous()
    _lib.call("mrec_put_rows_last_f32", _ptr(table), ld, D, _ptr(rows), rows.numel(), _ptr(vals), _ptr(winner), _stream())


def compose_i32(table, idx):
    """out[i] = table[idx[i]] on int32 arrays: rows per position from rows per unique key."""
    n = idx.numel()
    out = torch.empty(max(n, 1), dtype=torch.int32, device=idx.device)[:n]
    _lib.call("mrec_compose_i32", _ptr(table), _ptr(idx), n, _ptr(out), _stream())
    return out


def widen_keys(keys):
    """int32 -> int64 key widening (no-op for int64)."""
    if keys.dtype == torch.int64:
        return keys.contiguous()
    if keys.dtype != torch.int32:
        raise TypeError(f"keys must be int32 or int64, got {keys.dtype}")
    k = keys.contiguous()
    out = torch.empty(k.shape, dtype=torch.int64, device=k.device)
    _lib.call("mrec_widen_i32_i64", _ptr(k), k.numel(), _ptr(out), _stream())
    return out


def scatter_unique_rows_(table, plan, vals):
    """table[plan.uniq[u], :] = vals[u, :] for u < U (device-side U: rows past it are masked off).
    Densifies a segment-sum into the [V, D] gradient of a Gather (deep_and_cross.py:199 bprop)."""
    rows = plan.uniq_buf if plan.uniq_buf.dtype == torch.int32 else plan.uniq_buf.to(torch.int32)
    n = rows.numel()
    valid = torch.arange(n, device=rows.device) < plan.n_uniq_dev
    rows = torch.where(valid, rows, torch.full_like(rows, -1))
    scatter_rows_(table, rows, vals[:n])


def scatter_unique_rows_add_(table, plan, vals):
    """table[plan.uniq[u], :] += vals[u, :] for u < U (rows are distinct)."""
    rows = plan.uniq_buf if plan.uniq_buf.dtype == torch.int32 else plan.uniq_buf.to(torch.int32)
    n = rows.numel()
    valid = torch.arange(n, device=rows.device) < plan.n_uniq_dev
    rows = torch.where(valid, rows, torch.full_like(rows, -1))
    V, D, ld = _table(table)
    v = vals[:n].reshape(n, D).contiguous()
    _lib.call("mrec_scatter_add_rows_f32", _ptr(table), ld, D, _ptr(rows), n, _ptr(v), _stream())


def scatter_rows_(table, rows, vals):
    """MapTensorPut on the row storage (README.md:188-190)."""
    V, D, ld = _table(table)
    n = rows.numel()
    vals = vals.reshape(n, D).contiguous()
    _lib.call("mrec_scatter_rows_f32", _ptr(table), ld, D, _ptr(rows), n, _ptr(vals), _stream())


# ---- DCN-v1 cross layers ---------------------------------------------------------------------
def cross_layers(x0, w, b):
    """CrossLayer x L (deep_and_cross.py:139-149, 300-306).  x0 [B,D], w/b [L,D] -> [B,D]."""
    _need_cuda(x0, w, b)
    x0 = x0.contiguous(); w = w.contiguous(); b = b.contiguous()
    B, D = x0.shape
    L = w.shape[0]
    out = torch.empty_like(x0)
    _lib.call("mrec_cross_layers_f32", _ptr(x0), _ptr(w), _ptr(b), L, B, D, _ptr(out), _stream())
    return out


def cross_layers_bwd(x0, w, b, dy, dx0_out=None, dw_out=None, db_out=None, accumulate=False):
    """(dx0, dw, db) of the cross stack; *_out: contiguous tensors to write into (views of a flat gradient buffer).
    accumulate: dx0_out += dx0 (the gradient another branch left there)."""
    _need_cuda(x0, w, b, dy, dx0_out, dw_out, db_out)
    x0 = x0.contiguous(); w = w.contiguous(); b = b.contiguous(); dy = dy.contiguous()
    B, D = x0.shape
    L = w.shape[0]
    for t_, ref in ((dx0_out, x0), (dw_out, w), (db_out, b)):
        if t_ is not None and (t_.shape != ref.shape or t_.dtype != torch.float32 or not t_.is_contiguous()):
            raise TypeError("cross_layers_bwd: outputs must be contiguous float32 tensors of the inputs' shapes")
    dx0 = dx0_out if dx0_out is not None else torch.empty_like(x0)
    dw = dw_out if dw_out is not None else torch.empty_like(w)
    db = db_out if db_out is not None else torch.empty_like(b)
    nb = _lib.query_bytes("mrec_cross_layers_bwd_workspace_bytes", L, B, D)
    ws = workspace("cross", nb, x0.device)
    if accumulate and dx0_out is None:
        raise TypeError("cross_layers_bwd: accumulate needs dx0_out")
    _lib.call("mrec_cross_layers_bwd_acc_f32" if accumulate else "mrec_cross_layers_bwd_f32", _ptr(x0), _ptr(w), _ptr(b), L, B, D, _ptr(dy), _ptr(dx0), _ptr(dw), _ptr(db),
              _ptr(ws), ws.numel(), _stream())
    return dx0, dw, db


# ---- row-shard routing -----------------------------------------------------------------------
def shard_route(ids, n_shards, hashed=False):
    """Buckets ids by owner = id mod n_shards (stable).  Returns (send_local, send_perm, counts_dev).
    hashed=True: hash tables keyed by the raw id -- owner = hash(key) mod n_shards and send_local holds the raw keys."""
    _need_cuda(ids)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    dev = flat.device
    send_local = torch.empty(max(n, 1), dtype=flat.dtype, device=dev)[:n]
    send_perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    counts = torch.empty(n_shards, dtype=torch.int64, device=dev)
    nb = _lib.query_bytes("mrec_shard_route_workspace_bytes", n, n_shards)
    ws = workspace("route", nb, dev)
    _lib.call(f"mrec_shard_route_{'hash_' if hashed else ''}{sfx}", _ptr(flat), n, n_shards, _ptr(send_local), _ptr(send_perm), _ptr(counts), _ptr(ws),
              ws.numel(), _stream())
    return send_local, send_perm, counts


def shard_unroute(rows, send_perm, row_scale=None, out=None, cols=None):
    """out[send_perm[k], :] = rows[k, :cols] (* row_scale): a request's answers back in position order.  `rows` may be a
    column window of wider message rows (any row stride)."""
    n = rows.shape[0]
    D = rows.shape[1] if cols is None else int(cols)
    if rows.dim() != 2 or rows.stride(1) != 1 or rows.dtype != torch.float32:
        raise TypeError("shard_unroute: rows must be float32 [n, D] with unit column stride")
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=rows.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (n, D) or not out.is_contiguous():
        raise TypeError("shard_unroute: out must be a contiguous float32 [n, D] tensor")
    rs = row_scale.reshape(-1).contiguous() if row_scale is not None else None
    _lib.call("mrec_shard_unroute_ld_f32", _ptr(rows), rows.stride(0) if n > 1 else max(D, rows.shape[1]), _ptr(send_perm), n, D,
              _ptr(rs), _ptr(out), D, _stream())
    return out


def shard_route_rows(g, send_perm, row_scale=None, out=None):
    """rows_out[k, :D] = g[send_perm[k], :] (* row_scale): rows bucketed by owner.  out: a float32 [n, >= D] column window of a
    wider message (any row stride) to write into."""
    n = send_perm.numel()
    D = g.shape[-1]
    g2 = g.reshape(n, D)
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=g.device)
    elif out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] != n or out.shape[1] < D or out.stride(1) != 1:
        raise TypeError("shard_route_rows: out must be a float32 [n, >= D] tensor with unit column stride")
    rs = row_scale.reshape(-1).contiguous() if row_scale is not None else None
    _lib.call("mrec_shard_route_rows_ld_f32", _ptr(g2), g2.stride(0) if n > 1 else D, _ptr(send_perm), n, D, _ptr(rs),
              _ptr(out), out.stride(0) if n > 1 else max(D, out.shape[1]), _stream())
    return out


def shard_pack_iw(send_local, wts, send_perm):
    """One request message instead of two: int32 [n, 2] rows {local id, bits of the position's weight}."""
    _need_cuda(send_local, wts, send_perm)
    if send_local.dtype != torch.int32:
        raise TypeError("shard_pack_iw packs int32 ids")
    n = send_local.numel()
    out = torch.empty((max(n, 1), 2), dtype=torch.int32, device=send_local.device)[:n]
    _lib.call("mrec_shard_pack_iw_i32", _ptr(send_local), _ptr(wts.reshape(-1).contiguous()), _ptr(send_perm), n, _ptr(out), _stream())
    return out


def shard_unpack_iw(pairs):
    n = pairs.shape[0]
    ids = torch.empty(max(n, 1), dtype=torch.int32, device=pairs.device)[:n]
    wts = torch.empty(max(n, 1), dtype=torch.float32, device=pairs.device)[:n]
    _lib.call("mrec_shard_unpack_iw_i32", _ptr(pairs), n, _ptr(ids), _ptr(wts), _stream())
    return ids, wts


# ---- fixed-capacity routing: a sharded step with static message shapes (include/mrec.h) -----------------------------------
def shard_capacity(n, n_shards, factor=1.25):
    """Request slots a rank reserves per owner: ceil(factor * n / n_shards) rounded up to a multiple of 64, never more than n
    (one shard: exactly n)."""
    if n_shards <= 1:
        return int(n)
    c = -(-int(n * factor) // n_shards)
    return int(min(n, -(-c // 64) * 64))


def shard_route_slots(ids, wts, n_shards, cap, hashed=False, overflow=None, rot=0, out=None, n_valid_dev=None):
    """Request message of a step (all owners, `cap` slots each): returns (req, slot_of_pos int32 [n], pos_of_slot int32
    [n_shards * cap]); req is int32 [n_shards * cap, 2] ({id, weight bits}) for int32 ids, int64-addressable
    [n_shards * cap, 2] ({key, weight bits | 0}) for int64.  overflow: int64 [1] device counter (sticky).  rot: owner o's slots
    are chunk (o - rot) mod n_shards of the message.  out: the message buffer to write (contiguous, req's shape and dtype)."""
    _need_cuda(ids, wts, overflow, out)
    sfx = _suffix(ids)
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    dev = flat.device
    ns = int(n_shards) * int(cap)
    if out is not None:
        if out.dtype != flat.dtype or tuple(out.shape) != (ns, 2) or not out.is_contiguous():
            raise TypeError("out must be a contiguous [n_shards * cap, 2] tensor of the ids' dtype")
        req = out
    else:
        req = torch.empty((ns, 2), dtype=flat.dtype, device=dev)
    slot_of_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    pos_of_slot = torch.empty(max(ns, 1), dtype=torch.int32, device=dev)[:ns]
    w = None
    if wts is not None:
        w = wts.reshape(-1).contiguous()
        if w.dtype != torch.float32 or w.numel() != n:
            raise TypeError("wts must be float32 with one value per id")
    nb = _lib.query_bytes("mrec_shard_route_slots_workspace_bytes", n, n_shards)
    ws = workspace("route_slots", nb, dev)
    if n_valid_dev is not None:
        # the list's length lives on the device (a step's unique ids: Dedup.n_uniq_dev): entries past it get no slot
        _need_cuda(n_valid_dev)
        _lib.call(f"mrec_shard_route_slots_nv_{sfx}", _ptr(flat), _ptr(w), n, _ptr(n_valid_dev), int(n_shards), int(cap), int(bool(hashed)),
                  int(rot), _ptr(req), _ptr(slot_of_pos), _ptr(pos_of_slot), _ptr(overflow), _ptr(ws), ws.numel(), _stream())
        return req, slot_of_pos, pos_of_slot
    _lib.call(f"mrec_shard_route_slots_{sfx}", _ptr(flat), _ptr(w), n, int(n_shards), int(cap), int(bool(hashed)), int(rot), _ptr(req),
              _ptr(slot_of_pos), _ptr(pos_of_slot), _ptr(overflow), _ptr(ws), ws.numel(), _stream())
    return req, slot_of_pos, pos_of_slot


def shard_unpack_req(req):
    """(ids [n_slots] of req's dtype, wts float32 [n_slots]) of a received request message."""
    _need_cuda(req)
    ns = req.shape[0]
    ids = torch.empty(max(ns, 1), dtype=req.dtype, device=req.device)[:ns]
    wts = torch.empty(max(ns, 1), dtype=torch.float32, device=req.device)[:ns]
    _lib.call("mrec_shard_unpack_req", _ptr(req), req.element_size(), ns, _ptr(ids), _ptr(wts), _stream())
    return ids, wts


def shard_msg_words(D, act_dtype):
    """(Dw, W): words of a looked-up row / of a message row [row | wide value, 0 | pad] for rows of act_dtype."""
    if act_dtype == torch.float32:
        Dw = D
    else:
        if D % 2:
            raise ValueError("16-bit message rows need an even D")
        Dw = D // 2
    if Dw % 4:
        raise ValueError("message rows need D % 4 == 0 (fp32) / D % 8 == 0 (16-bit)")
    return Dw, Dw + 4


def gather_rows_req(table, rows, rows_stride, wts, wts_stride, n_slots, wide_col, act_dtype, out=None):
    """The owner's answer to a request message: float32 [n_slots, W] rows [looked-up row (act_dtype) | w * weight, 0 | pad] in
    one pass over the fused rows; `rows` / `wts` are read with the given element strides (straight out of the request
    entries); slots whose row is outside the table (padding: -1) are left alone."""
    _need_cuda(table, rows, wts)
    V, D, ld = _table(table)
    Dw, W = shard_msg_words(D, act_dtype)
    msg = out if out is not None else torch.empty((max(n_slots, 1), W), dtype=torch.float32, device=table.device)[:n_slots]
    if msg.dtype != torch.float32 or tuple(msg.shape) != (n_slots, W) or not msg.is_contiguous():
        raise TypeError("gather_rows_req: out must be a contiguous float32 [n_slots, W] tensor")
    kind = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[act_dtype]
    ldo = W if kind == 0 else 2 * W
    _lib.call("mrec_gather_rows_wide_ex", _ptr(table), V, ld, D, _ptr(rows), rows.element_size(), int(rows_stride), n_slots, _ptr(wts),
              int(wts_stride), _ptr(msg), kind, ldo, int(wide_col), C.c_void_p(msg.data_ptr() + 4 * Dw), W, None, 0, 1, None, _stream())
    return msg


def shard_unroute_slots(back, slot_of_pos, D, act_dtype, out=None):
    """The returned message in position order: (emb [n, D] act_dtype, wprod float32 [n, 2])."""
    _need_cuda(back, slot_of_pos, out)
    Dw, W = shard_msg_words(D, act_dtype)
    if back.dtype != torch.float32 or back.dim() != 2 or back.shape[1] != W or not back.is_contiguous():
        raise TypeError("shard_unroute_slots: back must be the contiguous float32 [n_slots, W] message")
    n = slot_of_pos.numel()
    if out is None:
        out = torch.empty((max(n, 1), D), dtype=act_dtype, device=back.device)[:n]
    elif out.dtype != act_dtype or out.numel() != n * D or not out.is_contiguous():
        raise TypeError("shard_unroute_slots: out must be a contiguous [n, D] tensor of the row dtype")
    wprod = torch.empty((max(n, 1), 2), dtype=torch.float32, device=back.device)[:n]
    _lib.call("mrec_shard_unroute_slots", _ptr(back), W, _ptr(slot_of_pos), n, Dw, _ptr(out), _ptr(wprod), _stream())
    return out.view(n, D), wprod


def shard_route_grads(g, dlogit, F, pos_of_slot, out=None):
    """The gradient message float32 [n_slots, W]: [row gradient of the slot's position | dlogit of its sample | pad].  out: the
    message buffer to write (contiguous float32 [n_slots, W])."""
    _need_cuda(g, dlogit, pos_of_slot, out)
    n, D = g.shape
    Dw, W = shard_msg_words(D, g.dtype)
    if g.stride(1) != 1 or dlogit.dtype != torch.float32 or not dlogit.is_contiguous() or dlogit.numel() * F != n:
        raise TypeError("shard_route_grads: g [n, D] with unit column stride, dlogit float32 [n / F]")
    ldg = g.stride(0) * g.element_size() // 4 if n > 1 else Dw
    ns = pos_of_slot.numel()
    if out is not None:
        if out.dtype != torch.float32 or tuple(out.shape) != (ns, W) or not out.is_contiguous():
            raise TypeError("out must be contiguous float32 [n_slots, W]")
        msg = out
    else:
        msg = torch.empty((max(ns, 1), W), dtype=torch.float32, device=g.device)[:ns]
    _lib.call("mrec_shard_route_grads", _ptr(g), ldg, _ptr(dlogit), int(F), _ptr(pos_of_slot), ns, Dw, _ptr(msg), W, _stream())
    return msg


# ---- measurement hook ------------------------------------------------------------------------
class StepState:
    """mrec_step_state_t in device memory (include/mrec.h): the Adam bias-correction powers as the reference keeps them -- as
    state the optimizer's own graph advances (nn.Adam: beta1_power *= beta1) -- so that a captured step has constant arguments."""
    RING = 256
    _DT = np.dtype([("beta1_power", "<f4"), ("beta2_power", "<f4"), ("lr_t", "<f4"), ("r0", "<f4"), ("step", "<i8"), ("r1", "<u8"),
                    ("stamps", "<u8", (256, 2)), ("stamps_aux", "<u8", (256, 4)), ("stamps_end", "<u8", (256, 64))])

    def __init__(self, device, beta1_power=1.0, beta2_power=1.0, step=0):
        self.buf = torch.empty(self._DT.itemsize, dtype=torch.uint8, device=device)
        _need_cuda(self.buf)
        self.reset(beta1_power, beta2_power, step)
        khz = C.c_int32()
        _lib.call("mrec_wall_clock_khz", C.byref(khz))
        self.clock_khz = int(khz.value)

    def reset(self, beta1_power, beta2_power, step):
        _lib.call("mrec_step_state_init", _ptr(self.buf), float(beta1_power), float(beta2_power), int(step), _stream())

    def advance(self, lr, beta1, beta2):
        _lib.call("mrec_step_advance", _ptr(self.buf), float(lr), float(beta1), float(beta2), _stream())

    def set_stamps(self, on):
        """Kernel stamps on (the state's initial setting) or off for the steps enqueued after this call (a store to the state's
        `stamps_off` word on the current stream; captured graphs read the word when they run)."""
        self.buf[24:32].view(torch.int64).fill_(0 if on else 1)

    def read(self):
        """Host copy (synchronises): a numpy record with beta1_power, beta2_power, lr_t, step, stamps[256, 2]."""
        return self.buf.cpu().numpy().view(self._DT)[0]

    def apply_ms(self, steps):
        """Durations (ms) of the main sparse-apply kernel in the given step numbers, from the kernel's own stamps."""
        r = self.read()
        st, en = r["stamps"], r["stamps_end"]
        out = []
        for k in steps:
            a, b = int(st[k % self.RING][0]), int(en[k % self.RING].max())
            if b > a and a != 0xFFFFFFFFFFFFFFFF:
                out.append((b - a) / self.clock_khz)
        return out

    def embed_ms(self, steps):
        """Per step: (fused lookup kernel ms, sparse apply incl. its finishing kernel ms: begin of k_apply_main -> end of
        k_apply_long) from the kernels' own stamps -- the in-graph times of EmbeddingLookup + sparse apply."""
        r = self.read()
        st, aux, en = r["stamps"], r["stamps_aux"], r["stamps_end"]
        out = []
        for k in steps:
            a0, a1, l0, l1, e2 = (int(st[k % self.RING][0]), int(en[k % self.RING].max()), int(aux[k % self.RING][0]), int(aux[k % self.RING][1]),
                                  int(aux[k % self.RING][2]))
            if a1 > a0 and a0 != 0xFFFFFFFFFFFFFFFF and l1 > l0 and l0 != 0xFFFFFFFFFFFFFFFF and (e2 >= a1 or e2 == 0):
                # (e2 == 0: no workgroup of the finishing pass had a run to finish in this step -- the apply ended with k_apply_main)
                out.append(((l1 - l0) / self.clock_khz, (max(e2, a1) - a0) / self.clock_khz))
        return out


class KernelTimer:
    """Times exactly the main kernel of the next sparse-apply call (see include/mrec.h,
    mrec_profile_next_apply): arm() before the call, ms() afterwards (waits for the stop event)."""

    def __init__(self):
        self._a, self._b = C.c_void_p(), C.c_void_p()
        _lib.call("mrec_event_create", C.byref(self._a))
        _lib.call("mrec_event_create", C.byref(self._b))

    def arm(self):
        _lib.call("mrec_profile_next_apply", self._a, self._b)

    def ms(self):
        out = C.c_float()
        _lib.call("mrec_event_elapsed_ms", self._a, self._b, C.byref(out))
        return float(out.value)

    def __del__(self):
        for e in (getattr(self, "_a", None), getattr(self, "_b", None)):
            if e:
                _lib.lib().mrec_event_destroy(e)


# ---- elementwise ends of the bf16 dense net ----------------------------------------------------
def head_supported(K5):
    return K5 % 8 == 0 and (K5 // 8) & (K5 // 8 - 1) == 0 and K5 // 8 <= 64


def head_fwd_bwd(h4, w5, b5, wide, label, dscale, dw5_out, db4_out, db5_out, dh_scale=1.0):
    """Output layer + wide/deep add + sigmoid cross-entropy, forward and backward, one pass over h4.
    Returns (loss [1], logit [B], dlogit [B], dh4 [B, K5] of h4's dtype: bfloat16, float16 or float32)."""
    _need_cuda(h4, w5, b5, wide, label)
    B, K5 = h4.shape
    dev = h4.device
    logit = torch.empty(B, dtype=torch.float32, device=dev)
    dlogit = torch.empty(B, dtype=torch.float32, device=dev)
    dh4 = torch.empty_like(h4)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nb = _lib.query_bytes("mrec_head_workspace_bytes", B, K5)
    ws = workspace("head", nb, dev)
    if h4.dtype not in _DT16 and h4.dtype != torch.float32:
        raise TypeError("h4 must be bfloat16, float16 or float32")
    _lib.call("mrec_head_fwd_bwd_" + ("f32" if h4.dtype == torch.float32 else _DT16[h4.dtype]), _ptr(h4.contiguous()), _ptr(w5), _ptr(b5), _ptr(wide.contiguous()),
              _ptr(label.contiguous()), B, K5, float(dscale), float(dh_scale), _ptr(logit), _ptr(dlogit), _ptr(dh4), _ptr(dw5_out),
              _ptr(db4_out), _ptr(db5_out), _ptr(loss), _ptr(ws), ws.numel(), _stream())
    return loss, logit, dlogit, dh4


def head_fwd_bwd_wide(h4, w5, b5, wide_prod, wide_bias, label, dscale, dw5_out, db4_out, db5_out, dwide_bias_out=None, dh_scale=1.0):
    """head_fwd_bwd with the wide branch given as the per-field products of gather_rows_wide ([B, F]) + the wide bias:
    the ReduceSum over the fields (wide_and_deep.py:305-306) happens inside the head, in field order."""
    _need_cuda(h4, w5, b5, wide_prod, wide_bias, label)
    B, K5 = h4.shape
    dev = h4.device
    if (wide_prod.dtype != torch.float32 or wide_prod.dim() != 3 or wide_prod.shape[0] != B or wide_prod.shape[2] != 2
            or not wide_prod.is_contiguous()):
        raise TypeError("wide_prod must be the contiguous float32 [B, F, 2] tensor of gather_rows_wide")
    if h4.dtype not in _DT16:
        raise TypeError("h4 must be bfloat16 or float16")
    logit = torch.empty(B, dtype=torch.float32, device=dev)
    dlogit = torch.empty(B, dtype=torch.float32, device=dev)
    dh4 = torch.empty_like(h4)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nb = _lib.query_bytes("mrec_head_workspace_bytes", B, K5)
    ws = workspace("head", nb, dev)
    _lib.call("mrec_head_fwd_bwd_wide", int(h4.dtype == torch.float16), _ptr(h4.contiguous()), _ptr(w5), _ptr(b5), _ptr(wide_prod),
              wide_prod.shape[1], _ptr(wide_bias), _ptr(label.contiguous()), B, K5, float(dscale), float(dh_scale), _ptr(logit), _ptr(dlogit),
              _ptr(dh4), _ptr(dw5_out), _ptr(db4_out), _ptr(db5_out), _ptr(dwide_bias_out), _ptr(loss), _ptr(ws), ws.numel(), _stream())
    return loss, logit, dlogit, dh4


# ---- the tail of the dense net in one launch (csrc/mrec_tail.hip: k_tail) -------------------------------------------------
def tail_supported(B, K2, N2, N3):
    return bool(_lib.lib().mrec_tail_supported(int(B), int(K2), int(N2), int(N3)))


def tail_pack_weights(w2, w3, out=None):
    """The two tail weights w2 [K2, N2], w3 [N2, N3] (contiguous, 16-bit) in the tail kernel's operand order (both the forward and
    the backward operand of either: 2 * (K2*N2 + N2*N3) elements).  To be refreshed whenever the weights change."""
    _need_cuda(w2, w3, out)
    if w2.dim() != 2 or w3.dim() != 2 or w2.dtype not in _DT16 or w3.dtype != w2.dtype or not w2.is_contiguous() or not w3.is_contiguous() \
            or w3.shape[0] != w2.shape[1]:
        raise TypeError("tail_pack_weights: contiguous 16-bit matrices w2 [K2, N2], w3 [N2, N3]")
    n = C.c_int64(0)
    _lib.call("mrec_tail_packed_elems", w2.shape[0], w2.shape[1], w3.shape[1], C.byref(n))
    if out is None:
        out = torch.empty(int(n.value), dtype=w2.dtype, device=w2.device)
    if out.numel() != int(n.value) or out.dtype != w2.dtype or not out.is_contiguous():
        raise TypeError("tail_pack_weights: out must hold mrec_tail_packed_elems elements of the weights' dtype")
    _lib.call("mrec_tail_pack_weights", _ptr(w2), _ptr(w3), w2.shape[0], w2.shape[1], w3.shape[1], _ptr(out), _stream())
    return out


class _Transpose(C.Structure):      # mrec_transpose_t
    _fields_ = [("src", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("dst", C.c_void_p)]


def operand_copies(transposes=(), tail=None):
    """Every derived copy of the 16-bit weights in one launch: transposes = [(src [r, c], dst [c, r]), ...] (at most 4), tail =
    (w2, w3, packed) as for tail_pack_weights."""
    arr = (_Transpose * max(len(transposes), 1))()
    for k, (src, dst) in enumerate(transposes):
        _need_cuda(src, dst)
        if src.dim() != 2 or not src.is_contiguous() or src.dtype not in _DT16 or dst.shape != (src.shape[1], src.shape[0]) \
                or dst.dtype != src.dtype or not dst.is_contiguous():
            raise TypeError("operand_copies: (src [r, c], dst [c, r]) contiguous 16-bit matrices")
        arr[k] = _Transpose(src.data_ptr(), src.shape[0], src.shape[1], dst.data_ptr())
    w2 = w3 = packed = None
    K2 = N2 = N3 = 0
    if tail is not None:
        w2, w3, packed = tail
        _need_cuda(w2, w3, packed)
        K2, N2, N3 = w2.shape[0], w2.shape[1], w3.shape[1]
    _lib.call("mrec_dense_operand_copies", len(transposes), C.cast(arr, C.c_void_p), _ptr(w2), _ptr(w3), K2, N2, N3, _ptr(packed), _stream())


def tail_fwd_bwd(x, packed, b2, b3, w5, b5, wide, wide_bias, label, dscale, dw5_out, db4_out, db5_out, db3_out, db2_out,
                 dwide_bias_out=None, drop_in=None, out=None):
    """The last two hidden DenseLayers forward, the output head forward + backward and the input-gradient bprops back through
    both layers in ONE launch (64 samples per workgroup, intermediates in LDS).  packed: tail_pack_weights(w2, w3).  wide: the [B, F, 2] per-field products of
    gather_rows_wide (then wide_bias is needed) or the per-sample wide sum [B].  db3_out [N2], db2_out [K2]: bias gradients of the first
    tail layer and of the layer below it.  Returns (loss [1], dlogit [B], y2 [B, N2],
    dz4 [B, N3], dz3 [B, N2], dz2 [B, K2]); out: dict of preallocated tensors with those names (+ "logit")."""
    _need_cuda(x, packed, b2, b3, w5, b5, wide, label)
    B, K2, ldx = _mat16(x, "x")
    N2, N3 = b2.numel(), b3.numel()
    dt, dev = x.dtype, x.device
    if not tail_supported(B, K2, N2, N3):
        raise ValueError("tail_fwd_bwd: unsupported shape (see tail_supported)")
    if packed.dtype != dt or packed.numel() != 2 * (K2 * N2 + N2 * N3) or not packed.is_contiguous():
        raise TypeError("tail_fwd_bwd: packed must be tail_pack_weights(w2, w3) in x's dtype")
    for t in (b2, b3, w5, b5):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("tail_fwd_bwd: biases and the output layer's weight are contiguous float32")
    prod = wide.dim() == 3
    if prod and (wide.shape[0] != B or wide.shape[2] != 2 or wide.dtype != torch.float32 or not wide.is_contiguous() or wide_bias is None):
        raise TypeError("wide must be the contiguous float32 [B, F, 2] tensor of gather_rows_wide (with wide_bias) or a [B] tensor")
    o = out if out is not None else {}

    def buf(name, shape, dtype):
        t = o.get(name)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=dev)
            o[name] = t
        return t
    y2, dz4, dz3, dz2 = buf("y2", (B, N2), dt), buf("dz4", (B, N3), dt), buf("dz3", (B, N2), dt), buf("dz2", (B, K2), dt)
    logit, dlogit, loss = buf("logit", (B,), torch.float32), buf("dlogit", (B,), torch.float32), buf("loss", (1,), torch.float32)
    for t, n_ in ((db3_out, N2), (db2_out, K2), (dw5_out, N3), (db4_out, N3)):
        if t.numel() != n_ or t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("tail_fwd_bwd: dw5_out [N3], db4_out [N3], db3_out [N2], db2_out [K2] contiguous float32")
    nb = _lib.query_bytes("mrec_tail_workspace_bytes", B)
    ws = workspace("tail", nb, dev)
    _lib.call("mrec_tail_fwd_bwd", int(dt == torch.float16), _ptr(x), ldx, _ptr(packed), _ptr(b2), _ptr(b3),
              _ptr(w5), _ptr(b5), None if prod else _ptr(wide.contiguous()), _ptr(wide) if prod else None, wide.shape[1] if prod else 0,
              _ptr(wide_bias) if prod else None, _ptr(label.contiguous()), B, K2, N2, N3, float(dscale), _ptr(y2), _ptr(dz4), _ptr(dz3),
              _ptr(dz2), _ptr(logit), _ptr(dlogit), _ptr(dw5_out), _ptr(db4_out), _ptr(db5_out), _ptr(dwide_bias_out), _ptr(loss),
              _ptr(db3_out), _ptr(db2_out), _ptr(ws), ws.numel(), _drop_ref(drop_in), _stream())
    return loss, dlogit, y2, dz4, dz3, dz2


# ---- Dropout (csrc/mrec_dropout.h) ---------------------------------------------------------------------
class _DropDesc(C.Structure):      # mrec_dropout_t
    _fields_ = [("step_state", C.c_void_p), ("seed", C.c_uint64), ("step", C.c_int64), ("row0", C.c_int64), ("layer", C.c_int32),
                ("keep_prob", C.c_float)]


class Dropout:
    """Names the DenseLayer whose input is dropped out (wide_and_deep.py:98,117-118) and the mask: a pure function of
    (seed, step, layer, row0 + row, column).  step_state (ops.StepState): the step is read from device memory instead."""

    def __init__(self, keep_prob, seed, layer, step=0, row0=0, step_state=None):
        if not 0.0 < keep_prob <= 1.0:
            raise ValueError("keep_prob must be in (0, 1]")
        self.keep_prob, self.seed, self.layer, self.step, self.row0, self.step_state = float(keep_prob), int(seed), int(layer), int(step), int(row0), step_state
        self._c = _DropDesc(step_state.buf.data_ptr() if step_state is not None else None, self.seed & (2 ** 64 - 1), self.step, self.row0,
                            self.layer, self.keep_prob)

    @property
    def scale(self):
        return float(np.float32(1.0) / np.float32(self.keep_prob))


def _drop_ref(d):
    return C.cast(C.pointer(d._c), C.c_void_p) if d is not None else None


_KIND = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def dropout_(x, drop, out=None):
    """Dropout over a [M, W] matrix (fp32 / bf16 / f16), in place unless `out`; its own bprop (apply it to the gradient)."""
    _need_cuda(x, out)
    y = x if out is None else out
    if x.dim() != 2 or x.stride(1) != 1 or y.shape != x.shape or y.dtype != x.dtype or y.stride(1) != 1 or x.dtype not in _KIND:
        raise TypeError("dropout_: [M, W] fp32 / bf16 / f16 matrices with unit column stride")
    _lib.call("mrec_dropout", _ptr(x), x.stride(0), _ptr(y), y.stride(0), _KIND[x.dtype], x.shape[0], x.shape[1], _drop_ref(drop), _stream())
    return y


def dropout_mask(M, W, drop, device):
    """The mask as fp32 {0, 1 / keep_prob} [M, W]."""
    m = torch.empty((M, W), dtype=torch.float32, device=device)
    _need_cuda(m)
    _lib.call("mrec_dropout_mask_f32", _ptr(m), W, M, W, _drop_ref(drop), _stream())
    return m


# ---- DenseLayer on the matrix cores (csrc/mrec_dense.hip) --------------------------------------------
class _WGrad(C.Structure):         # mrec_wgrad_t
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("dy", C.c_void_p), ("lddy", C.c_int64), ("K", C.c_int32), ("N", C.c_int32),
                ("S", C.c_int32), ("reserved", C.c_int32), ("dw_slabs", C.c_void_p)]


def _mat16(t, name):
    if t.dtype not in _DT16 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name} must be a bfloat16 / float16 [rows, cols] tensor with unit column stride")
    return t.shape[0], t.shape[1], t.stride(0)


def dense_supported(M, K, N):
    """Shapes the MFMA DenseLayer kernels cover (forward and both bprops): widths multiples of 8 (16-byte rows)."""
    return K % 8 == 0 and N % 8 == 0


def dense_fwd(x, w, bias, relu=True, out=None, drop_next=None, wt=None):
    """DenseLayer.construct (wide_and_deep.py:113-133): act(x . w + bias).  x [M, K], w [K, N] 16-bit (same dtype),
    bias fp32 [N] or None.  Returns y [M, N] in x's dtype (fp32 accumulation, one rounding).  drop_next (Dropout): y is the
    input of the DenseLayer the descriptor names and leaves the kernel dropped out (:117-118).  wt: the transposed weight
    [N, K] (operand_copies keeps it current): the same result from the faster both-operands-K-contiguous kernel; w may then be None."""
    _need_cuda(x, w, bias, out)
    M, K, ldx = _mat16(x, "x")
    if wt is not None:
        _need_cuda(wt)
        N, K2, ldw = _mat16(wt, "wt")
        if K2 != K or ldw != K or wt.dtype != x.dtype:
            raise TypeError("wt must be a contiguous [N, K] tensor of x's dtype")
    else:
        K2, N, ldw = _mat16(w, "w")
        if K2 != K or ldw != N or w.dtype != x.dtype:
            raise TypeError("w must be a contiguous [K, N] tensor of x's dtype")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous()):
        raise TypeError("bias must be contiguous float32 [N]")
    y = out if out is not None else torch.empty((M, N), dtype=x.dtype, device=x.device)
    _, _, ldy = _mat16(y, "out")
    if y.shape != (M, N) or y.dtype != x.dtype:
        raise TypeError("out must be [M, N] of x's dtype")
    _lib.call(("mrec_dense_fwd_wt_" if wt is not None else "mrec_dense_fwd_") + _DT16[x.dtype], _ptr(x), ldx, _ptr(wt if wt is not None else w),
              _ptr(bias), M, K, N, int(bool(relu)), _ptr(y), ldy, _drop_ref(drop_next), _stream())
    return y


def dense_bwd_input(dy, w, h=None, db_out=None, out=None, db_slabs=None, drop_in=None):
    """MatMul bprop with respect to the input, fused with the ReLU + BiasAdd bprops of the layer below:
    dx = (dy . w^T) * (h > 0); db_out[:] = dx.sum(0).  dy [M, N], w [K, N], h [M, K] or None, db_out fp32 [K] or None.
    db_slabs (fp32 [ceil(M/256), K], instead of db_out): the bias gradient is left as per-tile-row partial sums for
    dense_adam_slabs_ / sum_slabs to add up (no finishing kernel on the backward chain)."""
    _need_cuda(dy, w, h, db_out, out)
    M, N, lddy = _mat16(dy, "dy")
    K, N2, ldw = _mat16(w, "w")
    if N2 != N or ldw != N or w.dtype != dy.dtype:
        raise TypeError("w must be a contiguous [K, N] tensor of dy's dtype")
    dx = out if out is not None else torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    _, _, lddx = _mat16(dx, "out")
    if h is not None and (h.shape != (M, K) or h.dtype != dy.dtype or h.stride(1) != 1 or h.stride(0) != lddx):
        raise TypeError("h must be [M, K] of dy's dtype with the row stride of the output")
    if db_out is not None and (db_out.dtype != torch.float32 or db_out.numel() != K or not db_out.is_contiguous()):
        raise TypeError("db_out must be contiguous float32 [K]")
    ws, nb = None, 0
    if db_slabs is not None:
        if db_out is not None:
            raise ValueError("pass db_out or db_slabs, not both")
        if db_slabs.dtype != torch.float32 or not db_slabs.is_contiguous() or db_slabs.shape != (dense_bwd_bias_slabs(M, K, N, False), K):
            raise TypeError("db_slabs must be contiguous float32 [dense_bwd_bias_slabs(M, K, N, fused=False), K]")
        ws = db_slabs.view(torch.uint8).view(-1)
    elif db_out is not None:
        nb = _lib.query_bytes("mrec_dense_bwd_input_workspace_bytes", M, K)
        ws = workspace("dense_bwd_input", nb, dy.device)
    _lib.call("mrec_dense_bwd_input_" + _DT16[dy.dtype], _ptr(dy), lddy, _ptr(w), _ptr(h), M, K, N, _ptr(dx), lddx,
              _ptr(db_out), _ptr(ws), ws.numel() if ws is not None else 0, _drop_ref(drop_in), _stream())
    return dx


def dense_bwd_bias_slabs(M, K, N, fused=True):
    """Rows of the bias-gradient slabs [T, K] dense_bwd (fused) / dense_bwd_input leave behind for this shape."""
    t = C.c_int32(0)
    _lib.call("mrec_dense_bwd_bias_slabs", M, K, N, int(bool(fused)), C.byref(t))
    return int(t.value)


def dense_bwd_weight_slabs(M, K, N):
    """Number of batch slabs mrec_dense_bwd_weight_* should use for this shape on this device."""
    s = C.c_int32(0)
    _lib.call("mrec_dense_bwd_weight_slabs", M, K, N, C.byref(s))
    return int(s.value)


def dense_bwd_weight(x, dy, out_slabs):
    """MatMul bprop with respect to the weight: out_slabs[s] = x[slab s]^T . dy[slab s], fp32 [S, K, N]."""
    _need_cuda(x, dy, out_slabs)
    M, K, ldx = _mat16(x, "x")
    M2, N, lddy = _mat16(dy, "dy")
    if M2 != M or dy.dtype != x.dtype:
        raise TypeError("x and dy must share the batch dimension and the dtype")
    if out_slabs.dtype != torch.float32 or out_slabs.dim() != 3 or out_slabs.shape[1:] != (K, N) or not out_slabs.is_contiguous():
        raise TypeError("out_slabs must be contiguous float32 [S, K, N]")
    _lib.call("mrec_dense_bwd_weight_" + _DT16[x.dtype], _ptr(x), ldx, _ptr(dy), lddy, M, K, N, out_slabs.shape[0],
              _ptr(out_slabs), _stream())
    return out_slabs


def dense_bwd(dy, w, x, dw_slabs, mask=True, db_slabs=None, out=None, drop_in=None, extra=None):
    """Both bprops of one DenseLayer in one launch: returns dx = (dy . w^T) [* (x > 0) when mask], fills dw_slabs
    [S, K, N] with x^T . dy by batch slab and db_slabs [ceil(M/256), K] (optional) with the bias-gradient partials of
    the layer below.  dy [M, N], w [K, N], x [M, K] (the layer's input = the activation of the layer below).  drop_in
    (Dropout): x went through Dropout on its way into this layer; dx is the gradient in front of it.  extra: up to two
    (x_e, dy_e, dw_slabs_e) weight-gradient problems of other layers over the same batch, computed by the same launch."""
    _need_cuda(dy, w, x, dw_slabs, db_slabs, out)
    M, N, lddy = _mat16(dy, "dy")
    K, N2, ldw = _mat16(w, "w")
    M2, K2, ldx = _mat16(x, "x")
    if N2 != N or ldw != N or K2 != K or M2 != M or w.dtype != dy.dtype or x.dtype != dy.dtype:
        raise TypeError("dense_bwd: dy [M, N], w [K, N] contiguous, x [M, K], one dtype")
    dx = out if out is not None else torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    _, _, lddx = _mat16(dx, "out")
    if mask and ldx != lddx:
        raise TypeError("dense_bwd: the mask source x must have the row stride of the output")
    if dw_slabs.dtype != torch.float32 or dw_slabs.dim() != 3 or dw_slabs.shape[1:] != (K, N) or not dw_slabs.is_contiguous():
        raise TypeError("dw_slabs must be contiguous float32 [S, K, N]")
    nb = 0
    if db_slabs is not None:
        if db_slabs.dtype != torch.float32 or not db_slabs.is_contiguous() or db_slabs.shape != (dense_bwd_bias_slabs(M, K, N, True), K):
            raise TypeError("db_slabs must be contiguous float32 [dense_bwd_bias_slabs(M, K, N), K]")
        nb = db_slabs.numel() * 4
    ex, nex = None, 0
    if extra:
        if len(extra) > 2:
            raise ValueError("dense_bwd: at most two extra weight-gradient problems")
        arr = (_WGrad * len(extra))()
        for e, (xe, dye, dwe) in enumerate(extra):
            _need_cuda(xe, dye, dwe)
            Me, Ke, ldxe = _mat16(xe, "extra x")
            Me2, Ne, lddye = _mat16(dye, "extra dy")
            if Me != M or Me2 != M or xe.dtype != dy.dtype or dye.dtype != dy.dtype:
                raise TypeError("dense_bwd: extra problems share the batch and the dtype")
            if dwe.dtype != torch.float32 or dwe.dim() != 3 or dwe.shape[1:] != (Ke, Ne) or not dwe.is_contiguous():
                raise TypeError("dense_bwd: extra dw_slabs must be contiguous float32 [S, K, N]")
            arr[e] = _WGrad(xe.data_ptr(), ldxe, dye.data_ptr(), lddye, Ke, Ne, dwe.shape[0], 0, dwe.data_ptr())
        ex, nex = C.cast(arr, C.c_void_p), len(extra)
    _lib.call("mrec_dense_bwd_" + _DT16[dy.dtype], _ptr(dy), lddy, _ptr(w), _ptr(x) if mask else None, _ptr(x), ldx, M, K, N,
              _ptr(dx), lddx, _ptr(db_slabs), nb, dw_slabs.shape[0], _ptr(dw_slabs), _drop_ref(drop_in), ex, nex, _stream())
    return dx


def sum_slabs(slabs, out):
    """out[...] = slabs.sum(0) in slab order (fp32): the weight gradient of dense_bwd_weight as one tensor."""
    _need_cuda(slabs, out)
    if slabs.dtype != torch.float32 or out.dtype != torch.float32 or not slabs.is_contiguous() or not out.is_contiguous():
        raise TypeError("sum_slabs needs contiguous float32 tensors")
    if out.numel() != slabs[0].numel():
        raise ValueError("out must have the size of one slab")
    _lib.call("mrec_dense_sum_slabs_f32", _ptr(slabs), slabs.shape[0], out.numel(), _ptr(out), _stream())
    return out


def sum_slab_segments_(g, slabs):
    """g[start : start + len] = tensor.sum(0) in slab order for every (start, tensor [S, ...]) of `slabs`, ONE launch."""
    _need_cuda(g)
    k = len(slabs)
    if k == 0:
        return
    if k > 16:
        raise ValueError("at most 16 slab segments")
    ptrs = (C.c_void_p * k)()
    starts = (C.c_int64 * k)()
    lens = (C.c_int64 * k)()
    splits = (C.c_int32 * k)()
    for q, (start, part) in enumerate(slabs):
        if part.dtype != torch.float32 or not part.is_contiguous():
            raise TypeError("slabs must be contiguous float32 [S, ...]")
        ptrs[q], starts[q], lens[q], splits[q] = part.data_ptr(), int(start), part[0].numel(), part.shape[0]
    _lib.call("mrec_dense_sum_slab_segments_f32", _ptr(g), g.numel(), k, C.cast(ptrs, C.c_void_p), C.cast(starts, C.c_void_p),
              C.cast(lens, C.c_void_p), C.cast(splits, C.c_void_p), _stream())


def dense_adam_slabs_(p, m, v, g, slabs, shadow16=None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, beta1_power=0.9,
                      beta2_power=0.999, grad_scale=1.0, use_nesterov=False, step_state=None, ftrl1=None, finish=None):
    """dense_adam_ whose gradient is, for some segments, still the fp32 batch slabs of dense_bwd_weight:
    slabs = [(start, tensor [S, ...] fp32)], start = element offset of the segment in the flat buffers; the slabs
    are added in slab order inside the Adam kernel.  shadow16 (bf16 / fp16, optional) receives the updated parameters."""
    _need_cuda(p, m, v, g, shadow16)
    n = p.numel()
    if n % 4:
        raise ValueError("dense_adam_slabs_ needs flat buffers padded to a multiple of 4 elements")
    k = len(slabs)
    if k > 16:
        raise ValueError("at most 16 slab segments")
    kind = 0
    if shadow16 is not None:
        if shadow16.dtype not in _DT16 or shadow16.numel() != n or not shadow16.is_contiguous():
            raise TypeError("shadow16 must be a contiguous bfloat16 / float16 tensor of the parameter's size")
        kind = 1 if shadow16.dtype == torch.bfloat16 else 2
    ptrs = (C.c_void_p * max(k, 1))()
    starts = (C.c_int64 * max(k, 1))()
    lens = (C.c_int64 * max(k, 1))()
    splits = (C.c_int32 * max(k, 1))()
    for q, (start, part) in enumerate(slabs):
        if part.dtype != torch.float32 or not part.is_contiguous():
            raise TypeError("slabs must be contiguous float32 [S, ...]")
        ptrs[q], starts[q], lens[q], splits[q] = part.data_ptr(), int(start), part[0].numel(), part.shape[0]
    f = _ftrl1(ftrl1) if ftrl1 is not None else None
    args = (_ptr(p), _ptr(m), _ptr(v), _ptr(g), _ptr(shadow16), kind, n, k,
            C.cast(ptrs, C.c_void_p), C.cast(starts, C.c_void_p), C.cast(lens, C.c_void_p), C.cast(splits, C.c_void_p),
            lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale, int(use_nesterov),
            _ptr(step_state.buf) if step_state is not None else None, C.cast(C.pointer(f), C.c_void_p) if f is not None else None)
    if finish is not None:        # + the finishing pass of a deferred sparse apply, as the first workgroups of this launch
        _lib.call("mrec_dense_adam_slabs_finish_f32", *args, C.cast(C.pointer(finish), C.c_void_p), _stream())
        finish.keep = None
        return
    _lib.call("mrec_dense_adam_slabs_one_ftrl_f32", *args, _stream())


# ---- DenseLayer in fp32 on the fp32-input matrix instruction (csrc/mrec_gemm_f32.hip) -----------------------------------
def _mat32(t, name):
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name} must be a float32 [rows, cols] tensor with unit column stride")
    return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])


def dense32_fwd(x, w, bias=None, relu=True, out=None):
    """DenseLayer.construct with convert_dtype=False (deep_and_cross.py:94-114): act(x . w + bias) in exact fp32."""
    _need_cuda(x, w, bias, out)
    M, K, ldx = _mat32(x, "x")
    K2, N, ldw = _mat32(w, "w")
    if K2 != K:
        raise TypeError("dense32_fwd: x [M, K], w [K, N]")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous()):
        raise TypeError("bias must be contiguous float32 [N]")
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x.device)
    M2, N2, ldy = _mat32(y, "out")
    if (M2, N2) != (M, N):
        raise TypeError("out must be [M, N]")
    _lib.call("mrec_dense32_fwd", _ptr(x), ldx, _ptr(w), ldw, _ptr(bias), M, K, N, int(bool(relu)), _ptr(y), ldy, _stream())
    return y


def dense32_colsum_tiles(M):
    return (int(M) + 63) // 64


def dense32_bwd_input(dy, w, h=None, out=None, colsum=None):
    """dx = (dy . w^T) masked by h > 0; colsum (optional fp32 [ceil(M / 64), K]): column sums of dx per 64 rows."""
    _need_cuda(dy, w, h, out, colsum)
    M, N, lddy = _mat32(dy, "dy")
    K, N2, ldw = _mat32(w, "w")
    if N2 != N:
        raise TypeError("dense32_bwd_input: dy [M, N], w [K, N]")
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy.device)
    M2, K2, lddx = _mat32(dx, "out")
    if (M2, K2) != (M, K):
        raise TypeError("out must be [M, K]")
    ldh = 0
    if h is not None:
        M3, K3, ldh = _mat32(h, "h")
        if (M3, K3) != (M, K):
            raise TypeError("h must be [M, K]")
    if colsum is not None and (colsum.dtype != torch.float32 or tuple(colsum.shape) != (dense32_colsum_tiles(M), K) or not colsum.is_contiguous()):
        raise TypeError("colsum must be contiguous float32 [ceil(M / 64), K]")
    _lib.call("mrec_dense32_bwd_input", _ptr(dy), lddy, _ptr(w), ldw, _ptr(h), ldh, M, K, N, _ptr(dx), lddx, _ptr(colsum), _stream())
    return dx


def dense32_bwd_weight_slabs(M, K, N):
    s = C.c_int32(0)
    _lib.call("mrec_dense32_bwd_weight_slabs", M, K, N, C.byref(s))
    return int(s.value)


def dense32_bwd_weight(x, dy, out_slabs):
    """out_slabs[s] = x[slab s]^T . dy[slab s], fp32 [S, K, N]."""
    _need_cuda(x, dy, out_slabs)
    M, K, ldx = _mat32(x, "x")
    M2, N, lddy = _mat32(dy, "dy")
    if M2 != M or out_slabs.dtype != torch.float32 or out_slabs.dim() != 3 or tuple(out_slabs.shape[1:]) != (K, N) or not out_slabs.is_contiguous():
        raise TypeError("dense32_bwd_weight: x [M, K], dy [M, N], out_slabs contiguous float32 [S, K, N]")
    _lib.call("mrec_dense32_bwd_weight", _ptr(x), ldx, _ptr(dy), lddy, M, K, N, out_slabs.shape[0], _ptr(out_slabs), _stream())
    return out_slabs


# ---- fp32 DenseLayers at the 16-bit matrix rate: three-part bf16 operands (csrc/mrec_gemm_x3.hip) -----------------------------------
def _up64(x):
    return (int(x) + 63) // 64 * 64


def x3_supported(M, K, N):
    """Shapes the split path takes: everything an fp32 DenseLayer of realistic size has; tiny problems stay on dense32_*."""
    return (M >= 256 and K >= 64 and N >= 64 and K % 2 == 0 and N % 2 == 0          # (rows of the fp32 operands 8-byte aligned)
            and 3 * _up64(M) * max(_up64(K), _up64(N)) * 2 < 2 ** 31 and 3 * _up64(K) * _up64(N) * 2 < 2 ** 31)


def x3_parts(rows, cols, device):
    """A parts image [3, Rp, Cp] (zeroed: the fused output ends of x3_fwd / x3_dgrad never write its padding)."""
    return torch.zeros((3, _up64(rows), _up64(cols)), dtype=torch.bfloat16, device=device)


def x3_wgrad_slabs(M, K, N):
    """Batch slabs the weight gradient (x3_gemm form 2) runs fastest with."""
    s = C.c_int32(0)
    _lib.call("mrec_x3_wgrad_slabs", int(M), int(K), int(N), C.byref(s))
    return int(s.value)


def x3_slabs(M, S):
    """The largest slab count <= S the batch-reduction form (x3_gemm form 2) takes for M rows: every slab needs a non-empty share of
    the 6 * ceil(M / 64) reduction tiles."""
    T = 6 * (_up64(M) // 64)
    S = max(1, min(int(S), T))
    while S > 1 and -(-T // S) * (S - 1) >= T:
        S -= 1
    return S


def x3_split(x, out=None):
    """fp32 [R, C] -> its parts image bf16 [3][Rp][Cp] (x = x1 + x2 + x3, zero padded to multiples of 64)."""
    _need_cuda(x, out)
    R, Cc, ld = _mat32(x, "x")
    p = out if out is not None else x3_parts(R, Cc, x.device)
    if tuple(p.shape) != (3, _up64(R), _up64(Cc)) or p.dtype != torch.bfloat16 or not p.is_contiguous():
        raise TypeError("x3_split: out must be the contiguous bfloat16 parts image [3, Rp, Cp]")
    _lib.call("mrec_x3_split", _ptr(x), ld, R, Cc, _ptr(p), _stream())
    return p


def x3_gemm(form, P, Q, M, K, N, out, S=1):
    """form 0: out [M, N] = x . w; 1: out [M, K] = dy . w^T; 2: out [S, K, N] = x^T . dy in S batch slabs (parts images in)."""
    _need_cuda(P, Q, out)
    if out.dtype != torch.float32 or out.stride(-1) != 1:
        raise TypeError("x3_gemm: out must be float32 with unit column stride")
    ldc = out.stride(-2)
    _lib.call("mrec_x3_gemm", int(form), _ptr(P), _ptr(Q), int(M), int(K), int(N), _ptr(out), int(ldc), int(S), _stream())
    return out


def _x3_out(out, M, C, name):
    if out.dtype != torch.float32 or out.stride(-1) != 1 or tuple(out.shape) != (M, C):
        raise TypeError(f"{name}: out must be float32 [{M}, {C}] with unit column stride")
    return out.stride(0)


def _x3_img(parts, M, C, name):
    if parts is not None and (tuple(parts.shape) != (3, _up64(M), _up64(C)) or parts.dtype != torch.bfloat16 or not parts.is_contiguous()):
        raise TypeError(f"{name}: parts_out must be the contiguous bfloat16 parts image [3, {_up64(M)}, {_up64(C)}]")


def x3_fwd(xP, wP, M, K, N, out, bias=None, relu=True, parts_out=None, drop_next=None):
    """DenseLayer forward on parts images, output end in the GEMM's epilogue: out = relu?(x . w + bias) -- dropped out as the next
    layer's input when drop_next (ops.Dropout) -- and, optionally, out's parts."""
    _need_cuda(xP, wP, out, bias, parts_out)
    ld = _x3_out(out, M, N, "x3_fwd")
    _x3_img(parts_out, M, N, "x3_fwd")
    _lib.call("mrec_x3_gemm_fwd", _ptr(xP), _ptr(wP), int(M), int(K), int(N), _ptr(out), ld, _ptr(bias), int(bool(relu)),
              _drop_ref(drop_next), _ptr(parts_out), _stream())
    return out


def x3_dgrad(dyP, wP, M, K, N, out, h=None, scale=1.0, colsum=None, parts_out=None):
    """Input gradient on parts images, output end in the epilogue: out = (h > 0 ? dy . w^T : 0) * scale, colsum [ceil(M / 64), K] the
    bias gradient of the layer below in 64-row partial sums, and, optionally, out's parts."""
    _need_cuda(dyP, wP, out, h, colsum, parts_out)
    ld = _x3_out(out, M, K, "x3_dgrad")
    _x3_img(parts_out, M, K, "x3_dgrad")
    ldh = 0
    if h is not None:
        M2, K2, ldh = _mat32(h, "h")
        if (M2, K2) != (M, K):
            raise TypeError("h must be [M, K]")
    if colsum is not None and (colsum.dtype != torch.float32 or tuple(colsum.shape) != ((M + 63) // 64, K) or not colsum.is_contiguous()):
        raise TypeError("colsum must be contiguous float32 [ceil(M / 64), K]")
    ws, nb = None, 0
    if h is None and colsum is None and parts_out is None and scale == 1.0:
        nb = _lib.query_bytes("mrec_x3_gemm_dgrad_workspace_bytes", int(M), int(K), int(N))
        ws = workspace("x3_dgrad", nb, out.device) if nb else None
    _lib.call("mrec_x3_gemm_dgrad", _ptr(dyP), _ptr(wP), int(M), int(K), int(N), _ptr(out), ld, _ptr(h), ldh, float(scale), _ptr(colsum),
              _ptr(parts_out), _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return out


def x3_bias_relu_(acc, bias, relu=True, parts_out=None):
    _need_cuda(acc, bias, parts_out)
    M, N, ld = _mat32(acc, "acc")
    _lib.call("mrec_x3_bias_relu", _ptr(acc), ld, M, N, _ptr(bias), int(bool(relu)), _ptr(parts_out), _stream())
    return acc


def x3_mask_colsum_(acc, h=None, colsum=None, parts_out=None, scale=1.0):
    _need_cuda(acc, h, colsum, parts_out)
    M, K, ld = _mat32(acc, "acc")
    ldh = 0
    if h is not None:
        M2, K2, ldh = _mat32(h, "h")
        if (M2, K2) != (M, K):
            raise TypeError("h must be [M, K]")
    if colsum is not None and (colsum.dtype != torch.float32 or tuple(colsum.shape) != ((M + 63) // 64, K) or not colsum.is_contiguous()):
        raise TypeError("colsum must be contiguous float32 [ceil(M / 64), K]")
    _lib.call("mrec_x3_mask_colsum", _ptr(acc), ld, M, K, _ptr(h), ldh, float(scale), _ptr(colsum), _ptr(parts_out), _stream())
    return acc


def dcn_head_supported(H, X):
    return H % 4 == 0 and H <= 1024 and X % 2 == 0 and X <= 1280


def dcn_head_fwd_bwd(d2, c, w3, b3, label, dscale, dw3_out, db2_out, db3_out, out=None):
    """Deep&Cross output layer over [d2 | c] without the concat + sigmoid cross-entropy + every bprop that hangs off the logit, one
    pass (csrc/mrec_dcn.hip).  Returns (loss [1], logit [B], dd2 [B, H], dc [B, X]); dw3_out [H + X], db2_out [H], db3_out [1]
    receive the batch sums.  out: dict of preallocated "loss", "logit", "dd2", "dc"."""
    _need_cuda(d2, c, w3, b3, label, dw3_out, db2_out, db3_out)
    B, H, ldd = _mat32(d2, "d2")
    B2, X, ldc = _mat32(c, "c")
    if B2 != B or w3.numel() != H + X or not w3.is_contiguous() or w3.dtype != torch.float32 or label.numel() != B:
        raise TypeError("dcn_head_fwd_bwd: d2 [B, H], c [B, X], w3 [H + X], label [B]")
    o = out if out is not None else {}
    dev = d2.device

    def buf(name, shape):
        t = o.get(name)
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
            o[name] = t
        return t
    loss, logit, dd2, dc = buf("loss", (1,)), buf("logit", (B,)), buf("dd2", (B, H)), buf("dc", (B, X))
    nb = _lib.query_bytes("mrec_dcn_head_workspace_bytes", B, H, X)
    ws = workspace("dcn_head", nb, dev)
    _lib.call("mrec_dcn_head_fwd_bwd", _ptr(d2), ldd, _ptr(c), ldc, _ptr(w3), _ptr(b3), _ptr(label.contiguous()), B, H, X, float(dscale),
              _ptr(logit), _ptr(dd2), dd2.stride(0), _ptr(dc), dc.stride(0), _ptr(dw3_out), _ptr(db2_out), _ptr(db3_out), _ptr(loss), _ptr(ws),
              ws.numel(), _stream())
    return loss, logit, dd2, dc


# ---- DeepFM second-order term ------------------------------------------------------------------
def fm_forward(vx, add=None, out16=None):
    """vx [B, F, D] fp32 -> (fm_out [B], colsum [B, D]) (deepfm.py:221-228).  add [B] (the linear term): fm_out = add + fm.
    out16 (bfloat16 / float16 [B, F, D], D % 4 == 0): receives vx rounded to 16 bits in the same pass (the dense net's input)."""
    _need_cuda(vx, add, out16)
    B, F_, D = vx.shape
    vx = vx.contiguous()
    if add is not None and (add.dtype != torch.float32 or add.numel() != B or not add.is_contiguous()):
        raise TypeError("fm_forward: add must be contiguous float32 [B]")
    if out16 is not None and (out16.dtype not in _DT16 or out16.numel() != vx.numel() or not out16.is_contiguous() or D % 4):
        raise TypeError("fm_forward: out16 must be a contiguous bfloat16 / float16 tensor of vx's size (D % 4 == 0)")
    fm = torch.empty(B, dtype=torch.float32, device=vx.device)
    cs = torch.empty((B, D), dtype=torch.float32, device=vx.device)
    _lib.call("mrec_fm_fwd_add16_f32", _ptr(vx), B, F_, D, _ptr(add), _ptr(fm), _ptr(cs), _ptr(out16),
              0 if out16 is None else (1 if out16.dtype == torch.bfloat16 else 2), _stream())
    return fm, cs


def fm_backward_mix(g16, vx, colsum, dout):
    """fp32 [B, F, D]: widen(g16) + dout[b] * (colsum[b, d] - vx[b, f, d]) -- the FM gradient added to the 16-bit input gradient
    of the mixed-precision MLP, one pass."""
    _need_cuda(g16, vx, colsum, dout)
    B, F_, D = vx.shape
    if g16.dtype not in _DT16 or g16.numel() != vx.numel() or not g16.is_contiguous():
        raise TypeError("fm_backward_mix: g16 must be a contiguous bfloat16 / float16 tensor of vx's size")
    out = torch.empty_like(vx)
    _lib.call("mrec_fm_bwd_mix_f32", _ptr(vx.contiguous()), _ptr(colsum), _ptr(dout.contiguous()), _ptr(g16), 1 if g16.dtype == torch.bfloat16 else 2,
              B, F_, D, _ptr(out), _stream())
    return out


def fm_backward_(g, vx, colsum, dout):
    """g[b,f,d] += dout[b] * (colsum[b,d] - vx[b,f,d]) in place."""
    _need_cuda(g, vx, colsum, dout)
    B, F_, D = vx.shape
    _lib.call("mrec_fm_bwd_f32", _ptr(vx.contiguous()), _ptr(colsum), _ptr(dout.contiguous()), B, F_, D, _ptr(g), _stream())
    return g
