"""numpy/ctypes front-end of the CPU oracle (oracle/mrec_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED at the MindSpore boundary -- see the header of mrec_oracle.c.

Every function takes and returns numpy arrays; tables are modified in place where the reference
primitive mutates its Parameter (optimizers, MapParameter ops).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MREC_ORACLE_SANITIZE=1 (tools/sanitize_oracle.sh): the AddressSanitizer + UBSan build of the same source
_SAN = os.environ.get("MREC_ORACLE_SANITIZE", "") not in ("", "0")
_TARGET = "libmrec_oracle_san.so" if _SAN else "libmrec_oracle.so"
_SO = os.path.join(_HERE, _TARGET)


def build(force=False):
    src = os.path.join(_HERE, "mrec_oracle.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", _TARGET])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.mrec_o_unique_i64.restype = C.c_int64
        _lib.mrec_o_unique_i32.restype = C.c_int64
        _lib.mrec_o_map_create.restype = C.c_void_p
        _lib.mrec_o_map_size.restype = C.c_int64
        _lib.mrec_o_map_export.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def normal_rows(seed, rows, D, sigma):
    rows = _i64(rows).ravel()
    out = np.empty((rows.size, D), np.float32)
    lib().mrec_o_normal_rows_f32(C.c_uint64(seed), _p(rows), C.c_int64(rows.size), C.c_int32(D),
                                 C.c_float(sigma), _p(out))
    return out


def fill_normal(seed, nrows, D, sigma, row0=0):
    out = np.empty((nrows, D), np.float32)
    lib().mrec_o_fill_normal_f32(C.c_uint64(seed), C.c_int64(row0), C.c_int64(nrows), C.c_int32(D),
                                 C.c_int64(D), C.c_float(sigma), _p(out))
    return out


def unique(x):
    """ops.Unique: (y, idx) with first-occurrence order."""
    x = np.ascontiguousarray(x).ravel()
    n = x.size
    inv = np.empty(n, np.int32)
    if x.dtype == np.int32:
        uniq = np.empty(n, np.int32)
        U = lib().mrec_o_unique_i32(_p(x), C.c_int64(n), _p(uniq), _p(inv))
    else:
        x = _i64(x)
        uniq = np.empty(n, np.int64)
        U = lib().mrec_o_unique_i64(_p(x), C.c_int64(n), _p(uniq), _p(inv))
    return uniq[:U].copy(), inv


def gather_rows(table, ids, row_scale=None, threads=0):
    table = np.asarray(table)
    assert table.dtype == np.float32 and table.ndim == 2 and table.strides[1] == 4
    V, D = table.shape
    ld = table.strides[0] // 4
    ids_f = _i64(ids).ravel()
    out = np.empty((ids_f.size, D), np.float32)
    rs = _f32(row_scale).ravel() if row_scale is not None else None
    args = (_p(table), C.c_int64(V), C.c_int64(ld), C.c_int32(D), _p(ids_f), C.c_int64(ids_f.size), _p(rs), _p(out))
    if threads > 0:
        lib().mrec_o_gather_rows_f32_mt(*args, C.c_int(threads))
    else:
        lib().mrec_o_gather_rows_f32(*args)
    return out.reshape(tuple(np.shape(ids)) + (D,))


def wide_sum(w, ids, wts, bias, threads=0):
    w = _f32(w).ravel()
    ids2 = _i64(ids)
    B, F = ids2.shape
    wts = _f32(wts)
    out = np.empty(B, np.float32)
    args = (_p(w), C.c_int64(w.size), _p(ids2), _p(wts), C.c_int64(B), C.c_int32(F), C.c_float(bias), _p(out))
    if threads > 0:
        lib().mrec_o_wide_sum_f32_mt(*args, C.c_int(threads))
    else:
        lib().mrec_o_wide_sum_f32(*args)
    return out


def segment_sum(vals, seg, U):
    vals = _f32(vals)
    n, D = vals.shape
    seg = np.ascontiguousarray(seg, np.int32)
    out = np.empty((U, D), np.float32)
    lib().mrec_o_segment_sum_f32(_p(vals), C.c_int64(D), _p(seg), C.c_int64(n), C.c_int32(D), _p(out),
                                 C.c_int64(U))
    return out


def _tab(t):
    assert t.dtype == np.float32 and t.ndim == 2 and t.strides[1] == 4
    return t.shape[0], t.shape[1], t.strides[0] // 4


def sparse_lazy_adam(p, m, v, ids, g, row_scale=None, lr=3.5e-4, b1=0.9, b2=0.999, eps=1e-8,
                     b1_pow=0.9, b2_pow=0.999, grad_scale=1.0, nesterov=False, threads=0):
    """threads > 0 runs the *_mt entry (same bits, unique ids partitioned over host threads)."""
    V, D, ld = _tab(p)
    ids_f = _i64(ids).ravel()
    g = _f32(g).reshape(ids_f.size, D)
    rs = _f32(row_scale).ravel() if row_scale is not None else None
    args = (_p(p), _p(m), _p(v), C.c_int64(V), C.c_int64(ld), C.c_int32(D), _p(ids_f), C.c_int64(ids_f.size),
            _p(g), C.c_int64(D), _p(rs), C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps),
            C.c_float(b1_pow), C.c_float(b2_pow), C.c_float(grad_scale), C.c_int(int(nesterov)))
    if threads > 0:
        lib().mrec_o_sparse_lazy_adam_f32_mt(*args, C.c_int(threads))
    else:
        lib().mrec_o_sparse_lazy_adam_f32(*args)


def sparse_ftrl(var, accum, linear, ids, g, row_scale=None, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5,
                grad_scale=1.0, threads=0):
    V, D, ld = _tab(var)
    ids_f = _i64(ids).ravel()
    g = _f32(g).reshape(ids_f.size, D)
    rs = _f32(row_scale).ravel() if row_scale is not None else None
    args = (_p(var), _p(accum), _p(linear), C.c_int64(V), C.c_int64(ld), C.c_int32(D), _p(ids_f),
            C.c_int64(ids_f.size), _p(g), C.c_int64(D), _p(rs), C.c_float(lr), C.c_float(l1), C.c_float(l2),
            C.c_float(lr_power), C.c_float(grad_scale))
    if threads > 0:
        lib().mrec_o_sparse_ftrl_f32_mt(*args, C.c_int(threads))
    else:
        lib().mrec_o_sparse_ftrl_f32(*args)


def dense_adam(p, m, v, g, lr=3.5e-4, b1=0.9, b2=0.999, eps=1e-8, b1_pow=0.9, b2_pow=0.999,
               grad_scale=1.0, nesterov=False):
    g = _f32(g)
    assert p.flags.c_contiguous and m.flags.c_contiguous and v.flags.c_contiguous
    lib().mrec_o_dense_adam_f32(_p(p), _p(m), _p(v), _p(g), C.c_int64(p.size), C.c_float(lr), C.c_float(b1),
                                C.c_float(b2), C.c_float(eps), C.c_float(b1_pow), C.c_float(b2_pow),
                                C.c_float(grad_scale), C.c_int(int(nesterov)))


def dense_ftrl(var, accum, linear, g, lr=5e-2, l1=1e-8, l2=1e-8, lr_power=-0.5, grad_scale=1.0):
    g = _f32(g)
    lib().mrec_o_dense_ftrl_f32(_p(var), _p(accum), _p(linear), _p(g), C.c_int64(var.size), C.c_float(lr),
                                C.c_float(l1), C.c_float(l2), C.c_float(lr_power), C.c_float(grad_scale))


class Map:
    """MapParameter restatement (README.md:160-205; embedding.py:136-146)."""

    def __init__(self, D, capacity, seed=0, sigma=0.01, fill=None):
        self.D = D
        self.capacity = capacity
        s = -1.0 if fill is not None else sigma
        self._h = C.c_void_p(lib().mrec_o_map_create(C.c_int32(D), C.c_int64(capacity), C.c_uint64(seed),
                                                     C.c_float(s), C.c_float(fill or 0.0)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mrec_o_map_destroy(self._h)
            self._h = None

    def find_or_insert(self, keys, insert_default=True):
        keys = _i64(keys).ravel()
        rows = np.empty(keys.size, np.int32)
        lib().mrec_o_map_find_or_insert(self._h, _p(keys), C.c_int64(keys.size), C.c_int(int(insert_default)),
                                        _p(rows))
        return rows

    def get(self, keys, insert_default=True):
        keys = _i64(keys).ravel()
        out = np.empty((keys.size, self.D), np.float32)
        lib().mrec_o_map_get(self._h, _p(keys), C.c_int64(keys.size), C.c_int(int(insert_default)), _p(out))
        return out

    def put(self, keys, vals):
        keys = _i64(keys).ravel()
        vals = _f32(vals).reshape(keys.size, self.D)
        lib().mrec_o_map_put(self._h, _p(keys), C.c_int64(keys.size), _p(vals))

    def erase(self, keys):
        keys = _i64(keys).ravel()
        lib().mrec_o_map_erase(self._h, _p(keys), C.c_int64(keys.size))

    def size(self):
        return int(lib().mrec_o_map_size(self._h))

    def export(self):
        n = self.size()
        keys = np.empty(n, np.int64)
        vals = np.empty((n, self.D), np.float32)
        lib().mrec_o_map_export(self._h, _p(keys), _p(vals))
        return keys, vals


def cross_layers(x0, w, b):
    x0 = _f32(x0); w = _f32(w); b = _f32(b)
    B, D = x0.shape
    L = w.shape[0]
    out = np.empty_like(x0)
    lib().mrec_o_cross_layers_f32(_p(x0), _p(w), _p(b), C.c_int32(L), C.c_int64(B), C.c_int32(D), _p(out), None)
    return out


def cross_layers_bwd(x0, w, b, dy):
    x0 = _f32(x0); w = _f32(w); b = _f32(b); dy = _f32(dy)
    B, D = x0.shape
    L = w.shape[0]
    dx0 = np.empty_like(x0); dw = np.empty_like(w); db = np.empty_like(b)
    lib().mrec_o_cross_layers_bwd_f32(_p(x0), _p(w), _p(b), C.c_int32(L), C.c_int64(B), C.c_int32(D), _p(dy),
                                      _p(dx0), _p(dw), _p(db))
    return dx0, dw, db


def shard_route(ids, n_shards):
    ids_f = _i64(ids).ravel()
    n = ids_f.size
    send_local = np.empty(n, np.int64)
    perm = np.empty(n, np.int32)
    counts = np.empty(n_shards, np.int64)
    lib().mrec_o_shard_route_i64(_p(ids_f), C.c_int64(n), C.c_int32(n_shards), _p(send_local), _p(perm), _p(counts))
    return send_local, perm, counts


# ---- fixed-capacity routing of a sharded step (numpy restatements of include/mrec.h's mrec_shard_route_slots_* etc.;
# ---- reference semantics: owner = id mod n / hash(key) mod n, hybrid parallel, README.md:140-144) ------------------------
def shard_owner(ids, n_shards, hashed=False):
    ids = _i64(ids).ravel()
    if hashed:
        return ((_mix64(ids.astype(np.uint64)) >> np.uint64(33)) % np.uint64(n_shards)).astype(np.int64)
    return np.mod(ids, n_shards)


def shard_capacity(n, n_shards, factor=1.25):
    if n_shards <= 1:
        return int(n)
    c = -(-int(n * factor) // n_shards)
    return int(min(n, -(-c // 64) * 64))


def shard_route_slots(ids, wts, n_shards, cap, hashed=False, rot=0):
    """Returns (req_ids int64 [n_shards * cap] (-1: unused slot), req_wts float32, slot_of_pos int32 [n] (-1: bucket full),
    pos_of_slot int32 [n_shards * cap] (-1), dropped): position i takes slot chunk * cap + (its rank among the positions of the
    same owner, ascending position), chunk = (owner - rot) mod n_shards."""
    ids = _i64(ids).ravel()
    n = ids.size
    w = np.ones(n, np.float32) if wts is None else _f32(wts).ravel()
    own = shard_owner(ids, n_shards, hashed)
    req_ids = np.full(n_shards * cap, -1, np.int64)
    req_wts = np.zeros(n_shards * cap, np.float32)
    slot_of_pos = np.full(n, -1, np.int32)
    pos_of_slot = np.full(n_shards * cap, -1, np.int32)
    fill = np.zeros(n_shards, np.int64)
    dropped = 0
    for i in range(n):
        o = int(own[i])
        j = int(fill[o])
        fill[o] += 1
        if j >= cap:
            dropped += 1
            continue
        s = ((o - rot) % n_shards) * cap + j
        req_ids[s] = ids[i] if hashed else (ids[i] - o) // n_shards
        req_wts[s] = w[i]
        slot_of_pos[i] = s
        pos_of_slot[s] = i
    return req_ids, req_wts, slot_of_pos, pos_of_slot, dropped


def unique_skip_negative(ids):
    """ops.Unique over the non-negative ids only: (uniq, inv) with inv = -1 at the negative (padding) positions."""
    ids = _i64(ids).ravel()
    ok = ids >= 0
    u, inv_ok = unique(ids[ok])
    inv = np.full(ids.size, -1, np.int64)
    inv[ok] = inv_ok
    return u, inv


# ---- elementwise ends of the dense net (numpy restatements; reference: wide_and_deep.py:113-133,315,352-354)
def relu_bwd_colsum(g, h):
    """ReLU bprop + BiasAdd bprop: dh = g where h > 0 else 0; db = dh.sum(0) (float64 accumulate)."""
    g = np.asarray(g, np.float32); h = np.asarray(h, np.float32)
    dh = np.where(h > 0, g, np.float32(0))
    return dh, dh.astype(np.float64).sum(axis=0)


def head_fwd_bwd(h4, w5, b5, wide, label, dscale, dh_scale=1.0):
    """dense_layer_5 (K5 -> 1) + wide/deep add + SigmoidCrossEntropyWithLogits/ReduceMean, forward and
    backward.  Returns dict(loss, logit, dlogit, dh4, dw5, db4, db5) in float64 where reduced.  dh_scale: 1 / keep_prob
    when h4 went through Dropout (its zeros then carry the mask as well as the ReLU)."""
    h4 = np.asarray(h4, np.float64); w5 = np.asarray(w5, np.float64).ravel()
    z = h4 @ w5 + float(b5) + np.asarray(wide, np.float64)
    y = np.asarray(label, np.float64).ravel()
    loss = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    dl = (1.0 / (1.0 + np.exp(-z)) - y) * dscale
    dh4 = np.where(h4 > 0, dl[:, None] * w5[None, :] * float(np.float32(dh_scale)), 0.0)
    return dict(loss=loss.mean(), logit=z, dlogit=dl, dh4=dh4, dw5=h4.T @ dl, db4=dh4.sum(axis=0), db5=dl.sum())


def fm_forward(vx):
    """DeepFM second-order term (deepfm.py:221-228), sequential float32 sums over the field axis."""
    vx = np.asarray(vx, np.float32)
    B, F, D = vx.shape
    s = np.zeros((B, D), np.float32); q = np.zeros((B, D), np.float32)
    for f in range(F):
        s = s + vx[:, f, :]
        q = q + vx[:, f, :] * vx[:, f, :]
    return (0.5 * (s.astype(np.float64) ** 2 - q).sum(axis=1)), s


def fm_backward(vx, colsum, dout):
    return np.asarray(dout, np.float64)[:, None, None] * (np.asarray(colsum, np.float64)[:, None, :] - np.asarray(vx, np.float64))


# ---- DenseLayer in mixed precision (numpy restatements) ---------------------------------------------------
# Reference: DenseLayer.construct, models/wide_deep/src/wide_and_deep.py:113-133 -- with use_mixed_precision the input
# and the weight are cast to float16, MatMul runs on them, BiasAdd and ReLU follow.  Restated here with the rounding
# points of the MI355X path: 16-bit operands (float16 as in the reference, or bfloat16), products summed exactly
# (float64 here; fp32 on the matrix cores), fp32 bias added, ONE rounding of the result to 16 bits.
def round16(x, dtype):
    """Round float32 values to dtype ('bf16' or 'f16'), round-to-nearest-even; returns float32 holding those values."""
    x = np.ascontiguousarray(x, np.float32)
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    # (u + 0x7FFF + lsb) >> 16 << 16 in uint32: the sum wraps only for negative NaN payloads, which are fixed up below
    u = x.view(np.uint32)
    with np.errstate(over="ignore"):
        r = (u >> np.uint32(16)) & np.uint32(1)
        r += np.uint32(0x7FFF)
        r += u
        r &= np.uint32(0xFFFF0000)
    out = r.view(np.float32)
    nan = np.isnan(x)
    out[nan] = np.float32("nan")
    return out.reshape(x.shape)


def dense_layer(x16, w16, bias, relu, dtype):
    """y = round16(act(x . w + b)); x16 [M, K], w16 [K, N] hold 16-bit values (as float32), bias float32 [N]."""
    acc = np.asarray(x16, np.float64) @ np.asarray(w16, np.float64)
    if bias is not None:
        acc = acc + np.asarray(bias, np.float64)[None, :]
    if relu:
        acc = np.maximum(acc, 0.0)
    return round16(acc.astype(np.float32), dtype)


def dense_bwd_input(dy16, w16, h16, dtype, scale=1.0, mask=None):
    """MatMul bprop wrt the input, then the ReLU bprop of the layer below and its BiasAdd bprop:
    dx = round16(dy . w^T) where h > 0 else 0;  db = sum over the batch of the rounded dx (float64).
    With Dropout on this layer's input (scale = 1 / keep_prob): dx = round16((dy . w^T) * scale), masked by h > 0 when h (the
    dropped-out activation: its zeros are the ReLU's and the mask's) is given, by `mask` > 0 otherwise (the first layer)."""
    g = (np.asarray(dy16, np.float64) @ np.asarray(w16, np.float64).T).astype(np.float32)
    if scale != 1.0:
        g = g * np.float32(scale)
    g = round16(g, dtype)
    if h16 is not None:
        g = np.where(np.asarray(h16) > 0, g, np.float32(0))
    elif mask is not None:
        g = np.where(np.asarray(mask) > 0, g, np.float32(0))
    return g, g.astype(np.float64).sum(axis=0)


def dense_bwd_weight(x16, dy16):
    """MatMul bprop wrt the weight: dw = x^T . dy, exact products summed in float64 (the GPU sums in fp32)."""
    return np.asarray(x16, np.float64).T @ np.asarray(dy16, np.float64)


# ---- Dropout (numpy restatement) ----------------------------------------------------------------------------
# Reference: DenseLayer.construct, models/wide_deep/src/wide_and_deep.py:98,117-118 -- `x = self.dropout(x)` on the layer's
# INPUT while training, Dropout(p = 1 - keep_prob); DenseLayer's keep_prob defaults to 0.5 (:85) and WideDeepModel never
# passes one (:164-205), so dropout_flag alone switches a 0.5 dropout on.  MindSpore's generator cannot be restated
# (PARITY UNPINNED); the mask is the counter-based function the MI355X path documents in include/mrec.h, stated here
# independently over numpy uint64 arithmetic.
def _mix64(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dropout_mask(M, W, seed, step, layer, keep_prob, row0=0):
    """fp32 [M, W]: 1 / keep_prob where element (row0 + r, c) of the input of DenseLayer `layer` at step `step` is kept, else 0."""
    if W % 4:
        raise ValueError("W must be a multiple of 4")
    thresh = int(np.rint(np.float32(keep_prob) * np.float32(65536.0)))
    if thresh >= 65536:
        return np.ones((M, W), np.float32)
    thresh = max(thresh, 1)
    key = _mix64(np.uint64(seed % (1 << 64)) ^ _mix64(np.uint64(step * 16 + layer)))
    r = (np.arange(M, dtype=np.uint64) + np.uint64(row0))[:, None]
    c = np.arange(W, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        quad = _mix64(key + ((r * np.uint64(W) + c) >> np.uint64(2)))
    bits = (quad >> (np.uint64(16) * (c & np.uint64(3)))) & np.uint64(0xFFFF)
    return np.where(bits < np.uint64(thresh), np.float32(1.0) / np.float32(keep_prob), np.float32(0)).astype(np.float32)


def dropout(x, mask, dtype=None):
    """x * mask in fp32, rounded once to `dtype` ("bf16" / "f16"; None: fp32)."""
    y = np.asarray(x, np.float32) * np.asarray(mask, np.float32)
    return round16(y, dtype) if dtype else y
