"""Pins the CPU oracle (oracle/mrec_oracle.c) with hand-computed known answers.

The reference holds no golden vectors for this path (tests/ut and tests/st are
empty placeholders), and MindSpore -- where the arithmetic lives -- is not installable here, so
these known answers are derived by hand from the published formulas and the
reference's own hyper-parameters (models/wide_deep/src/wide_and_deep.py:415-433).  PARITY UNPINNED
at the MindSpore boundary; this file is what keeps the restatement honest.
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_unique_first_occurrence(oracle):
    # ops.Unique docs example: x=[1,2,5,2] -> y=[1,2,5], idx=[0,1,2,1]
    y, idx = oracle.unique(np.array([1, 2, 5, 2], np.int32))
    assert y.tolist() == [1, 2, 5] and idx.tolist() == [0, 1, 2, 1]
    y, idx = oracle.unique(np.array([5, 3, 5, 7, 3, 3, 9, -1, -2, 5], np.int64))
    assert y.tolist() == [5, 3, 7, 9, -1, -2] and idx.tolist() == [0, 1, 0, 2, 1, 1, 3, 4, 5, 0]
    y, idx = oracle.unique(np.array([], np.int32))
    assert y.size == 0 and idx.size == 0
    rng = np.random.default_rng(0)
    x = rng.integers(-50, 50, 1000)
    y, idx = oracle.unique(x)
    assert np.array_equal(y[idx], x)                       # reconstruction identity
    _, first = np.unique(x, return_index=True)
    assert np.array_equal(y, x[np.sort(first)])            # first-occurrence order


def test_gather_and_mask(oracle):
    table = np.arange(12, dtype=np.float32).reshape(4, 3)
    out = oracle.gather_rows(table, np.array([[3, 0], [1, 1]]))
    assert out.shape == (2, 2, 3)
    assert out[0, 0].tolist() == [9, 10, 11] and out[1, 1].tolist() == [3, 4, 5]
    out = oracle.gather_rows(table, np.array([2, 4, -1]), np.array([0.5, 1, 1], np.float32))
    assert out[0].tolist() == [3.0, 3.5, 4.0]
    assert out[1].tolist() == [0, 0, 0] and out[2].tolist() == [0, 0, 0]      # EmbeddingLookup: OOR -> 0


def test_wide_sum(oracle):
    w = np.array([[1.0], [2.0], [4.0]], np.float32)
    ids = np.array([[0, 2], [1, 1]])
    wts = np.array([[1.0, 0.5], [2.0, 3.0]], np.float32)
    assert oracle.wide_sum(w, ids, wts, 0.25).tolist() == [1 * 1 + 4 * 0.5 + 0.25, 2 * 2 + 2 * 3 + 0.25]


def test_segment_sum(oracle):
    vals = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.float32)
    out = oracle.segment_sum(vals, np.array([0, 2, 0, 2]), 3)
    assert out.tolist() == [[6, 8], [0, 0], [10, 12]]


def test_lazy_adam_one_step_by_hand(oracle):
    """First step from m=v=0, beta1_power=0.9, beta2_power=0.999, g=2 (after 1/1024 loss scale):
    m = 0.1*2 = 0.2, v = 0.001*4 = 0.004, lr_t = lr*sqrt(1-0.999)/(1-0.9),
    p -= lr_t * 0.2/(sqrt(0.004)+eps)."""
    p = np.full((6, 2), 0.5, np.float32); m = np.zeros_like(p); v = np.zeros_like(p)
    ids = np.array([4, 1, 4])                # id 4 twice: contributions 1024 + 1024 -> g = 2 after scale
    g = np.full((3, 2), 1024.0, np.float32)
    g[1] = 2048.0                             # id 1 once: g = 2
    oracle.sparse_lazy_adam(p, m, v, ids, g, None, lr=3.5e-4, eps=1e-8, b1_pow=0.9, b2_pow=0.999, grad_scale=1 / 1024)
    f = np.float32
    omb1, omb2 = f(1) - f(0.9), f(1) - f(0.999)          # fp32 constants: 1-0.999f = 0.00099998713
    m1, v1 = float(omb1) * 2.0, float(omb2) * 4.0
    lr_t = 3.5e-4 * np.sqrt(float(f(1) - f(0.999))) / float(f(1) - f(0.9))
    expect = 0.5 - lr_t * m1 / (np.sqrt(v1) + 1e-8)
    assert abs(m1 - 0.2) < 1e-7 and abs(v1 - 0.004) < 1e-7
    for r in (1, 4):
        assert np.allclose(p[r], expect, rtol=2e-7)
        assert np.allclose(m[r], m1, rtol=2e-7) and np.allclose(v[r], v1, rtol=2e-7)
    untouched = [0, 2, 3, 5]
    assert (p[untouched] == 0.5).all() and (m[untouched] == 0).all() and (v[untouched] == 0).all()   # lazy


def test_lazy_adam_mask_and_nesterov(oracle):
    p = np.zeros((3, 1), np.float32); m = np.full((3, 1), 0.1, np.float32); v = np.full((3, 1), 0.01, np.float32)
    oracle.sparse_lazy_adam(p, m, v, np.array([2]), np.array([[4.0]], np.float32), np.array([0.5], np.float32),
                            lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, b1_pow=0.5, b2_pow=0.75, grad_scale=1.0, nesterov=True)
    g = 2.0
    f = np.float32
    b1, b2, omb1, omb2 = float(f(0.9)), float(f(0.999)), float(f(1) - f(0.9)), float(f(1) - f(0.999))
    mn = b1 * float(f(0.1)) + omb1 * g; vn = b2 * float(f(0.01)) + omb2 * g * g
    lr_t = float(f(1e-2)) * np.sqrt(0.25) / 0.5
    expect = -lr_t * (b1 * mn + omb1 * g) / (np.sqrt(vn) + 1e-8)
    assert np.allclose(p[2, 0], expect, rtol=3e-7) and np.allclose(m[2, 0], mn, rtol=2e-7) and np.allclose(v[2, 0], vn, rtol=2e-7)


def test_ftrl_one_step_by_hand(oracle):
    """W&D wide optimizer (wide_and_deep.py:423-430): lr 5e-2, l1 = l2 = 1e-8, initial_accum 1.0, lr_power -0.5.
    w=0.3, a=1, lin=0, g=2: a'=5; sigma=(sqrt5-1)/lr; lin=g-sigma*w; w=(clip(lin)-lin)/(sqrt5/lr+2*l2)."""
    var = np.full((4, 1), 0.3, np.float32); acc = np.ones((4, 1), np.float32); lin = np.zeros((4, 1), np.float32)
    oracle.sparse_ftrl(var, acc, lin, np.array([3, 3]), np.array([[1024.0], [1024.0]], np.float32), None, lr=5e-2, l1=1e-8,
                       l2=1e-8, lr_power=-0.5, grad_scale=1 / 1024)
    lr, l1, l2 = 5e-2, 1e-8, 1e-8
    sigma = (np.sqrt(5.0) - 1.0) / lr
    ln = 2.0 - sigma * 0.3
    w = (np.clip(ln, -l1, l1) - ln) / (np.sqrt(5.0) / lr + 2 * l2)
    assert np.allclose(var[3, 0], w, rtol=3e-7) and np.allclose(acc[3, 0], 5.0) and np.allclose(lin[3, 0], ln, rtol=3e-7)
    assert (var[:3] == np.float32(0.3)).all() and (acc[:3] == 1).all()
    # |linear| <= l1 -> weight snaps to exactly 0
    var = np.full((1, 1), 0.0, np.float32); acc = np.ones((1, 1), np.float32); lin = np.zeros((1, 1), np.float32)
    oracle.sparse_ftrl(var, acc, lin, np.array([0]), np.array([[1e-9]], np.float32), None, lr=5e-2, l1=1e-3, l2=0.0)
    assert var[0, 0] == 0.0


def test_dense_equals_sparse_on_all_rows(oracle):
    rng = np.random.default_rng(1)
    V, D = 50, 4
    p = rng.standard_normal((V, D)).astype(np.float32); m = np.zeros_like(p); v = np.zeros_like(p)
    p2, m2, v2 = p.copy(), m.copy(), v.copy()
    g = rng.standard_normal((V, D)).astype(np.float32)
    oracle.sparse_lazy_adam(p, m, v, np.arange(V), g, None, grad_scale=0.5)
    oracle.dense_adam(p2, m2, v2, g, grad_scale=0.5)
    assert np.array_equal(p, p2) and np.array_equal(m, m2) and np.array_equal(v, v2)


def test_map_semantics(oracle):
    """MapParameter by example (README.md:160-205): get inserts defaults, put upserts, erase removes."""
    mp = oracle.Map(D=3, capacity=100, seed=5, sigma=0.01)
    a = mp.get(np.array([10, -7, 10**12]), insert_default=True)
    assert mp.size() == 3
    assert np.array_equal(a, oracle.normal_rows(5, [10, -7, 10**12], 3, 0.01))      # default rows keyed by key
    b = mp.get(np.array([10, 99]), insert_default=False)
    assert mp.size() == 3 and np.array_equal(b[0], a[0])
    mp.put(np.array([99, 10]), np.array([[1, 2, 3], [4, 5, 6]], np.float32))
    assert mp.size() == 4
    assert mp.get(np.array([10]), False)[0].tolist() == [4, 5, 6]
    mp.erase(np.array([10, 12345]))
    assert mp.size() == 3
    k, v = mp.export()
    assert sorted(k.tolist()) == [-7, 99, 10**12]
    assert v[k.tolist().index(99)].tolist() == [1, 2, 3]
    again = mp.get(np.array([10]), True)                    # re-inserted with its default row
    assert np.array_equal(again, oracle.normal_rows(5, [10], 3, 0.01)) and mp.size() == 4
    const = oracle.Map(D=2, capacity=4, fill=0.5)
    assert const.get(np.array([1]), True).tolist() == [[0.5, 0.5]]


def test_cross_layer_by_hand(oracle):
    # deep_and_cross.py:139-149: y = x0 * (x_l . w) + b + x_l
    x0 = np.array([[1.0, 2.0]], np.float32)
    w = np.array([[0.5, 0.25], [1.0, -1.0]], np.float32)
    b = np.array([[0.1, 0.2], [0.0, 0.0]], np.float32)
    y1 = x0 * (1 * 0.5 + 2 * 0.25) + b[0] + x0                      # [2.1, 4.2]
    y2 = x0 * (y1[0, 0] * 1 - y1[0, 1] * 1) + b[1] + y1
    assert np.allclose(oracle.cross_layers(x0, w[:1], b[:1]), y1)
    assert np.allclose(oracle.cross_layers(x0, w, b), y2)


def test_cross_layer_bwd_finite_difference(oracle):
    rng = np.random.default_rng(2)
    B, D, L = 5, 7, 3
    x0 = rng.standard_normal((B, D)).astype(np.float32); w = rng.standard_normal((L, D)).astype(np.float32) * 0.3
    b = rng.standard_normal((L, D)).astype(np.float32) * 0.1; dy = rng.standard_normal((B, D)).astype(np.float32)
    dx0, dw, db = oracle.cross_layers_bwd(x0, w, b, dy)

    def f(x0_, w_, b_):
        x0_ = x0_.astype(np.float64); cur = x0_.copy()
        for l in range(L):
            cur = x0_ * (cur @ w_[l].astype(np.float64))[:, None] + b_[l] + cur
        return float((cur * dy).sum())
    eps = 1e-3
    for (arr, grad) in ((x0, dx0), (w, dw), (b, db)):
        for idx in [(0, 0), (arr.shape[0] - 1, arr.shape[1] - 1), (1, 3)]:
            hi = arr.copy(); lo = arr.copy(); hi[idx] += eps; lo[idx] -= eps
            args_hi = [hi if a is arr else a for a in (x0, w, b)]
            args_lo = [lo if a is arr else a for a in (x0, w, b)]
            fd = (f(*args_hi) - f(*args_lo)) / (2 * eps)
            assert abs(fd - grad[idx]) <= 2e-2 * max(1.0, abs(fd)), (idx, fd, grad[idx])


def test_shard_route(oracle):
    loc, perm, counts = oracle.shard_route(np.array([5, 8, 3, 0, 9, 16, -1]), 4)
    # owners: 1,0,3,0,1,0,3 -> bucket order by owner, stable
    assert perm.tolist() == [1, 3, 5, 0, 4, 2, 6]
    assert loc.tolist() == [2, 0, 4, 1, 2, 0, -1]
    assert counts.tolist() == [3, 2, 0, 2]


def test_normal_generator_statistics_and_golden(oracle):
    x = oracle.fill_normal(1000, 20000, 16, 1.0).ravel()
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1.0) < 0.01
    assert abs((np.abs(x) < 1).mean() - 0.6827) < 0.005 and np.abs(x).max() < 6.0
    # committed golden bits of our generator (tests/golden/make_golden.py): guards the spec shared with the GPU
    gold = json.load(open(os.path.join(GOLD, "normal_seed1000.json")))
    rows = np.array(gold["rows"], np.int64)
    got = oracle.normal_rows(gold["seed"], rows, gold["D"], gold["sigma"])
    assert got.view(np.uint32).ravel().tolist() == gold["bits"]


def test_golden_step_fixture(oracle):
    """Replays the committed golden W&D embedding step (tests/golden/wd_step_small.npz)."""
    z = np.load(os.path.join(GOLD, "wd_step_small.npz"))
    p = oracle.fill_normal(int(z["seed"]), int(z["V"]), int(z["D"]), 0.01)
    m = np.zeros_like(p); v = np.zeros_like(p)
    emb = oracle.gather_rows(p, z["ids"], z["wts"])
    assert np.array_equal(emb, z["emb"])
    oracle.sparse_lazy_adam(p, m, v, z["ids"], z["g"], z["wts"], lr=3.5e-4, eps=1e-8, b1_pow=0.9, b2_pow=0.999,
                            grad_scale=1 / 1024)
    touched = np.unique(z["ids"])
    assert np.array_equal(p[touched], z["p_touched"]) and np.array_equal(m[touched], z["m_touched"])


def test_round16_and_dense_layer_known_answers(oracle):
    """The mixed-precision DenseLayer restatement (wide_and_deep.py:113-133): rounding to bf16 / f16 is round-to-nearest-
    even, the layer is act(x . w + b) on 16-bit operands with ONE rounding of the result."""
    r = oracle.round16(np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, -2.5], np.float32), "bf16")
    assert r.tolist() == [1.0, 1.0 + 2.0 ** -6, 1.0 + 2.0 ** -7, -2.5]          # ties to even, above a tie rounds up
    with np.errstate(over="ignore"):
        assert oracle.round16(np.array([1.0 + 2.0 ** -11, 65520.0], np.float32), "f16").tolist() == [1.0, float("inf")]
    x = np.array([[1.0, 2.0, -1.0], [0.5, 0.0, 4.0]], np.float32)
    w = np.array([[1.0, -1.0], [0.5, 0.25], [2.0, 1.0]], np.float32)
    b = np.array([0.125, -3.0], np.float32)
    y = oracle.dense_layer(x, w, b, True, "bf16")
    assert y.tolist() == [[0.125, 0.0], [8.625, 0.5]]
    g, db = oracle.dense_bwd_input(np.array([[1.0, 2.0]], np.float32), w, np.array([[1.0, 0.0, 3.0]], np.float32), "f16")
    assert g.tolist() == [[-1.0, 0.0, 4.0]] and db.tolist() == [-1.0, 0.0, 4.0]
    assert oracle.dense_bwd_weight(x, np.array([[1.0], [2.0]], np.float32)).tolist() == [[2.0], [2.0], [7.0]]


def test_mt_entries_are_bit_identical_to_single_thread(oracle):
    """The *_mt oracle entries (bench.py's all-core CPU baseline) partition rows / unique ids over threads and keep
    every per-id sum in ascending position order: same bits as the scalar entries."""
    rng = np.random.default_rng(5)
    V, D, B, F = 5000, 16, 300, 13
    ids = np.minimum(rng.zipf(1.2, size=(B, F)), V - 1).astype(np.int64)
    ids[0, 0] = V + 5                                          # out of range: skipped
    wts = rng.random((B, F)).astype(np.float32)
    p0 = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    g = rng.standard_normal((B * F, D)).astype(np.float32)
    assert np.array_equal(oracle.gather_rows(p0, ids, wts), oracle.gather_rows(p0, ids, wts, threads=5))
    w0 = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32)
    assert np.array_equal(oracle.wide_sum(w0, ids, wts, 0.25), oracle.wide_sum(w0, ids, wts, 0.25, threads=3))
    out = []
    for th in (0, 4):
        p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
        oracle.sparse_lazy_adam(p, m, v, ids, g, wts, grad_scale=1 / 1024, threads=th)
        w, a, l = w0.copy(), np.ones_like(w0), np.zeros_like(w0)
        oracle.sparse_ftrl(w, a, l, ids, g[:, :1].copy(), wts, grad_scale=1 / 1024, threads=th)
        out.append((p, m, v, w, a, l))
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_dropout_mask_golden_and_scalar_restatement(oracle):
    """The Dropout mask function: committed golden masks (tests/golden/dropout_masks.json) and a scalar restatement of the spec in
    plain Python integers (include/mrec.h, csrc/mrec_dropout.h) against the vectorised oracle."""
    import json
    M64 = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    for c in json.load(open(os.path.join(GOLD, "dropout_masks.json"))):
        mk = oracle.dropout_mask(c["M"], c["W"], c["seed"], c["step"], c["layer"], c["keep"], c["row0"])
        assert "".join("1" if x > 0 else "0" for x in mk.ravel()) == c["kept"]
        assert set(np.unique(mk)) <= {np.float32(0), np.float32(1) / np.float32(c["keep"])}
        key = mix((c["seed"] & M64) ^ mix(c["step"] * 16 + c["layer"]))
        thresh = int(round(c["keep"] * 65536))
        for r in range(c["M"]):
            for col in range(c["W"]):
                quad = mix((key + (((c["row0"] + r) * c["W"] + col) >> 2)) & M64)
                keep = ((quad >> (16 * (col & 3))) & 0xFFFF) < thresh
                assert keep == bool(mk[r, col] > 0)
    # keep_prob 1 is the identity; the kept fraction follows keep_prob
    assert (oracle.dropout_mask(7, 8, 1, 2, 3, 1.0) == 1).all()
    for keep in (0.5, 0.8, 0.1):
        mk = oracle.dropout_mask(2000, 512, 5, 9, 1, keep)
        assert abs((mk > 0).mean() - keep) <= 5 * np.sqrt(keep * (1 - keep) / mk.size) + 2.0 ** -16


def test_fast_mixed_oracle_agrees_with_the_float64_one():
    """tests/_oracle_mixed.py fast=True (fp32 GEMMs on the host, torch's 16-bit casts) against the float64 numpy restatement:
    three training steps, both 16-bit dtypes -- losses, tables and dense parameters agree to fp32 accumulation error."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_mixed import OracleMixedEngine
    from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
    for dt, name in (("f16", "fp16"), ("bf16", "bf16")):
        cfg = WideDeepConfig(vocab_size=5000, emb_dim=16, field_size=13, batch_size=256, deep_layer_dim=[64, 32, 16], mlp_dtype=name)
        a, b = OracleMixedEngine(cfg, dt), OracleMixedEngine(cfg, dt, fast=True)
        for s in range(3):
            ids, wts, label = synthetic_batch(cfg, "cpu", "zipf", seed=2 + s, signal=True)
            la = a.train_step(ids.numpy(), wts.numpy(), label.numpy().ravel())
            lb = b.train_step(ids.numpy(), wts.numpy(), label.numpy().ravel())
            assert abs(la - lb) <= 1e-6 * abs(la)
        assert np.abs(a.deep - b.deep).max() <= 2 * cfg.adam_lr * 3 and np.mean(a.deep == b.deep) > 0.99
        assert np.abs(a.flat - b.flat).max() <= 2 * cfg.adam_lr * 3 and np.mean(np.abs(a.flat - b.flat) <= 1e-6) > 0.99
        pa, pb = a.predict(ids.numpy(), wts.numpy())[1], b.predict(ids.numpy(), wts.numpy())[1]
        assert np.abs(pa - pb).max() <= 1e-3
