"""GPU parity: every HIP kernel, through the C-ABI, against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for ids / dedup / index work; fp32 rows within 1e-5
relative -- and bit-exact where the summation order provably matches the oracle's.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def ids_case(kind, n, V, rng, dtype):
    if kind == "uniform":
        x = rng.integers(0, V, size=n)
    elif kind == "dups":
        x = rng.integers(0, max(n // 8, 1), size=n)
    elif kind == "zipf":
        x = np.minimum(rng.zipf(1.05, size=n) - 1, V - 1)
    elif kind == "hot":  # Criteo-like: a few ids repeated thousands of times + uniform rest
        x = rng.integers(0, V, size=n)
        hot = rng.random(n) < 0.35
        x[hot] = rng.integers(0, 13, size=hot.sum())
    elif kind == "same":
        x = np.full(n, 7)
    else:
        raise ValueError(kind)
    return x.astype(dtype)


def crossing(plan, D):
    """Per unique id: does its run of sorted entries cross a window boundary of the apply kernel?
    (runs inside one window are summed in oracle order; the window -- 8 entries on every path today -- is
    whatever mrec_sparse_apply_window reports)."""
    from mindrec_amd import ops
    AW = ops.apply_window(D)
    assert AW == 8
    offs = plan.seg_offsets[: plan.U + 1].cpu().numpy().astype(np.int64)
    return (offs[:-1] // AW) != ((offs[1:] - 1) // AW)


def row_rel(a, b):
    """Row-wise relative error, inf-norm: max_r |a_r - b_r|_inf / |b_r|_inf."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.maximum(np.abs(b).max(axis=1), 1e-30)
    return float((np.abs(a - b).max(axis=1) / den).max()) if a.size else 0.0


def check_rows(got, ref, rows_exact, rows_close, tol=RTOL):
    """got/ref: tuples of [V, D] arrays, parameter first, then optimizer accumulators.  Rows outside
    rows_close must match bit for bit (they were summed in oracle order, or never touched).  Rows in
    rows_close: the parameter within tol row-relative; accumulators (sums of cancelling gradient
    terms, so an element can be arbitrarily close to 0) within tol of the tensor's magnitude."""
    V = ref[0].shape[0]
    other = np.ones(V, bool); other[rows_close] = False
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a[other].view(np.uint32), b[other].view(np.uint32))
        if len(rows_close) and k == 0:
            assert row_rel(a[rows_close], b[rows_close]) <= tol, row_rel(a[rows_close], b[rows_close])
        elif len(rows_close):
            err = np.abs(a[rows_close].astype(np.float64) - b[rows_close]).max()
            assert err <= tol * np.abs(b).max(), (err, np.abs(b).max())
    assert len(rows_exact) > 0 or len(rows_close) > 0


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-6))) if a.size else 0.0


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("kind,n", [("uniform", 1), ("uniform", 63), ("uniform", 5000), ("dups", 70001), ("zipf", 40000),
                                    ("hot", 100000), ("same", 3000)])
def test_unique_and_group(dev, oracle, dtype, kind, n):
    from mindrec_amd import ops
    rng = np.random.default_rng(n + (1 if dtype == np.int64 else 0))
    V = 2_000_000 if dtype == np.int32 else 2**40
    x = ids_case(kind, n, V, rng, dtype)
    if dtype == np.int64 and kind == "uniform":
        x = x - 2**39  # negative keys too
    u_ref, inv_ref = oracle.unique(x)
    plan = ops.sparse_plan(T(x, dev)) if n % 2 else ops.group_by_inverse(ops.unique(T(x, dev)))   # fused and two-call forms
    assert plan.U == u_ref.size
    assert np.array_equal(plan.uniq.cpu().numpy(), u_ref)          # first-occurrence order, bit-exact
    assert np.array_equal(plan.inv.cpu().numpy(), inv_ref)
    # inverted index: stable sort of positions by group
    order = np.argsort(inv_ref, kind="stable").astype(np.int32)
    assert np.array_equal(plan.sorted_pos[:n].cpu().numpy(), order)
    assert np.array_equal(plan.sorted_seg[:n].cpu().numpy(), inv_ref[order])
    offs = np.concatenate([[0], np.cumsum(np.bincount(inv_ref, minlength=u_ref.size))]).astype(np.int32)
    assert np.array_equal(plan.seg_offsets[: u_ref.size + 1].cpu().numpy(), offs)


def test_unique_empty(dev):
    from mindrec_amd import ops
    d = ops.unique(torch.empty(0, dtype=torch.int32, device=dev))
    assert d.U == 0
    p = ops.group_by_inverse(d)
    assert int(p.seg_offsets[0].item()) == 0
    p = ops.sparse_plan(torch.empty(0, dtype=torch.int64, device=dev))
    assert p.U == 0 and int(p.seg_offsets[0].item()) == 0


def test_unique_large_bitexact(dev, oracle):
    """BASELINE cfg2 size: 16384 x 39 ids (13 constant dense-field ids + Zipf categorical)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(1000)
    B, F = 16384, 39
    x = np.minimum(rng.zipf(1.05, size=(B, F)) + 13, 200_000_000 - 1).astype(np.int32)
    x[:, :13] = np.arange(13, dtype=np.int32)
    u_ref, inv_ref = oracle.unique(x)
    plan = ops.sparse_plan(T(x, dev))
    assert plan.U == u_ref.size
    assert np.array_equal(plan.uniq.cpu().numpy(), u_ref)
    assert np.array_equal(plan.inv.cpu().numpy(), inv_ref)
    order = np.argsort(inv_ref, kind="stable").astype(np.int32)
    assert np.array_equal(plan.sorted_pos.cpu().numpy(), order)


def test_fill_normal_bitexact(dev, oracle):
    from mindrec_amd import ops
    for (V, D) in [(1000, 80), (257, 16), (100, 1), (64, 30)]:
        t = torch.empty((V, D), dtype=torch.float32, device=dev)
        ops.fill_normal_(t, seed=1000, sigma=0.01)
        ref = oracle.fill_normal(1000, V, D, 0.01)
        assert np.array_equal(t.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # row offset + padded rows
    t = torch.zeros((50, 96), dtype=torch.float32, device=dev)
    ops.fill_normal_(t[:, :80], seed=3, sigma=1.0, row0=12345678901)
    ref = oracle.fill_normal(3, 50, 80, 1.0, row0=12345678901)
    assert np.array_equal(t[:, :80].cpu().numpy(), ref)
    assert float(t[:, 80:].abs().max()) == 0.0
    # a row shard: local row r holds global row rank + r * world
    t = torch.empty((40, 16), dtype=torch.float32, device=dev)
    ops.fill_normal_(t, seed=9, sigma=0.01, row0=3, row_stride=8)
    assert np.array_equal(t.cpu().numpy(), oracle.normal_rows(9, 3 + 8 * np.arange(40), 16, 0.01))


@pytest.mark.parametrize("D", [80, 16, 128, 1, 30, 4, 260, 512])
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_gather_rows(dev, oracle, D, dtype):
    from mindrec_amd import ops
    rng = np.random.default_rng(D)
    V = 5000
    table = rng.standard_normal((V, D)).astype(np.float32)
    ids = rng.integers(-3, V + 3, size=(37, 11)).astype(dtype)     # includes out-of-range both sides
    wts = rng.random((37, 11)).astype(np.float32)
    tt, ti, tw = T(table, dev), T(ids, dev), T(wts, dev)
    out = ops.gather_rows(tt, ti).cpu().numpy()
    assert np.array_equal(out, oracle.gather_rows(table, ids))
    out = ops.gather_rows(tt, ti, row_scale=tw).cpu().numpy()
    assert np.array_equal(out, oracle.gather_rows(table, ids, wts))   # one fp32 multiply: bit-exact


def test_gather_strided_table(dev, oracle):
    """Rows embedded in a wider allocation (ld > D)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(5)
    big = rng.standard_normal((300, 96)).astype(np.float32)
    ids = rng.integers(0, 300, size=1000).astype(np.int32)
    out = ops.gather_rows(T(big, dev)[:, :80], T(ids, dev)).cpu().numpy()
    assert np.array_equal(out, big[ids, :80])


@pytest.mark.parametrize("F", [26, 39, 3, 1, 300, 2500])
def test_wide_sum(dev, oracle, F):
    from mindrec_amd import ops
    rng = np.random.default_rng(F)
    V, B = 10000, 777
    w = (rng.standard_normal((V, 1)) * 0.01).astype(np.float32)
    ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
    wts = rng.random((B, F)).astype(np.float32)
    bias = np.array([0.125], np.float32)
    out = ops.wide_sum(T(w, dev), T(ids, dev), T(wts, dev), T(bias, dev)).cpu().numpy()
    ref = oracle.wide_sum(w, ids, wts, float(bias[0]))
    assert np.array_equal(out, ref)   # same sequential order over fields -> bit-exact


def _adam_case(dev, oracle, kind, n, D, V, dtype, use_scale, nesterov=False, steps=2):
    from mindrec_amd import ops
    rng = np.random.default_rng(n * 7 + D)
    p = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    m = np.zeros((V, D), np.float32); v = np.zeros((V, D), np.float32)
    tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
    b1p, b2p = np.float32(1.0), np.float32(1.0)
    cross_rows = set()
    for step in range(steps):
        ids = ids_case(kind, n, V, rng, dtype)
        g = (rng.standard_normal((n, D)) * 1024).astype(np.float32)
        sc = rng.random(n).astype(np.float32) if use_scale else None
        b1p = np.float32(b1p * np.float32(0.9)); b2p = np.float32(b2p * np.float32(0.999))
        oracle.sparse_lazy_adam(p, m, v, ids, g, sc, lr=3.5e-4, eps=1e-8, b1_pow=float(b1p), b2_pow=float(b2p),
                                grad_scale=1.0 / 1024, nesterov=nesterov)
        plan = ops.sparse_plan(T(ids, dev))
        ops.sparse_lazy_adam_(tp, tm, tv, plan, T(g, dev), T(sc, dev) if use_scale else None, lr=3.5e-4, eps=1e-8,
                              beta1_power=float(b1p), beta2_power=float(b2p), grad_scale=1.0 / 1024,
                              use_nesterov=nesterov)
        cross_rows.update(plan.uniq.cpu().numpy()[crossing(plan, D)].tolist())
    cross_rows = np.array(sorted(r for r in cross_rows if 0 <= r < V), dtype=np.int64)
    return (tp.cpu().numpy(), tm.cpu().numpy(), tv.cpu().numpy()), (p, m, v), cross_rows


@pytest.mark.parametrize("D", [80, 16, 128, 1, 30])
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_lazy_adam_unique_ids_bitexact(dev, oracle, D, dtype):
    """No duplicate spans a window -> summation order equals the oracle's -> bit-exact."""
    got, ref, cross = _adam_case(dev, oracle, "uniform", 3000, D, 1_000_000, dtype, use_scale=True)
    # 3000 ids over 1 M rows: at most a few duplicate PAIRS, and a + b does not depend on the order
    for a, b in zip(got, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("kind", ["dups", "zipf", "hot", "same"])
@pytest.mark.parametrize("D", [80, 16, 1, 30])
def test_lazy_adam_duplicates(dev, oracle, kind, D):
    got, ref, cross = _adam_case(dev, oracle, kind, 20000, D, 50000, np.int32, use_scale=True)
    # ids whose run stays inside one window: bit-exact.  ids whose run crosses windows are summed as
    # a fixed tree of window partials: rows within 1e-5 relative of the sequential oracle (1e-4 for
    # the 20000-copies-of-one-id case, where the oracle's own fp32 chain is ~1e-5 from exact).
    check_rows(got, ref, rows_exact=[0], rows_close=cross, tol=1e-4 if kind == "same" else RTOL)


def test_lazy_adam_nesterov_and_noscale(dev, oracle):
    got, ref, cross = _adam_case(dev, oracle, "dups", 5000, 80, 20000, np.int32, use_scale=False, nesterov=True)
    check_rows(got, ref, rows_exact=[0], rows_close=cross)


def test_lazy_adam_out_of_range_ids_skipped(dev, oracle):
    from mindrec_amd import ops
    V, D = 100, 16
    p = np.ones((V, D), np.float32); m = np.zeros_like(p); v = np.zeros_like(p)
    ids = np.array([5, 100, -1, 5, 99], np.int32)
    g = np.ones((5, D), np.float32)
    tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
    ops.sparse_lazy_adam_(tp, tm, tv, ops.sparse_plan(T(ids, dev)), T(g, dev))
    oracle.sparse_lazy_adam(p, m, v, ids, g)
    assert np.array_equal(tp.cpu().numpy(), p)
    assert (p[[5, 99]] != 1).all() and (np.delete(p, [5, 99], 0) == 1).all()


@pytest.mark.parametrize("kind", ["uniform", "hot", "same"])
@pytest.mark.parametrize("D", [1, 16])
def test_sparse_ftrl(dev, oracle, kind, D):
    from mindrec_amd import ops
    rng = np.random.default_rng(11 + D)
    V, n = 30000, 20000
    var = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    acc = np.ones((V, D), np.float32); lin = np.zeros((V, D), np.float32)
    tv, ta, tl = T(var, dev), T(acc, dev), T(lin, dev)
    cross = set()
    for step in range(3):
        ids = ids_case(kind, n, V, rng, np.int32)
        g = (rng.standard_normal((n, D)) * 1024).astype(np.float32)
        oracle.sparse_ftrl(var, acc, lin, ids, g, None, lr=5e-2, l1=1e-8, l2=1e-8, grad_scale=1.0 / 1024)
        plan = ops.sparse_plan(T(ids, dev))
        ops.sparse_ftrl_(tv, ta, tl, plan, T(g, dev), None, lr=5e-2, l1=1e-8, l2=1e-8, grad_scale=1.0 / 1024)
        cross.update(plan.uniq.cpu().numpy()[crossing(plan, D)].tolist())
    cross = np.array(sorted(cross), dtype=np.int64)
    got = (tv.cpu().numpy(), ta.cpu().numpy(), tl.cpu().numpy())
    # FTRL's weight is a ratio of cancelling sums; for ids with thousands of copies per step the
    # sequential fp32 oracle is itself ~1e-4 from exact, so those rows get the looser bound.
    check_rows(got, (var, acc, lin), rows_exact=[0], rows_close=cross, tol=5e-5 if kind == "uniform" else 2e-3)


def test_ftrl_general_lr_power(dev, oracle):
    from mindrec_amd import ops
    rng = np.random.default_rng(2)
    V, n, D = 1000, 500, 8
    var = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    acc = np.full((V, D), 0.1, np.float32); lin = np.zeros((V, D), np.float32)
    tv, ta, tl = T(var, dev), T(acc, dev), T(lin, dev)
    ids = rng.integers(0, V, n).astype(np.int32); g = rng.standard_normal((n, D)).astype(np.float32)
    oracle.sparse_ftrl(var, acc, lin, ids, g, None, lr=0.01, l1=1e-3, l2=1e-3, lr_power=-0.3)
    ops.sparse_ftrl_(tv, ta, tl, ops.sparse_plan(T(ids, dev)), T(g, dev), None, lr=0.01, l1=1e-3, l2=1e-3, lr_power=-0.3)
    assert np.allclose(tv.cpu().numpy(), var, rtol=1e-4, atol=1e-6)   # powf differs host/device by ulps


@pytest.mark.parametrize("kind", ["uniform", "zipf", "hot"])
@pytest.mark.parametrize("D", [80, 1, 30, 300, 7, 130])
def test_segment_sum(dev, oracle, kind, D):
    from mindrec_amd import ops
    rng = np.random.default_rng(3 + D)
    n = 9000
    ids = ids_case(kind, n, 100000, rng, np.int32)
    g = rng.standard_normal((n, D)).astype(np.float32)
    u_ref, inv_ref = oracle.unique(ids)
    ref = oracle.segment_sum(g, inv_ref, u_ref.size)
    plan = ops.sparse_plan(T(ids, dev))
    out = ops.segment_sum(plan, T(g, dev))[: plan.U].cpu().numpy()
    cr = crossing(plan, D)
    assert (~cr).sum() > 0
    assert np.array_equal(out[~cr].view(np.uint32), ref[~cr].view(np.uint32))     # oracle order: bit-exact
    # any-order fp32 summation bound against the exact (float64) sum: |err| <= (count-1) * eps * sum|x|
    exact = np.zeros((u_ref.size, D)); absum = np.zeros((u_ref.size, D))
    np.add.at(exact, inv_ref, g.astype(np.float64)); np.add.at(absum, inv_ref, np.abs(g).astype(np.float64))
    cnt = np.bincount(inv_ref, minlength=u_ref.size)[:, None]
    bound = np.maximum(cnt - 1, 1) * 2.0 ** -23 * absum
    assert (np.abs(out - exact) <= bound).all()
    assert (np.abs(ref - exact) <= bound).all()        # the oracle obeys the same bound


def test_dense_optimizers(dev, oracle):
    from mindrec_amd import ops
    rng = np.random.default_rng(4)
    for n in (1, 7, 4096, 100003):
        p = rng.standard_normal(n).astype(np.float32); m = rng.standard_normal(n).astype(np.float32) * 0.1
        v = rng.random(n).astype(np.float32) * 0.01; g = rng.standard_normal(n).astype(np.float32) * 1000
        tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
        ops.dense_adam_(tp, tm, tv, T(g, dev), lr=1e-4, beta1_power=0.81, beta2_power=0.998, grad_scale=1e-3)
        oracle.dense_adam(p, m, v, g, lr=1e-4, b1_pow=0.81, b2_pow=0.998, grad_scale=1e-3)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tm.cpu().numpy(), m) and np.array_equal(tv.cpu().numpy(), v)
        w = rng.standard_normal(n).astype(np.float32) * 0.01; a = np.ones(n, np.float32); l = np.zeros(n, np.float32)
        tw, ta, tl = T(w, dev), T(a, dev), T(l, dev)
        ops.dense_ftrl_(tw, ta, tl, T(g, dev), grad_scale=1e-3)
        oracle.dense_ftrl(w, a, l, g, grad_scale=1e-3)
        assert np.array_equal(tw.cpu().numpy(), w) and np.array_equal(ta.cpu().numpy(), a) and np.array_equal(tl.cpu().numpy(), l)


def test_key_index_matches_oracle_map(dev, oracle):
    from mindrec_amd import ops
    rng = np.random.default_rng(9)
    D, cap = 16, 5000
    om = oracle.Map(D, cap, seed=77, sigma=0.01)
    ki = ops.KeyIndex(cap, dev)
    table = torch.zeros((cap, D), dtype=torch.float32, device=dev)
    for step in range(6):
        keys = rng.integers(-2**40, 2**40, size=700)
        keys[:100] = rng.integers(0, 50, size=100)          # recurring keys
        ukeys, _ = oracle.unique(keys.astype(np.int64))
        tk = T(ukeys, dev)
        rows, is_new = ki.find_or_insert(tk, insert=True)
        ops.init_rows_(table, rows, tk, is_new, seed=77, sigma=0.01)
        ref_rows = om.find_or_insert(ukeys, True)
        assert np.array_equal(rows.cpu().numpy(), ref_rows)             # same deterministic row numbering
        got = ops.gather_rows(table, rows).cpu().numpy()
        assert np.array_equal(got, om.get(ukeys, True))
        if step == 2:                                                    # erase some, then keep going
            er = ukeys[::3].copy()
            ki.erase(T(er, dev)); om.erase(er)
            assert len(ki) == om.size()
    # lookups without insertion
    probe = np.concatenate([ukeys[:10], np.array([2**50 + 1, 2**50 + 2])]).astype(np.int64)
    rows, _ = ki.find_or_insert(T(probe, dev), insert=False)
    assert np.array_equal(rows.cpu().numpy(), om.find_or_insert(probe, False))
    assert len(ki) == om.size()
    k, r = ki.export()
    ok, ov = om.export()
    assert np.array_equal(np.sort(k.cpu().numpy()), np.sort(ok))
    vals = ops.gather_rows(table, r).cpu().numpy()
    order_g, order_o = np.argsort(k.cpu().numpy()), np.argsort(ok)
    assert np.array_equal(vals[order_g], ov[order_o])


def test_key_index_full_and_reuse(dev):
    from mindrec_amd import ops
    ki = ops.KeyIndex(100, dev)
    k1 = torch.arange(0, 100, dtype=torch.int64, device=dev)
    rows, new = ki.find_or_insert(k1)
    assert rows.tolist() == list(range(100)) and int(new.sum()) == 100
    rows, new = ki.find_or_insert(torch.arange(100, 110, dtype=torch.int64, device=dev))
    assert (rows == -1).all() and int(new.sum()) == 0
    hwm, live, dropped, free = ki.counters()
    assert (hwm, live, dropped, free) == (100, 100, 10, 0)
    ki.erase(torch.tensor([3, 5, 7], dtype=torch.int64, device=dev))
    rows, new = ki.find_or_insert(torch.tensor([200, 201], dtype=torch.int64, device=dev))
    assert rows.tolist() == [7, 5]          # free list is LIFO in erase order
    assert ki.counters()[1] == 99


def test_key_index_long_churn_keeps_empty_slots(dev):
    """Insert / erase churn with a key space far larger than the slot array (the pattern of the feature-cache
    tier's evict + prepare and of MapParameter.evict): without tombstone reclamation the empty slots run out
    after ~20 steps and a probe for a missing key never ends.  300 steps here, checked against a host dict:
    every live key keeps its row, rows are distinct and in range, misses terminate with -1, tombstones stay
    under the rebuild threshold and the slot array was rebuilt many times.  (Row NUMBERS after reuse are this
    index's own free-list policy; the oracle's map is append-only, so they are not compared here.)"""
    from mindrec_amd import ops
    rng = np.random.default_rng(21)
    cap = 900                                   # 2048 slots
    ki = ops.KeyIndex(cap, dev)
    row_of = {}
    order = []                                  # residents, oldest first
    for step in range(300):
        fresh = np.unique(rng.integers(0, 2**45, size=660).astype(np.int64))
        fresh = np.array([k for k in fresh.tolist() if k not in row_of], np.int64)
        room = cap - len(order)
        if fresh.size > room:                   # evict the oldest residents to make room
            k = fresh.size - room
            out, order = order[:k], order[k:]
            ki.erase(T(np.array(out, np.int64), dev))
            for key in out:
                del row_of[key]
        rows, is_new = ki.find_or_insert(T(fresh, dev), insert=True)
        rows = rows.cpu().numpy()
        assert int(is_new.sum()) == fresh.size and rows.min() >= 0 and rows.max() < cap
        for key, r in zip(fresh.tolist(), rows.tolist()):
            row_of[key] = r
        order += fresh.tolist()
        assert len(set(row_of.values())) == len(row_of) == len(order)          # live keys own distinct rows
        if step % 25 == 24:
            c = ki.counters_all()
            assert c[1] == len(order) and c[2] == 0
            assert 0 <= c[4] * 5 <= 2048 + 5 * 660, c       # tombstones bounded by the rebuild threshold (+ one call)
            res = np.array(order[:300], np.int64)
            probe = np.concatenate([res, rng.integers(2**46, 2**47, size=200)]).astype(np.int64)
            r, _ = ki.find_or_insert(T(probe, dev), insert=False)       # misses must terminate
            r = r.cpu().numpy()
            assert r[:res.size].tolist() == [row_of[k] for k in res.tolist()] and (r[res.size:] == -1).all()
    assert ki.counters_all()[6] >= 10           # the slot array was rebuilt many times


@pytest.mark.parametrize("B,D,L", [(1000, 1170, 6), (77, 64, 3), (33, 30, 1), (500, 1500, 8),
                                   (3, 1170, 6),          # most waves of the one workgroup see no row
                                   (20000, 200, 2),       # every CU a slab, several rows per wave, narrow rows
                                   (600, 2048, 7)])       # the widest instantiation (two LDS exchange regions)
def test_cross_layers(dev, oracle, B, D, L):
    from mindrec_amd import ops
    rng = np.random.default_rng(B)
    x0 = rng.standard_normal((B, D)).astype(np.float32) * 0.5
    w = (rng.standard_normal((L, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.standard_normal((L, D)) * 0.1).astype(np.float32)
    out = ops.cross_layers(T(x0, dev), T(w, dev), T(b, dev)).cpu().numpy()
    ref = oracle.cross_layers(x0, w, b)
    # the dots are summed in another order than the oracle's sequential fp32 loop: both are held against the same layers in
    # float64 (1e-5 of the row's magnitude for the kernel; the oracle's own distance from it bounds their difference)
    x64, xl = x0.astype(np.float64), x0.astype(np.float64)
    for l in range(L):
        xl = x64 * (xl @ w[l].astype(np.float64))[:, None] + b[l].astype(np.float64) + xl
    assert row_rel(out, xl) <= 1e-5, row_rel(out, xl)
    assert row_rel(out, ref) <= 1e-5 + 2 * row_rel(ref, xl), (row_rel(out, ref), row_rel(ref, xl))
    dy = rng.standard_normal((B, D)).astype(np.float32)
    dx0, dw, db = ops.cross_layers_bwd(T(x0, dev), T(w, dev), T(b, dev), T(dy, dev))
    rdx0, rdw, rdb = oracle.cross_layers_bwd(x0, w, b, dy)
    assert np.allclose(dx0.cpu().numpy(), rdx0, rtol=1e-4, atol=1e-4)
    scale = np.abs(rdw).max()
    assert np.abs(dw.cpu().numpy() - rdw).max() <= 2e-5 * max(scale, 1) * np.sqrt(B)
    assert np.abs(db.cpu().numpy() - rdb).max() <= 2e-5 * max(np.abs(rdb).max(), 1) * np.sqrt(B)
    # dx0 += ...: onto another branch's gradient, the same sum to the bit
    base = rng.standard_normal((B, D)).astype(np.float32)
    acc = T(base, dev)
    _, dw2, db2 = ops.cross_layers_bwd(T(x0, dev), T(w, dev), T(b, dev), T(dy, dev), dx0_out=acc, accumulate=True)
    assert np.array_equal(acc.cpu().numpy(), base + dx0.cpu().numpy())
    assert torch.equal(dw2, dw) and torch.equal(db2, db)


@pytest.mark.parametrize("S,D", [(2, 80), (8, 80), (3, 1), (8, 30), (4, 300)])
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_shard_route_roundtrip(dev, oracle, S, D, dtype):
    from mindrec_amd import ops
    rng = np.random.default_rng(S)
    n = 10007
    ids = rng.integers(0, 10**6, size=n).astype(dtype)
    if dtype == np.int64:
        ids[:50] = -ids[:50]
    loc, perm, counts = ops.shard_route(T(ids, dev), S)
    rloc, rperm, rcounts = oracle.shard_route(ids, S)
    assert np.array_equal(loc.cpu().numpy(), rloc.astype(dtype))
    assert np.array_equal(perm.cpu().numpy(), rperm)
    assert np.array_equal(counts.cpu().numpy(), rcounts)
    rows = rng.standard_normal((n, D)).astype(np.float32)
    sc = rng.random(n).astype(np.float32)
    back = ops.shard_unroute(T(rows, dev), perm, T(sc, dev)).cpu().numpy()
    exp = np.empty_like(rows); exp[rperm] = rows * sc[rperm][:, None]
    assert np.array_equal(back, exp)
    fwd = ops.shard_route_rows(T(rows, dev), perm, T(sc, dev)).cpu().numpy()
    assert np.array_equal(fwd, rows[rperm] * sc[rperm][:, None])


def test_cpu_tensor_is_refused():
    """No silent CPU fallback: the product path refuses host tensors."""
    from mindrec_amd import ops
    with pytest.raises(RuntimeError):
        ops.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int32))


@pytest.mark.parametrize("D", [80, 16, 30])
def test_gather_bf16_out_matches_cast(dev, oracle, D):
    """bf16-output gather == fp32 gather followed by a round-to-nearest-even cast, bit for bit."""
    from mindrec_amd import ops
    rng = np.random.default_rng(D)
    V = 4000
    table = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    ids = rng.integers(-2, V + 2, size=(50, 13)).astype(np.int32)
    wts = rng.random((50, 13)).astype(np.float32)
    tt, ti, tw = T(table, dev), T(ids, dev), T(wts, dev)
    got = ops.gather_rows(tt, ti, tw, out_dtype=torch.bfloat16)
    ref = torch.from_numpy(oracle.gather_rows(table, ids, wts)).to(torch.bfloat16)
    assert got.dtype == torch.bfloat16 and torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("kind", ["uniform", "hot"])
@pytest.mark.parametrize("D", [80, 30])
def test_lazy_adam_bf16_gradients_equal_widened_fp32(dev, oracle, kind, D):
    """Feeding bf16 row gradients gives exactly the result of feeding the same values widened to fp32."""
    from mindrec_amd import ops
    rng = np.random.default_rng(17 + D)
    V, n = 20000, 12000
    ids = ids_case(kind, n, V, rng, np.int32)
    g16 = torch.from_numpy((rng.standard_normal((n, D)) * 100).astype(np.float32)).to(torch.bfloat16).to(dev)
    sc = T(rng.random(n).astype(np.float32), dev)
    res = []
    for g in (g16, g16.float()):
        p = T((np.random.default_rng(1).standard_normal((V, D)) * 0.01).astype(np.float32), dev)
        m = torch.zeros_like(p); v = torch.zeros_like(p)
        ops.sparse_lazy_adam_(p, m, v, ops.sparse_plan(T(ids, dev)), g, sc, grad_scale=1 / 1024)
        res.append((p, m, v))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_dense_adam_bf16_grad_and_shadow(dev, oracle):
    from mindrec_amd import ops
    rng = np.random.default_rng(8)
    for n in (5, 4096, 100003):
        p = rng.standard_normal(n).astype(np.float32) * 0.01
        g16 = torch.from_numpy(rng.standard_normal(n).astype(np.float32) * 50).to(torch.bfloat16)
        m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
        sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        ops.dense_adam_(tp, tm, tv, g16.to(dev), grad_scale=1 / 1024, shadow_bf16=sh)
        oracle.dense_adam(p, m, v, g16.float().numpy(), grad_scale=1 / 1024)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tm.cpu().numpy(), m) and np.array_equal(tv.cpu().numpy(), v)
        assert torch.equal(sh.cpu().view(torch.int16), torch.from_numpy(p).to(torch.bfloat16).view(torch.int16))



def test_dense_adam_with_l2_term_any_length(dev, oracle):
    """nn.Adam over a whole table with the loss's L2 term inside the kernel (sparse=False; DeepFM): bit-exact against the restatement,
    sum(p^2) of the old values in float64 -- also for lengths that are not a multiple of 4 (DeepFM's [184 965, 1] linear table)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(11)
    for n in (3, 4096, 184965, 100003):
        p = rng.standard_normal(n).astype(np.float32) * 0.01
        g = rng.standard_normal(n).astype(np.float32) * 30
        m = (rng.standard_normal(n) * 0.1).astype(np.float32); v = (rng.random(n) * 0.01).astype(np.float32)
        tp, tm, tv = T(p, dev), T(m, dev), T(v, dev)
        ss = torch.full((1,), 7.0, dtype=torch.float64, device=dev)
        ops.dense_adam_l2_(tp, tm, tv, T(g, dev), 8e-5 * 1024, sumsq=ss, accumulate=True, grad_scale=1 / 1024)
        want = 7.0 + float((p.astype(np.float64) ** 2).sum())
        gg = (g + (p * np.float32(8e-5 * 1024)).astype(np.float32)).astype(np.float32)
        oracle.dense_adam(p, m, v, gg, grad_scale=1 / 1024)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tm.cpu().numpy(), m) and np.array_equal(tv.cpu().numpy(), v), n
        assert abs(float(ss) - want) <= 1e-12 * want, n


@pytest.mark.parametrize("V,D", [(5000, 80), (3001, 30), (184965, 1), (700, 16)])
def test_dense_adam_over_a_gather_gradient_equals_scatter_then_adam(dev, oracle, V, D):
    """ops.dense_adam_rows_l2_ (the whole table's nn.Adam with the gradient looked up per row from the step's group sums) against the
    three-pass form it replaces -- zero a [V, D] gradient, scatter the sums, ops.dense_adam_l2_ -- bit for bit, sum(p^2) included;
    duplicate-heavy ids, ids outside the table (skipped), with and without the L2 term."""
    from mindrec_amd import ops
    rng = np.random.default_rng(V + D)
    n = 4000
    ids = np.minimum(rng.zipf(1.2, size=n), V + 3).astype(np.int32)            # a few ids >= V: no row
    ids[:50] = 7
    g = (rng.standard_normal((n, D)) * 20).astype(np.float32)
    sc = rng.random(n).astype(np.float32)
    p0 = (rng.standard_normal((V, D)) * 0.01).astype(np.float32)
    m0 = (rng.standard_normal((V, D)) * 0.1).astype(np.float32)
    v0 = (rng.random((V, D)) * 0.01).astype(np.float32)
    for l2 in (0.0, 8e-5 * 1024):
        kw = dict(lr=5e-4, eps=5e-8, beta1_power=0.81, beta2_power=0.998, grad_scale=1 / 1024)
        plan = ops.sparse_plan(T(ids, dev))
        sums = ops.segment_sum(plan, T(g, dev), T(sc, dev))
        a = [T(x.copy(), dev) for x in (p0, m0, v0)]
        sa = torch.full((1,), 3.0, dtype=torch.float64, device=dev)
        ops.dense_adam_rows_l2_(a[0], a[1], a[2], plan, sums, l2, sumsq=sa, accumulate=True, **kw)
        b = [T(x.copy(), dev) for x in (p0, m0, v0)]
        gt = torch.zeros((V, D), device=dev)
        ops.scatter_unique_rows_(gt, plan, sums)
        sb = torch.full((1,), 3.0, dtype=torch.float64, device=dev)
        ops.dense_adam_l2_(b[0].view(-1), b[1].view(-1), b[2].view(-1), gt.view(-1), l2, sumsq=sb, accumulate=True, **kw)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert abs(float(sa) - float(sb)) <= 1e-12 * float(sb)
        touched = np.unique(ids[ids < V])
        assert not np.array_equal(a[0].cpu().numpy()[touched], p0[touched])


@pytest.mark.parametrize("B,K5", [(16384, 128), (1000, 64), (77, 8), (4096, 512)])
def test_head_fwd_bwd(dev, oracle, B, K5):
    from mindrec_amd import ops
    rng = np.random.default_rng(K5 + B)
    h4 = torch.from_numpy(np.maximum(rng.standard_normal((B, K5)), 0).astype(np.float32)).to(torch.bfloat16)
    w5 = (rng.standard_normal(K5) * 0.1).astype(np.float32); b5 = np.array([0.05], np.float32)
    wide = rng.standard_normal(B).astype(np.float32) * 0.3
    label = (rng.random(B) < 0.3).astype(np.float32)
    dscale = 1024.0 / B
    dw5 = torch.empty(K5, device=dev); db4 = torch.empty(K5, device=dev); db5 = torch.empty(1, device=dev)
    loss, logit, dlogit, dh4 = ops.head_fwd_bwd(h4.to(dev), T(w5, dev), T(b5, dev), T(wide, dev), T(label, dev), dscale,
                                                dw5, db4, db5)
    r = oracle.head_fwd_bwd(h4.float().numpy(), w5, b5[0], wide, label, dscale)
    assert abs(float(loss) - r["loss"]) <= 1e-5 * abs(r["loss"])
    assert np.allclose(logit.cpu().numpy(), r["logit"], rtol=1e-5, atol=1e-5)
    assert np.allclose(dlogit.cpu().numpy(), r["dlogit"], rtol=1e-4, atol=1e-7)
    assert np.allclose(dh4.cpu().float().numpy(), r["dh4"], rtol=1e-2, atol=1e-6)          # bf16 output
    assert np.array_equal(dh4.cpu().float().numpy() == 0, r["dh4"] == 0)
    s = np.abs(r["dw5"]).max()
    assert np.abs(dw5.cpu().numpy() - r["dw5"]).max() <= 1e-4 * s
    assert np.abs(db4.cpu().numpy() - r["db4"]).max() <= 1e-2 * np.abs(r["db4"]).max()      # sums of bf16-rounded dh4
    assert abs(float(db5) - r["db5"]) <= 1e-4 * max(abs(r["db5"]), 1e-3)


@pytest.mark.parametrize("B,K5", [(16384, 128), (1000, 64), (77, 8)])
def test_head_fwd_bwd_fp32(dev, oracle, B, K5):
    """The fp32 instantiation (h4 in, dh4 out as float32: the mlp_dtype="fp32" net) against the same oracle, at fp32 tolerances."""
    from mindrec_amd import ops
    rng = np.random.default_rng(K5 + B + 1)
    h4 = np.maximum(rng.standard_normal((B, K5)), 0).astype(np.float32)
    w5 = (rng.standard_normal(K5) * 0.1).astype(np.float32); b5 = np.array([0.05], np.float32)
    wide = rng.standard_normal(B).astype(np.float32) * 0.3
    label = (rng.random(B) < 0.3).astype(np.float32)
    dscale = 1.0 / B
    dw5 = torch.empty(K5, device=dev); db4 = torch.empty(K5, device=dev); db5 = torch.empty(1, device=dev)
    loss, logit, dlogit, dh4 = ops.head_fwd_bwd(T(h4, dev), T(w5, dev), T(b5, dev), T(wide, dev), T(label, dev), dscale, dw5, db4, db5)
    assert dh4.dtype == torch.float32
    r = oracle.head_fwd_bwd(h4, w5, b5[0], wide, label, dscale)
    assert abs(float(loss) - r["loss"]) <= 1e-5 * abs(r["loss"])
    assert np.allclose(logit.cpu().numpy(), r["logit"], rtol=1e-5, atol=1e-5)
    assert np.allclose(dlogit.cpu().numpy(), r["dlogit"], rtol=1e-4, atol=1e-9)
    assert np.allclose(dh4.cpu().numpy(), r["dh4"], rtol=1e-4, atol=1e-10)
    assert np.array_equal(dh4.cpu().numpy() == 0, r["dh4"] == 0)
    assert np.abs(dw5.cpu().numpy() - r["dw5"]).max() <= 1e-4 * np.abs(r["dw5"]).max()
    assert np.abs(db4.cpu().numpy() - r["db4"]).max() <= 1e-4 * np.abs(r["db4"]).max()
    assert abs(float(db5) - r["db5"]) <= 1e-4 * max(abs(r["db5"]), 1e-6)


@pytest.mark.parametrize("B,F,D", [(300, 39, 80), (64, 26, 16), (5, 3, 200), (7, 5, 30), (1000, 39, 128)])
def test_fm_term(dev, oracle, B, F, D):
    from mindrec_amd import ops
    rng = np.random.default_rng(B + D)
    vx = rng.standard_normal((B, F, D)).astype(np.float32) * 0.1
    fm, cs = ops.fm_forward(T(vx, dev))
    rfm, rcs = oracle.fm_forward(vx)
    assert np.array_equal(cs.cpu().numpy(), rcs)                          # same sequential field order
    assert np.allclose(fm.cpu().numpy(), rfm, rtol=1e-5, atol=1e-6)
    dout = rng.standard_normal(B).astype(np.float32)
    g0 = rng.standard_normal((B, F, D)).astype(np.float32)
    g = T(g0, dev)
    ops.fm_backward_(g, T(vx, dev), cs, T(dout, dev))
    assert np.allclose(g.cpu().numpy(), g0 + oracle.fm_backward(vx, rcs, dout), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,D,F", [("uniform", 80, 26), ("zipf", 80, 39), ("same", 16, 4), ("hot", 16, 13)])
def test_fused_row_kernels_wide_branch(dev, oracle, dtype, kind, D, F):
    """The wide branch riding the deep kernels over fused rows [p | w accum linear pad | m | v]:
    gather_rows_wide (rows + per-field products), head_fwd_bwd_wide (sum of the products in field order) and
    sparse_lazy_adam_wide_ (LazyAdam + FTRL in one pass) against the oracle's separate restatements."""
    from mindrec_amd import ops
    rng = np.random.default_rng(D * 100 + F)
    V, B = 3000, 512
    ld = -(-(3 * D + 4) // 32) * 32 if kind != "hot" else 3 * D + 4       # 128-byte aligned rows (the engine's), and tight ones
    state = torch.zeros((V, ld), dtype=torch.float32, device=dev)
    p, m, v = state[:, :D], state[:, D + 4:2 * D + 4], state[:, 2 * D + 4:3 * D + 4]
    w, wa, wl = state[:, D:D + 1], state[:, D + 1:D + 2], state[:, D + 2:D + 3]
    ops.fill_normal_(p, seed=5, sigma=0.01); ops.fill_normal_(w, seed=6, sigma=0.01); wa.fill_(1.0)
    rp, rm, rv = oracle.fill_normal(5, V, D, 0.01), np.zeros((V, D), np.float32), np.zeros((V, D), np.float32)
    rw, rwa, rwl = oracle.fill_normal(6, V, 1, 0.01), np.ones((V, 1), np.float32), np.zeros((V, 1), np.float32)
    ids = ids_case(kind, B * F, V, rng, np.int32).reshape(B, F)
    ids[0, 0] = V + 3                                                   # out of range: zero row, zero product, no update
    wts = rng.random((B, F)).astype(np.float32)
    tid, twt = T(ids, dev), T(wts, dev)
    dname = "bf16" if dtype == torch.bfloat16 else "f16"
    emb, wprod = ops.gather_rows_wide(p, tid, twt, D, out_dtype=dtype)
    assert np.array_equal(emb.float().cpu().numpy(), oracle.round16(oracle.gather_rows(rp, ids, wts), dname))
    ref_prod = oracle.gather_rows(rw, ids, wts).reshape(B, F)
    assert np.array_equal(wprod.cpu().numpy()[..., 0], ref_prod) and (wprod[..., 1] == 0).all()
    # head: the sum of the products inside the head == the head on the oracle's wide sum, bit for bit
    K5 = 128
    h4 = T(np.maximum(rng.standard_normal((B, K5)), 0).astype(np.float32), dev).to(dtype)
    w5 = T((rng.standard_normal(K5) * 0.1).astype(np.float32), dev); b5 = T(np.array([0.3], np.float32), dev)
    wb = T(np.array([-0.2], np.float32), dev)
    label = T((rng.random(B) < 0.3).astype(np.float32), dev)
    outs = []
    for mode in ("prod", "sum"):
        dw5 = torch.zeros(K5, device=dev); db4 = torch.zeros(K5, device=dev); db5 = torch.zeros(1, device=dev)
        if mode == "prod":
            r = ops.head_fwd_bwd_wide(h4, w5, b5, wprod.contiguous(), wb, label, 2.0, dw5, db4, db5)
        else:
            wide = T(oracle.wide_sum(rw, ids, wts, -0.2), dev)
            r = ops.head_fwd_bwd(h4, w5, b5, wide, label, 2.0, dw5, db4, db5)
        outs.append([t.clone() for t in r] + [dw5, db4, db5])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    # apply: deep LazyAdam + wide FTRL in one pass vs the oracle's two applies
    g = (rng.standard_normal((B * F, D)) * 1.024).astype(np.float32)
    g16 = oracle.round16(g, dname)
    gw = (rng.standard_normal(B) * 1.024).astype(np.float32)
    plan = ops.sparse_plan(tid)
    ops.sparse_lazy_adam_wide_(p, m, v, plan, T(g16, dev).to(dtype), twt, T(gw, dev), F, D, grad_scale=1 / 1024)
    oracle.sparse_lazy_adam(rp, rm, rv, ids, g16, wts, grad_scale=1 / 1024)
    oracle.sparse_ftrl(rw, rwa, rwl, ids, np.repeat(gw, F).reshape(B * F, 1), wts, grad_scale=1 / 1024)
    cross = plan.uniq.cpu().numpy()[crossing(plan, D)]
    inwin = np.ones(V, bool)
    inwin[cross[(cross >= 0) & (cross < V)]] = False                    # rows whose run stays inside one window: bit-exact
    gp, gw_, gwa, gwl = p.cpu().numpy(), w.cpu().numpy(), wa.cpu().numpy(), wl.cpu().numpy()
    assert np.array_equal(gp[inwin], rp[inwin]) and np.array_equal(m.cpu().numpy()[inwin], rm[inwin])
    assert np.array_equal(gw_[inwin], rw[inwin]) and np.array_equal(gwa[inwin], rwa[inwin]) and np.array_equal(gwl[inwin], rwl[inwin])
    assert inwin.sum() > V // 2
    den = np.maximum(np.abs(rp).max(axis=1), 1e-30)
    assert float((np.abs(gp - rp).max(axis=1) / den).max()) <= 1e-5
    assert np.abs(gw_ - rw).max() <= 1e-4 * np.abs(rw).max() and np.allclose(gwa, rwa, rtol=1e-5)
    assert (state[:, D + 3] == 0).all() and (state[:, 3 * D + 4:] == 0).all()        # the pad words are never written


def test_shard_messages_pack_unpack_and_column_windows(dev, oracle):
    """The merged shard messages: (id, weight) pairs; permutations that write into / read from column windows of wider
    message rows; the owner's one-pass answer rows [D 16-bit values | wide weight * mask, 0 | pad]; the folded apply with one
    wide gradient per position read as a column of the gradient message (F = 1)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(21)
    n, S, D = 9001, 4, 80
    ids = rng.integers(0, 50000, size=n).astype(np.int32)
    wts = rng.random(n).astype(np.float32)
    loc, perm, counts = ops.shard_route(T(ids, dev), S)
    rperm = perm.cpu().numpy()
    pairs = ops.shard_pack_iw(loc, T(wts, dev), perm)
    pk = pairs.cpu().numpy()
    assert np.array_equal(pk[:, 0], loc.cpu().numpy()) and np.array_equal(pk[:, 1].view(np.float32), wts[rperm])
    i2, w2 = ops.shard_unpack_iw(pairs)
    assert np.array_equal(i2.cpu().numpy(), loc.cpu().numpy()) and np.array_equal(w2.cpu().numpy(), wts[rperm])
    # route_rows into column windows of one message, unroute out of them
    W = D // 2 + 4
    g = rng.standard_normal((n, D // 2)).astype(np.float32)
    gw = rng.standard_normal((n, 1)).astype(np.float32)
    msg = torch.full((n, W), 7.0, device=dev)
    ops.shard_route_rows(T(g, dev), perm, None, out=msg)
    ops.shard_route_rows(T(gw, dev), perm, None, out=msg[:, D // 2:])
    m = msg.cpu().numpy()
    assert np.array_equal(m[:, : D // 2], g[rperm]) and np.array_equal(m[:, D // 2], gw[rperm, 0]) and (m[:, D // 2 + 1:] == 7.0).all()
    back = ops.shard_unroute(msg, perm, None, cols=D // 2).cpu().numpy()
    assert np.array_equal(back, g)
    backw = ops.shard_unroute(msg[:, D // 2:], perm, None, cols=1).cpu().numpy()
    assert np.array_equal(backw, gw)
    # the owner's answer rows, one pass over fused rows
    V = 4000
    ld = -(-(3 * D + 4) // 32) * 32
    state = torch.zeros((V, ld), dtype=torch.float32, device=dev)
    p, w = state[:, :D], state[:, D:D + 1]
    ops.fill_normal_(p, seed=5, sigma=0.01); ops.fill_normal_(w, seed=6, sigma=0.01)
    rp, rw = oracle.fill_normal(5, V, D, 0.01), oracle.fill_normal(6, V, 1, 0.01)
    rid = rng.integers(0, V, size=n).astype(np.int32)
    ans = ops.gather_rows_wide(p, T(rid, dev), T(wts, dev), D, out_dtype=torch.bfloat16, packed_words=W)
    assert tuple(ans.shape) == (n, W)
    rows16 = ans.view(torch.bfloat16)[:, :D].float().cpu().numpy()
    assert np.array_equal(rows16, oracle.round16(oracle.gather_rows(rp, rid, wts), "bf16"))
    a = ans.cpu().numpy()
    assert np.array_equal(a[:, D // 2], oracle.gather_rows(rw, rid, wts)[:, 0]) and (a[:, D // 2 + 1] == 0).all()
    # folded apply with per-position wide gradients == the two separate applies
    m_, v_ = state[:, D + 4:2 * D + 4], state[:, 2 * D + 4:3 * D + 4]
    wa, wl = state[:, D + 1:D + 2], state[:, D + 2:D + 3]
    wa.fill_(1.0)
    ref = state.clone()
    plan = ops.sparse_plan(T(rid, dev))
    gmsg = torch.zeros((n, W), device=dev)
    g16 = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).to(dev).to(torch.bfloat16)
    gmsg.view(torch.bfloat16)[:, :D] = g16
    gmsg[:, D // 2] = T(gw[:, 0], dev)
    ops.sparse_lazy_adam_wide_(p, m_, v_, plan, gmsg.view(torch.bfloat16)[:, :D], T(wts, dev), gmsg[:, D // 2:D // 2 + 1], 1, D,
                               grad_scale=1 / 64)
    rp2, rm2, rv2 = ref[:, :D], ref[:, D + 4:2 * D + 4], ref[:, 2 * D + 4:3 * D + 4]
    ops.sparse_lazy_adam_(rp2, rm2, rv2, plan, g16, T(wts, dev), grad_scale=1 / 64)
    ops.sparse_ftrl_(ref[:, D:D + 1], ref[:, D + 1:D + 2], ref[:, D + 2:D + 3], plan, T(gw, dev), T(wts, dev), grad_scale=1 / 64)
    assert torch.equal(state[:, :D], ref[:, :D]) and torch.equal(state[:, D + 4:], ref[:, D + 4:])
    assert float((state[:, D:D + 3] - ref[:, D:D + 3]).abs().max()) <= 1e-6 * float(ref[:, D:D + 3].abs().max())


def test_sum_slab_segments(dev):
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    flat = torch.full((5000,), -1.0, device=dev)
    segs = [(0, torch.from_numpy(rng.standard_normal((7, 40, 20)).astype(np.float32)).to(dev)),
            (1000, torch.from_numpy(rng.standard_normal((64, 128)).astype(np.float32)).to(dev)),
            (2000, torch.from_numpy(rng.standard_normal((1, 16, 4)).astype(np.float32)).to(dev))]
    ops.sum_slab_segments_(flat, segs)
    out = flat.cpu().numpy()
    for start, t in segs:
        a = t.cpu().numpy()
        acc = a[0].copy()
        for s in range(1, a.shape[0]):
            acc = acc + a[s]                                    # slab order
        assert np.array_equal(out[start:start + acc.size], acc.ravel())
    assert (out[800:1000] == -1).all() and (out[1128:2000] == -1).all() and (out[2064:] == -1).all()


def test_committed_goldens_replay_on_the_gpu(dev):
    """tests/golden/normal_seed1000.json and wd_step_small.npz (committed bits; tests/test_oracle.py holds the oracle to them) through
    the HIP kernels: the table initialiser to the bit, the lookup to the bit, one LazyAdam step on the touched rows -- to the bit
    where an id's duplicates sit inside one window of the sorted index, 1e-5 row-relative otherwise."""
    import json
    import os
    from mindrec_amd import ops
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gold = json.load(open(os.path.join(gold_dir, "normal_seed1000.json")))
    rows = np.array(gold["rows"], np.int64)
    got = []
    for r in rows.tolist():                     # (the golden rows reach far past any table: one row each, keyed by its global number)
        t = ops.fill_normal_(torch.empty((1, gold["D"]), dtype=torch.float32, device=dev), gold["seed"], gold["sigma"], row0=int(r))
        got += t.cpu().numpy().view(np.uint32).ravel().tolist()
    assert got == gold["bits"]
    z = np.load(os.path.join(gold_dir, "wd_step_small.npz"))
    V, D = int(z["V"]), int(z["D"])
    p = ops.fill_normal_(torch.empty((V, D), dtype=torch.float32, device=dev), int(z["seed"]), 0.01)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ids, wts = T(z["ids"], dev), T(z["wts"], dev)
    assert np.array_equal(ops.gather_rows(p, ids, wts).cpu().numpy().reshape(z["emb"].shape), z["emb"])
    plan = ops.sparse_plan(ids)
    ops.sparse_lazy_adam_(p, m, v, plan, T(z["g"].reshape(-1, D), dev), wts.reshape(-1), lr=3.5e-4, eps=1e-8, beta1_power=0.9,
                          beta2_power=0.999, grad_scale=1 / 1024)
    touched = np.unique(z["ids"])
    got_p, got_m = p.cpu().numpy()[touched], m.cpu().numpy()[touched]
    cross = set(plan.uniq.cpu().numpy()[crossing(plan, D)].tolist())
    inside = np.array([r not in cross for r in touched])
    assert np.array_equal(got_p[inside], z["p_touched"][inside]) and np.array_equal(got_m[inside], z["m_touched"][inside])
    den = np.abs(z["p_touched"]).max(axis=1) + 1e-30
    assert float((np.abs(got_p - z["p_touched"]).max(axis=1) / den).max()) <= 1e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fm_forward_16bit_copy_is_the_rounded_input(dev, dt):
    """ops.fm_forward(out16=...): the FM term's pass also writes its input rounded to 16 bits (the dense net's input, which DeepFM used to
    look up a second time) -- the same bits as a cast of the tensor, the FM outputs untouched."""
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    B, F, D = 1000, 39, 80
    vx = T((rng.standard_normal((B, F, D)) * 0.05).astype(np.float32), dev)
    add = T(rng.standard_normal(B).astype(np.float32), dev)
    fm0, cs0 = ops.fm_forward(vx, add=add)
    out = torch.empty((B, F, D), dtype=dt, device=dev)
    fm1, cs1 = ops.fm_forward(vx, add=add, out16=out)
    assert torch.equal(fm0, fm1) and torch.equal(cs0, cs1)
    assert torch.equal(out.view(torch.int16), vx.to(dt).view(torch.int16))
