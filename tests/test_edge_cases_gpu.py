"""Edge cases through the C ABI on the GPU: empty inputs on every op, workspace too small, wrong dtypes,
single-element and all-duplicate batches, ids at the table bounds."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_inputs_everywhere(dev):
    from mindrec_amd import ops
    e32 = torch.empty((0, 5), dtype=torch.int32, device=dev)
    table = torch.ones((10, 8), device=dev)
    assert ops.gather_rows(table, e32).shape == (0, 5, 8)
    assert ops.gather_rows(table, e32, out_dtype=torch.bfloat16).shape == (0, 5, 8)
    assert ops.wide_sum(table[:, :1], torch.empty((0, 3), dtype=torch.int32, device=dev),
                        torch.empty((0, 3), device=dev)).shape == (0,)
    plan = ops.sparse_plan(e32)
    assert plan.U == 0 and plan.n == 0
    m = torch.zeros_like(table); v = torch.zeros_like(table)
    ops.sparse_lazy_adam_(table, m, v, plan, torch.empty((0, 8), device=dev))
    ops.sparse_ftrl_(table, m, v, plan, torch.empty((0, 8), device=dev))
    assert float((table - 1).abs().max()) == 0.0                      # nothing moved
    assert ops.segment_sum(plan, torch.empty((0, 8), device=dev)).shape[1] == 8
    loc, perm, counts = ops.shard_route(e32, 4)
    assert loc.numel() == 0 and counts.tolist() == [0, 0, 0, 0]
    ki = ops.KeyIndex(16, dev)
    rows, new = ki.find_or_insert(torch.empty(0, dtype=torch.int64, device=dev))
    assert rows.numel() == 0 and len(ki) == 0
    ki.erase(torch.empty(0, dtype=torch.int64, device=dev))
    k, r = ki.export()
    assert k.numel() == 0
    ops.dense_adam_(torch.empty(0, device=dev), torch.empty(0, device=dev), torch.empty(0, device=dev), torch.empty(0, device=dev))
    assert ops.cross_layers(torch.empty((0, 16), device=dev), torch.ones((2, 16), device=dev), torch.ones((2, 16), device=dev)).shape == (0, 16)


def test_single_id_and_table_bounds(dev, oracle):
    from mindrec_amd import ops
    V, D = 7, 80
    table = torch.arange(V * D, dtype=torch.float32, device=dev).view(V, D)
    ids = torch.tensor([[V - 1]], dtype=torch.int64, device=dev)
    assert torch.equal(ops.gather_rows(table, ids)[0, 0], table[V - 1])
    edge = torch.tensor([0, V - 1, V, -1, 2**31 - 1, -2**31], dtype=torch.int32, device=dev)
    out = ops.gather_rows(table, edge)
    assert torch.equal(out[0], table[0]) and torch.equal(out[1], table[V - 1]) and float(out[2:].abs().max()) == 0.0
    big = torch.tensor([2**40, -2**40, 3], dtype=torch.int64, device=dev)
    out = ops.gather_rows(table, big)
    assert float(out[:2].abs().max()) == 0.0 and torch.equal(out[2], table[3])
    # one id, applied: only that row moves
    p = torch.zeros((V, D), device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    ops.sparse_lazy_adam_(p, m, v, ops.sparse_plan(ids), torch.ones((1, D), device=dev))
    assert float(p[:V - 1].abs().max()) == 0.0 and float(p[V - 1].abs().min()) > 0.0


def test_workspace_too_small_is_reported(dev):
    from mindrec_amd import _lib
    l = _lib.lib()
    n = 5000
    ids = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev)
    uniq = torch.empty(n, dtype=torch.int32, device=dev); inv = torch.empty(n, dtype=torch.int32, device=dev)
    nu = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert l.mrec_dedup_i32(vp(ids), n, vp(uniq), vp(inv), vp(nu), vp(ws), ws.numel(), st) == -2     # MREC_EWORKSPACE
    assert l.mrec_group_by_inverse(vp(inv), n, vp(uniq), vp(uniq), vp(inv), vp(ws), ws.numel(), st) == -2
    p = torch.zeros((100, 8), device=dev)
    g = torch.zeros((n, 8), device=dev)
    assert l.mrec_sparse_lazy_adam_f32_i32(vp(p), vp(p), vp(p), 100, 8, 8, vp(uniq), vp(inv), vp(inv), vp(inv), n, vp(g), 8,
                                           None, 0.1, 0.9, 0.999, 1e-8, 0.9, 0.999, 1.0, 0, vp(ws), ws.numel(), st) == -2
    torch.cuda.synchronize()


def test_wrong_dtypes_raise(dev):
    from mindrec_amd import ops
    table = torch.zeros((4, 4), device=dev)
    with pytest.raises(TypeError):
        ops.gather_rows(table, torch.zeros(2, dtype=torch.int16, device=dev))
    with pytest.raises(TypeError):
        ops.gather_rows(table.double(), torch.zeros(2, dtype=torch.int32, device=dev))
    with pytest.raises(TypeError):
        ops.gather_rows(table, torch.zeros(2, dtype=torch.int32, device=dev), row_scale=torch.zeros(3, device=dev))
    with pytest.raises(TypeError):
        ops.sparse_lazy_adam_(table, table.clone(), table.clone(), ops.sparse_plan(torch.zeros(2, dtype=torch.int32, device=dev)),
                              torch.zeros((2, 4), dtype=torch.float64, device=dev))
    with pytest.raises(ValueError):
        ops.sparse_lazy_adam_(table, torch.zeros((4, 8), device=dev)[:, :4], table.clone(),
                              ops.sparse_plan(torch.zeros(2, dtype=torch.int32, device=dev)), torch.zeros((2, 4), device=dev))
    with pytest.raises(TypeError):
        ops.unique(torch.zeros(3, device=dev))


def test_all_duplicates_and_max_run(dev, oracle):
    """Every id identical: one group whose run spans every window (the carry pass does all the summing)."""
    from mindrec_amd import ops
    n, D = 50_000, 16
    ids = torch.full((n,), 3, dtype=torch.int32, device=dev)
    g = torch.ones((n, D), device=dev)
    plan = ops.sparse_plan(ids)
    assert plan.U == 1
    out = ops.segment_sum(plan, g)[:1]
    assert torch.equal(out, torch.full((1, D), float(n), device=dev))           # integers: exact in any order


def test_cross_and_fm_degenerate_shapes(dev, oracle):
    """Cross stack with no layers / one row / D not a multiple of 64 at the buffer-resource boundary; FM term with one
    sample and one field."""
    from mindrec_amd import ops
    rng = np.random.default_rng(4)
    # L = 0: identity forward, dx0 = dy backward
    x0 = torch.from_numpy(rng.standard_normal((5, 70)).astype(np.float32)).to(dev)
    w = torch.zeros((0, 70), device=dev); b = torch.zeros((0, 70), device=dev)
    assert torch.equal(ops.cross_layers(x0, w, b), x0)
    dy = torch.from_numpy(rng.standard_normal((5, 70)).astype(np.float32)).to(dev)
    dx0, dw, db = ops.cross_layers_bwd(x0, w, b, dy)
    assert torch.equal(dx0, dy) and dw.numel() == 0 and db.numel() == 0
    # one row, D = 65 (one full 64-lane pass + a single column in the second), 2 layers
    for B, D, L in ((1, 65, 2), (3, 2048, 1), (2, 1, 3)):
        x = rng.standard_normal((B, D)).astype(np.float32)
        ww = (rng.standard_normal((L, D)) / np.sqrt(D)).astype(np.float32)
        bb = (rng.standard_normal((L, D)) * 0.1).astype(np.float32)
        out = ops.cross_layers(torch.from_numpy(x).to(dev), torch.from_numpy(ww).to(dev), torch.from_numpy(bb).to(dev))
        assert np.allclose(out.cpu().numpy(), oracle.cross_layers(x, ww, bb), rtol=1e-5, atol=1e-5), (B, D, L)
        g = rng.standard_normal((B, D)).astype(np.float32)
        dx, dww, dbb = ops.cross_layers_bwd(torch.from_numpy(x).to(dev), torch.from_numpy(ww).to(dev),
                                            torch.from_numpy(bb).to(dev), torch.from_numpy(g).to(dev))
        rdx, rdw, rdb = oracle.cross_layers_bwd(x, ww, bb, g)
        assert np.allclose(dx.cpu().numpy(), rdx, rtol=1e-4, atol=1e-4), (B, D, L)
        assert np.allclose(dww.cpu().numpy(), rdw, rtol=1e-4, atol=1e-4) and np.allclose(dbb.cpu().numpy(), rdb, rtol=1e-4, atol=1e-4)
    with pytest.raises(Exception):
        ops.cross_layers(torch.zeros((2, 4096), device=dev), torch.zeros((1, 4096), device=dev), torch.zeros((1, 4096), device=dev))
    # FM: one sample, one field -> (x^2 - x^2) / 2 = 0 and colsum = x
    vx = torch.from_numpy(rng.standard_normal((1, 1, 16)).astype(np.float32)).to(dev)
    fm, cs = ops.fm_forward(vx)
    assert float(fm.abs().max()) == 0.0 and torch.equal(cs.view(-1), vx.view(-1))
