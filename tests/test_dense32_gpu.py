"""DenseLayer in fp32 on v_mfma_f32_32x32x2_f32 (csrc/mrec_gemm_f32.hip) and the Deep&Cross output end (csrc/mrec_dcn.hip)
through the C ABI against float64 numpy restatements (reference: DenseLayer with convert_dtype=False,
models/deep_and_cross/src/deep_and_cross.py:94-114; output layer + loss :306-309,326-331).  The fp32 matrix instruction is a
k-ordered chain of fmaf's: the tolerance is the fp32 accumulation bound of a K-term dot product, stated at each assert."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bound(x, w, K):
    """|fl(sum) - sum| <= K * 2^-24 * sum |x| |w| (any order of K fp32 fma's), with a little headroom."""
    return 1.5 * K * 2.0 ** -24 * (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)) + 1e-30


SHAPES = [(512, 1170, 1024), (300, 64, 32), (16384, 1024, 1024), (1000, 1170, 200), (129, 34, 130), (128, 31, 7)]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_dense32_forward_backward_vs_float64(dev, M, K, N):
    from mindrec_amd import ops
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    tx, tw, tb = T(x, dev), T(w, dev), T(b, dev)
    # forward: bias + ReLU epilogue
    y = ops.dense32_fwd(tx, tw, tb, relu=True).cpu().numpy()
    pre = x.astype(np.float64) @ w.astype(np.float64) + b
    tol = _bound(x, w, K) + 2.0 ** -23 * np.abs(pre)
    ref = np.maximum(pre, 0.0)
    near0 = np.abs(pre) <= tol                       # the ReLU's corner: either side is right
    assert (np.abs(y - ref) <= tol)[~near0].all() and (np.abs(y) <= 2 * tol)[near0].all()
    y_lin = ops.dense32_fwd(tx, tw, None, relu=False).cpu().numpy()
    assert (np.abs(y_lin - (pre - b)) <= tol).all()
    # input gradient: mask by h > 0, per-tile column sums
    dy = rng.standard_normal((M, N)).astype(np.float32)
    h = rng.standard_normal((M, K)).astype(np.float32)
    cs = torch.empty((ops.dense32_colsum_tiles(M), K), dtype=torch.float32, device=dev)
    dx = ops.dense32_bwd_input(T(dy, dev), tw, h=T(h, dev), colsum=cs).cpu().numpy()
    full = dy.astype(np.float64) @ w.astype(np.float64).T
    tol_dx = _bound(dy, w.T, N) + 2.0 ** -23 * np.abs(full)
    assert (np.abs(dx - np.where(h > 0, full, 0.0)) <= tol_dx).all() and (dx[h <= 0] == 0).all()
    got_cs = cs.cpu().numpy().astype(np.float64).sum(axis=0)
    assert np.allclose(got_cs, dx.astype(np.float64).sum(axis=0), rtol=1e-5, atol=1e-5 * np.abs(dx).sum(axis=0).max())
    dx_nomask = ops.dense32_bwd_input(T(dy, dev), tw).cpu().numpy()
    assert (np.abs(dx_nomask - full) <= tol_dx).all()
    # weight gradient: fp32 batch slabs
    S = ops.dense32_bwd_weight_slabs(M, K, N)
    for s_ in sorted({1, S}):
        slabs = torch.full((s_, K, N), float("nan"), dtype=torch.float32, device=dev)
        ops.dense32_bwd_weight(tx, T(dy, dev), slabs)
        dw = slabs.cpu().numpy().astype(np.float64).sum(axis=0)
        ref_dw = x.astype(np.float64).T @ dy.astype(np.float64)
        assert (np.abs(dw - ref_dw) <= _bound(x.T, dy, M) + 2.0 ** -22 * np.abs(ref_dw)).all(), s_


def test_dense32_strided_rows_and_views(dev):
    """Row strides: an operand that is a column window of a wider buffer (8-byte and 4-byte aligned rows)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(5)
    M, K, N = 200, 70, 50
    big = T(rng.standard_normal((M, K + 7)).astype(np.float32), dev)
    w = T((rng.standard_normal((K, N)) * 0.1).astype(np.float32), dev)
    for off in (0, 1, 2):
        x = big[:, off:off + K]
        out = torch.zeros((M, N + 3), dtype=torch.float32, device=dev)
        ops.dense32_fwd(x, w, None, relu=False, out=out[:, 1:1 + N])
        ref = x.double() @ w.double()
        assert float((out[:, 1:1 + N].double() - ref).abs().max()) <= 1e-4 and float(out[:, 0].abs().max()) == 0.0 and float(out[:, N + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,X", [(16384, 1024, 1170), (100, 32, 1170), (37, 64, 30)])
def test_dcn_head_vs_float64(dev, B, H, X):
    from mindrec_amd import ops
    rng = np.random.default_rng(B)
    d2 = np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32)           # a ReLU output: about half zeros
    c = rng.standard_normal((B, X)).astype(np.float32)
    w3 = (rng.standard_normal(H + X) * 0.03).astype(np.float32)
    b3 = np.array([0.1], np.float32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    dscale = 1000.0 / B
    dw3, db2, db3 = (torch.empty(n, dtype=torch.float32, device=dev) for n in (H + X, H, 1))
    loss, logit, dd2, dc = ops.dcn_head_fwd_bwd(T(d2, dev), T(c, dev), T(w3, dev), T(b3, dev), T(y, dev), dscale, dw3, db2, db3)
    z = d2.astype(np.float64) @ w3[:H] + c.astype(np.float64) @ w3[H:] + 0.1
    assert np.allclose(logit.cpu().numpy(), z, rtol=1e-5, atol=1e-5)
    ref_loss = (np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))).mean()
    assert abs(float(loss) - ref_loss) <= 1e-5 * ref_loss
    dl = (1.0 / (1.0 + np.exp(-z)) - y) * dscale
    assert np.allclose(dd2.cpu().numpy(), np.where(d2 > 0, dl[:, None] * w3[None, :H], 0.0), rtol=1e-4, atol=1e-7 * dscale)
    assert np.allclose(dc.cpu().numpy(), dl[:, None] * w3[None, H:], rtol=1e-4, atol=1e-7 * dscale)
    ref_dw3 = np.concatenate([d2.astype(np.float64).T @ dl, c.astype(np.float64).T @ dl])
    scale = np.abs(np.concatenate([d2, c], axis=1)).astype(np.float64).T @ np.abs(dl)
    assert (np.abs(dw3.cpu().numpy() - ref_dw3) <= 1e-5 * scale + 1e-9).all()
    assert abs(float(db3) - dl.sum()) <= 1e-5 * np.abs(dl).sum()
    ref_db2 = np.where(d2 > 0, dl[:, None] * w3[None, :H], 0.0).sum(axis=0)
    assert np.allclose(db2.cpu().numpy(), ref_db2, rtol=1e-4, atol=1e-5 * np.abs(ref_db2).max())
    # reproducible: fixed summation order
    dw3b, db2b, db3b = (torch.empty(n, dtype=torch.float32, device=dev) for n in (H + X, H, 1))
    ops.dcn_head_fwd_bwd(T(d2, dev), T(c, dev), T(w3, dev), T(b3, dev), T(y, dev), dscale, dw3b, db2b, db3b)
    assert torch.equal(dw3, dw3b) and torch.equal(db2, db2b)


@pytest.mark.timeout(900)
def test_deep_cross_engine_at_the_configurations_batch(dev, oracle):
    """BASELINE configs[2]: batch 16384, 39 fields x 30, DenseLayer 1170-1024-1024, 6 cross layers, all fp32 -- the hand-written
    step (no library GEMM, no autograd; from the third step on ONE HIP graph) against the oracle-side engine (torch restatement
    over the oracle's kernels), 4 steps."""
    from _oracle_engine import OracleDeepCrossEngine
    from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine
    from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
    cfg = DeepCrossConfig(vocab_size=200000, batch_size=16384)
    g, c = DeepCrossEngine(cfg, dev), OracleDeepCrossEngine(cfg, "cpu")
    assert g._native and torch.equal(g.dense_flat.detach().cpu(), c.dense_flat.detach())
    bcfg = WideDeepConfig(vocab_size=cfg.vocab_size, emb_dim=30, field_size=39, batch_size=16384)
    for s in range(4):
        ids, wts, label = synthetic_batch(bcfg, "cpu", "zipf", seed=31 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 2e-5 * max(abs(lc), 1e-3), (s, lc, lg)
    assert g._graph is not None                      # steps 3 and 4 replayed the captured step
    a, b = g.table.cpu().numpy(), c.table.numpy()
    den = np.maximum(np.abs(b).max(axis=1), 1e-30)
    assert float((np.abs(a - b).max(axis=1) / den).max()) <= 1e-4
    d = np.abs(g.dense_flat.detach().cpu().numpy() - c.dense_flat.detach().numpy())
    # Adam: parameters whose gradient is ~0 may step +-lr either way; everything else agrees closely
    assert d.max() <= 2 * cfg.learning_rate * 4 and np.mean(d <= 1e-2 * cfg.learning_rate + 2e-4 * np.abs(c.dense_flat.detach().numpy())) > 0.995
