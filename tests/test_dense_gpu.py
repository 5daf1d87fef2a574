"""DenseLayer on the matrix cores (csrc/mrec_dense.hip, through the C ABI) against the oracle's mixed-precision
restatement (oracle/oracle.py: dense_layer / dense_bwd_input / dense_bwd_weight; reference DenseLayer.construct,
models/wide_deep/src/wide_and_deep.py:113-133).

Bars, written out: the kernels accumulate in fp32 on the MFMA units, the oracle in float64.  With 16-bit operands the
products are exact in fp32, so the two sums differ by at most K * 2^-24 * sum|x||w| (any-order fp32 summation bound);
both results are then rounded to 16 bits, so they can land on the two sides of a rounding boundary: one 16-bit ulp.
Outputs: |gpu - oracle| <= ulp16(|oracle|) + 2 K 2^-24 (|x| . |w|), elementwise, and > 98 % of the elements are equal.  Weight gradients (fp32 out): |gpu - oracle| <= 2 M 2^-24 (|x|^T . |dy|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
EPS16 = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}      # one ulp relative to the value (upper bound): 2^-(mantissa bits)
TINY = {"bf16": 1e-38, "f16": 2.0 ** -24}           # f16 subnormal spacing


def _vals(rng, shape, scale, dtype, oracle):
    return oracle.round16((rng.standard_normal(shape) * scale).astype(np.float32), dtype)


def _dev(a, dtype, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(DT[dtype])


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N,relu", [(512, 2080, 1024, True), (300, 3120, 264, True), (777, 72, 40, False), (64, 128, 8, True),
                                        (1000, 1024, 512, True)])
def test_dense_fwd_matches_oracle(dev, oracle, dtype, M, K, N, relu):
    from mindrec_amd import ops
    rng = np.random.default_rng(M + K + N)
    x = _vals(rng, (M, K), 1.0, dtype, oracle)
    x[rng.random((M, K)) < 0.3] = 0.0                      # post-ReLU activations are sparse
    w = _vals(rng, (K, N), 0.05, dtype, oracle)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    y = ops.dense_fwd(_dev(x, dtype, dev), _dev(w, dtype, dev), torch.from_numpy(b).to(dev), relu=relu).float().cpu().numpy()
    ref = oracle.dense_layer(x, w, b, relu, dtype)
    bound = EPS16[dtype] * np.abs(ref) + TINY[dtype] + 2 * K * 2.0 ** -24 * (np.abs(x).astype(np.float64) @ np.abs(w) + np.abs(b))
    assert np.all(np.abs(y.astype(np.float64) - ref) <= bound), float((np.abs(y - ref) / bound).max())
    assert np.mean(y == ref) > 0.98                        # and almost every element is the exactly rounded value
    if relu:
        assert (y >= 0).all() and (y == 0).mean() > 0.2


def test_dense_fwd_strided_and_no_bias(dev, oracle):
    """Row strides larger than the width (views into wider buffers) and a missing bias."""
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    M, K, N = 260, 160, 72
    x = _vals(rng, (M, K), 1.0, "bf16", oracle); w = _vals(rng, (K, N), 0.1, "bf16", oracle)
    xb = torch.zeros((M, K + 24), dtype=torch.bfloat16, device=dev)
    xb[:, :K] = _dev(x, "bf16", dev)
    out = torch.full((M, N + 8), 7.0, dtype=torch.bfloat16, device=dev)
    ops.dense_fwd(xb[:, :K], _dev(w, "bf16", dev), None, relu=False, out=out[:, :N])
    ref = oracle.dense_layer(x, w, None, False, "bf16")
    got = out.float().cpu().numpy()
    assert np.abs(got[:, :N] - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
    assert (got[:, N:] == 7.0).all()                       # nothing written past the row


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N,mask", [(512, 1024, 512, True), (300, 3120, 136, False), (1000, 200, 72, True), (64, 8, 32, True)])
def test_dense_bwd_input_matches_oracle(dev, oracle, dtype, M, K, N, mask):
    from mindrec_amd import ops
    rng = np.random.default_rng(M * 3 + K + N)
    dy = _vals(rng, (M, N), 1e-2, dtype, oracle)
    w = _vals(rng, (K, N), 0.05, dtype, oracle)
    h = np.maximum(_vals(rng, (M, K), 1.0, dtype, oracle), 0)
    db = torch.full((K,), 99.0, dtype=torch.float32, device=dev)
    dx = ops.dense_bwd_input(_dev(dy, dtype, dev), _dev(w, dtype, dev), h=_dev(h, dtype, dev) if mask else None,
                             db_out=db if mask else None).float().cpu().numpy()
    ref, ref_db = oracle.dense_bwd_input(dy, w, h if mask else None, dtype)
    bound = EPS16[dtype] * np.abs(ref) + TINY[dtype] + 2 * N * 2.0 ** -24 * (np.abs(dy).astype(np.float64) @ np.abs(w).T)
    assert np.all(np.abs(dx.astype(np.float64) - ref) <= bound)
    if mask:
        assert np.array_equal(dx == 0, (ref == 0) | (dx == 0)) and ((h > 0) | (dx == 0)).all()
        # bias gradient: fp32 sum of the ROUNDED dx in a fixed order; against the float64 sum of the GPU's own dx
        own = dx.astype(np.float64).sum(axis=0)
        assert np.allclose(db.cpu().numpy(), own, rtol=0, atol=M * 2.0 ** -24 * np.abs(dx).sum(axis=0).max() + 1e-30)
        assert np.allclose(db.cpu().numpy(), ref_db, rtol=1e-2, atol=1e-3 * np.abs(ref_db).max())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(2048, 2080, 1024), (1000, 3120, 264), (96, 264, 72), (50, 8, 8), (16384, 256, 128)])
def test_dense_bwd_weight_matches_oracle(dev, oracle, dtype, M, K, N):
    from mindrec_amd import ops
    rng = np.random.default_rng(M + 7 * K + N)
    x = np.maximum(_vals(rng, (M, K), 1.0, dtype, oracle), 0)
    dy = _vals(rng, (M, N), 1e-2, dtype, oracle)
    S = ops.dense_bwd_weight_slabs(M, K, N)
    slabs = torch.full((S, K, N), float("nan"), dtype=torch.float32, device=dev)
    ops.dense_bwd_weight(_dev(x, dtype, dev), _dev(dy, dtype, dev), slabs)
    out = torch.empty((K, N), dtype=torch.float32, device=dev)
    ops.sum_slabs(slabs, out)
    got = out.cpu().numpy().astype(np.float64)
    ref = oracle.dense_bwd_weight(x, dy)
    bound = 2 * M * 2.0 ** -24 * (np.abs(x).astype(np.float64).T @ np.abs(dy)) + 1e-30
    assert np.all(np.abs(got - ref) <= bound), float((np.abs(got - ref) / bound).max())
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    # slabs are added in slab order: the same bits as a sequential fp32 sum on the host
    sl = slabs.cpu().numpy()
    acc = sl[0].copy()
    for s_ in range(1, S):
        acc = acc + sl[s_]
    assert np.array_equal(out.cpu().numpy(), acc)


def test_dense_adam_slabs_equals_plain_adam_on_summed_gradient(dev, oracle):
    """The dense Adam that adds fp32 slabs inside the kernel == the oracle's Adam on the slab sums (same fp32 add
    order), bit for bit, and writes the bf16 / fp16 operand shadow."""
    from mindrec_amd import ops
    rng = np.random.default_rng(11)
    n, seg0, len0, seg1, len1 = 4096, 256, 1024, 2048, 512
    p = (rng.standard_normal(n) * 0.01).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    s0 = rng.standard_normal((5, len0)).astype(np.float32); s1 = rng.standard_normal((3, len1)).astype(np.float32)
    gsum = g.copy()
    acc = s0[0].copy()
    for s in range(1, 5):
        acc = acc + s0[s]
    gsum[seg0:seg0 + len0] = acc
    acc = s1[0].copy()
    for s in range(1, 3):
        acc = acc + s1[s]
    gsum[seg1:seg1 + len1] = acc
    for sh_dt in (torch.bfloat16, torch.float16):
        tp, tm, tv = (torch.from_numpy(a.copy()).to(dev) for a in (p, m, v))
        shadow = torch.zeros(n, dtype=sh_dt, device=dev)
        ops.dense_adam_slabs_(tp, tm, tv, torch.from_numpy(g).to(dev), [(seg0, torch.from_numpy(s0).to(dev)), (seg1, torch.from_numpy(s1).to(dev))],
                              shadow16=shadow, lr=1e-3, beta1_power=0.9, beta2_power=0.999, grad_scale=1 / 1024)
        rp, rm, rv = p.copy(), m.copy(), v.copy()
        oracle.dense_adam(rp, rm, rv, gsum, lr=1e-3, b1_pow=0.9, b2_pow=0.999, grad_scale=1 / 1024)
        assert np.array_equal(tp.cpu().numpy(), rp) and np.array_equal(tm.cpu().numpy(), rm) and np.array_equal(tv.cpu().numpy(), rv)
        assert torch.equal(shadow.cpu(), torch.from_numpy(rp).to(sh_dt))


def test_dense_rejects_unsupported_shapes(dev):
    from mindrec_amd import _lib, ops
    x = torch.zeros((64, 20), dtype=torch.bfloat16, device=dev)       # K = 20: rows are not 16-byte multiples
    w = torch.zeros((20, 16), dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.MrecError):
        ops.dense_fwd(x, w, None)
    with pytest.raises(RuntimeError):
        ops.dense_fwd(x.cpu(), w.cpu(), None)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N,mask", [(1024, 512, 256, True), (300, 3120, 136, False), (2048, 256, 128, True)])
def test_dense_bwd_fused_launch_equals_separate_kernels(dev, oracle, dtype, M, K, N, mask):
    """Both bprops of a layer in one launch (what the engine runs) == the two separate kernels, bit for bit: the same
    workgroup bodies, only dispatched together."""
    from mindrec_amd import ops
    rng = np.random.default_rng(M + K)
    dy = _dev(_vals(rng, (M, N), 1e-2, dtype, oracle), dtype, dev)
    w = _dev(_vals(rng, (K, N), 0.05, dtype, oracle), dtype, dev)
    x = _dev(np.maximum(_vals(rng, (M, K), 1.0, dtype, oracle), 0), dtype, dev)
    S = ops.dense_bwd_weight_slabs(M, K, N)
    dw1 = torch.empty((S, K, N), dtype=torch.float32, device=dev); dw2 = torch.empty_like(dw1)
    db1 = torch.empty((ops.dense_bwd_bias_slabs(M, K, N, True), K), dtype=torch.float32, device=dev)
    db2 = torch.empty((ops.dense_bwd_bias_slabs(M, K, N, False), K), dtype=torch.float32, device=dev)
    dx1 = ops.dense_bwd(dy, w, x, dw1, mask=mask, db_slabs=db1 if mask else None)
    dx2 = ops.dense_bwd_input(dy, w, h=x if mask else None, db_slabs=db2 if mask else None)
    ops.dense_bwd_weight(x, dy, dw2)
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2)
    if mask:
        assert torch.equal(db1, db2)
        ref, ref_db = oracle.dense_bwd_input(dy.float().cpu().numpy(), w.float().cpu().numpy(), x.float().cpu().numpy(), dtype)
        tot = torch.empty(K, dtype=torch.float32, device=dev)
        ops.sum_slabs(db1, tot)
        assert np.allclose(tot.cpu().numpy(), ref_db, rtol=1e-2, atol=1e-3 * np.abs(ref_db).max())
