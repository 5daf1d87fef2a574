"""fp32 DenseLayers at the 16-bit matrix rate (csrc/mrec_gemm_x3.hip): every fp32 operand as three bf16 parts, six bf16 products per
fp32 product accumulated in fp32.  Against float64 on the same inputs: the error must be that of an fp32 GEMM -- a few ulps of
sum |a| |b| -- for the forward product, the input gradient and the weight gradient, at ragged shapes (Deep&Cross's K = 1170), with
operands spanning many binades (gradients at the loss scale down to 1e-12)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(got, ref64, absum):
    return float((np.abs(got.astype(np.float64) - ref64) / absum).max())


@pytest.mark.parametrize("M,K,N", [(1024, 1170, 1024), (512, 1024, 256), (768, 200, 136), (256, 64, 64)])
def test_x3_products_have_fp32_accuracy(dev, M, K, N):
    from mindrec_amd import ops
    rng = np.random.default_rng(M + K)
    x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-12, 2, (M, 1)))).astype(np.float32)       # rows over 6 decades
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    dy = (rng.standard_normal((M, N)) * np.exp(rng.uniform(-20, 0, (M, 1)))).astype(np.float32)
    tx, tw, tdy = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
    xp, wp, dyp = ops.x3_split(tx), ops.x3_split(tw), ops.x3_split(tdy)
    # the parts add up to the operand exactly (three 8-bit pieces of a 24-bit mantissa)
    back = xp[:, :M, :K].to(torch.float32).sum(0)
    assert torch.equal(back, tx)
    x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    u = 2.0 ** -24
    # forward
    y = ops.x3_gemm(0, xp, wp, M, K, N, torch.empty((M, N), dtype=torch.float32, device=dev)).cpu().numpy()
    assert _err(y, x64 @ w64, np.abs(x64) @ np.abs(w64)) <= 6 * u
    # the exact-fp32 kernel on the same inputs, for scale: the split path is no less accurate than twice its error + 2 ulp
    y32 = ops.dense32_fwd(tx, tw, None, relu=False).cpu().numpy()
    assert _err(y, x64 @ w64, np.abs(x64) @ np.abs(w64)) <= 2 * _err(y32, x64 @ w64, np.abs(x64) @ np.abs(w64)) + 2 * u
    # input gradient: dx = dy . w^T  (ld of the output not a multiple of 4 when K = 1170)
    dx = ops.x3_gemm(1, dyp, wp, M, K, N, torch.empty((M, K), dtype=torch.float32, device=dev)).cpu().numpy()
    assert _err(dx, dy64 @ w64.T, np.abs(dy64) @ np.abs(w64.T)) <= 6 * u
    # weight gradient in S batch slabs: a reduction over the batch, whose rows span many binades -- held to the exact-fp32 kernel's
    # own error on the same data (an fp32 accumulation of M terms is not better than a few ulps of sum |a| |b| either)
    S32 = ops.dense32_bwd_weight_slabs(M, K, N)
    w32 = ops.dense32_bwd_weight(tx, tdy, torch.empty((S32, K, N), dtype=torch.float32, device=dev)).cpu().numpy().astype(np.float64).sum(0)
    e32 = _err(w32, x64.T @ dy64, np.abs(x64.T) @ np.abs(dy64))
    for S in (1, 3):
        slabs = ops.x3_gemm(2, xp, dyp, M, K, N, torch.empty((S, K, N), dtype=torch.float32, device=dev), S=S).cpu().numpy()
        e = _err(slabs.astype(np.float64).sum(0), x64.T @ dy64, np.abs(x64.T) @ np.abs(dy64))
        assert e <= max(8 * u, 2 * e32 + 2 * u), (e / u, e32 / u)


def test_x3_post_ops(dev):
    """bias + ReLU in place with the result's own parts image; ReLU mask + column sums per 64 rows with the parts image."""
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    M, N = 300, 136
    acc = rng.standard_normal((M, N)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    t = torch.from_numpy(acc.copy()).to(dev)
    parts = ops.x3_parts(M, N, dev)
    ops.x3_bias_relu_(t, torch.from_numpy(b).to(dev), relu=True, parts_out=parts)
    ref = np.maximum(acc + b, 0).astype(np.float32)
    assert np.array_equal(t.cpu().numpy(), ref)
    assert torch.equal(parts[:, :M, :N].to(torch.float32).sum(0).cpu(), torch.from_numpy(ref))
    assert float(parts[:, M:, :].abs().max()) == 0 and float(parts[:, :, N:].abs().max()) == 0      # padding is zero
    h = rng.standard_normal((M, N)).astype(np.float32)
    t2 = torch.from_numpy(acc.copy()).to(dev)
    cs = torch.empty(((M + 63) // 64, N), dtype=torch.float32, device=dev)
    ops.x3_mask_colsum_(t2, torch.from_numpy(h).to(dev), cs, parts)
    ref2 = np.where(h > 0, acc, 0).astype(np.float32)
    assert np.array_equal(t2.cpu().numpy(), ref2)
    for tile in range(cs.shape[0]):
        assert np.allclose(cs[tile].cpu().numpy(), ref2[tile * 64:(tile + 1) * 64].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    assert torch.equal(parts[:, :M, :N].to(torch.float32).sum(0).cpu(), torch.from_numpy(ref2))


@pytest.mark.parametrize("M,K,N", [(1024, 1170, 1024), (300, 1024, 136), (16384, 200, 72)])
@pytest.mark.parametrize("scale", [1.0, 1.25])
def test_x3_fused_output_ends_equal_the_two_pass_form(dev, M, K, N, scale):
    """x3_fwd / x3_dgrad (bias + ReLU, or ReLU mask + 1 / keep + bias gradient, and the output's own parts inside the GEMM's epilogue)
    against the GEMM followed by the post pass: outputs and parts to the bit, 64-row column sums to summation order."""
    from mindrec_amd import ops
    rng = np.random.default_rng(M + N)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    x, w, b, dy, h = f(M, K), f(K, N) * 0.05, f(N), f(M, N), f(M, K)
    xp, wp, dyp = ops.x3_split(x), ops.x3_split(w), ops.x3_split(dy)
    # forward
    y1, p1 = torch.empty((M, N), device=dev), ops.x3_parts(M, N, dev)
    ops.x3_bias_relu_(ops.x3_gemm(0, xp, wp, M, K, N, y1), b, True, parts_out=p1)
    y2, p2 = torch.empty((M, N), device=dev), ops.x3_parts(M, N, dev)
    ops.x3_fwd(xp, wp, M, K, N, y2, bias=b, relu=True, parts_out=p2)
    assert torch.equal(y1, y2) and torch.equal(p1.view(torch.int16), p2.view(torch.int16))
    # ... with the Dropout on the next layer's input in the epilogue too: == bias + ReLU, then ops.dropout_, then the split
    if N % 4 == 0:
        drop = ops.Dropout(0.5 if scale == 1.0 else 0.8, seed=77, layer=2, step=5, row0=1000)
        y4, p4 = torch.empty((M, N), device=dev), ops.x3_parts(M, N, dev)
        ops.x3_fwd(xp, wp, M, K, N, y4, bias=b, relu=True, parts_out=p4, drop_next=drop)
        y5 = ops.dropout_(y1.clone(), drop)
        assert torch.equal(y4, y5) and torch.equal(p4.view(torch.int16), ops.x3_split(y5).view(torch.int16))
        assert 0.3 < float((y4 == 0).float().mean()) < 0.95
    y3 = ops.x3_fwd(xp, wp, M, K, N, torch.empty((M, N), device=dev), bias=None, relu=False)
    assert torch.equal(y3, ops.x3_gemm(0, xp, wp, M, K, N, torch.empty((M, N), device=dev)))
    # input gradient (the output's row stride is not a multiple of 4 when K = 1170)
    T = (M + 63) // 64
    d1, c1, q1 = torch.empty((M, K), device=dev), torch.empty((T, K), device=dev), ops.x3_parts(M, K, dev)
    ops.x3_mask_colsum_(ops.x3_gemm(1, dyp, wp, M, K, N, d1), h=h, colsum=c1, parts_out=q1, scale=scale)
    d2, c2, q2 = torch.empty((M, K), device=dev), torch.full((T, K), float("nan"), device=dev), ops.x3_parts(M, K, dev)
    ops.x3_dgrad(dyp, wp, M, K, N, d2, h=h, scale=scale, colsum=c2, parts_out=q2)
    assert torch.equal(d1, d2) and torch.equal(q1.view(torch.int16), q2.view(torch.int16))
    ref = torch.stack([d2[t * 64:(t + 1) * 64].double().sum(0) for t in range(T)])
    mag = torch.stack([d2[t * 64:(t + 1) * 64].double().abs().sum(0) for t in range(T)])
    assert float(((c2.double() - ref).abs() / (mag + 1e-30)).max()) <= 64 * 2.0 ** -24
    assert float(((c1.double() - ref).abs() / (mag + 1e-30)).max()) <= 64 * 2.0 ** -24
    # no mask, no sums, no parts: the plain input gradient
    d3 = ops.x3_dgrad(dyp, wp, M, K, N, torch.empty((M, K), device=dev))
    assert torch.equal(d3, ops.x3_gemm(1, dyp, wp, M, K, N, torch.empty((M, K), device=dev)))


def test_x3_plain_dgrad_with_a_narrow_last_tile(dev):
    """Deep&Cross's input gradient into 1170 columns at the benchmark batch: 4 full column tiles in one round over the chip and the
    146-column rest as slabs of the reduction (through the workspace) -- every column still an fp32-class product."""
    from mindrec_amd import _lib, ops
    M, K, N = 16384, 1170, 256
    assert _lib.query_bytes("mrec_x3_gemm_dgrad_workspace_bytes", M, K, N) > 0          # this shape takes the split form
    assert _lib.query_bytes("mrec_x3_gemm_dgrad_workspace_bytes", 1024, K, N) == 0
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    dy = (rng.standard_normal((M, N)) * np.exp(rng.uniform(-20, 0, (M, 1)))).astype(np.float32)
    tw, tdy = torch.from_numpy(w).to(dev), torch.from_numpy(dy).to(dev)
    wp, dyp = ops.x3_split(tw), ops.x3_split(tdy)
    dx = torch.full((M, K), float("nan"), device=dev)
    ops.x3_dgrad(dyp, wp, M, K, N, dx)
    one = ops.x3_gemm(1, dyp, wp, M, K, N, torch.empty((M, K), device=dev))            # the one-launch form
    assert torch.equal(dx[:, :1024], one[:, :1024])
    w64, dy64 = w.astype(np.float64), dy.astype(np.float64)
    assert _err(dx.cpu().numpy(), dy64 @ w64.T, np.abs(dy64) @ np.abs(w64.T)) <= 6 * 2.0 ** -24
