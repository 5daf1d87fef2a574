"""The tail of the dense net as ONE launch (mrec_tail_fwd_bwd: two DenseLayers forward, the output head, two input-gradient
bprops; csrc/mrec_tail.hip k_tail) against the five separate launches it replaces -- which are the ones checked against the
oracle (tests/test_dense_gpu.py, test_gpu_parity.py, test_bench_shape_gpu.py).  Same products in the same order: the 16-bit
tensors, the logits and the per-sample gradients must be IDENTICAL; the batch reductions (dw5, bias gradients, loss) are taken
over different partial groupings and agree to fp32 summation accuracy."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

T16 = {"bf16": torch.bfloat16, "f16": torch.float16}
K2, N2, N3 = 512, 256, 128


def _inputs(dev, dt, B, seed, F=0):
    rng = np.random.default_rng(seed)
    t16 = T16[dt]
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, K2)), 0).astype(np.float32)).to(dev).to(t16)     # a ReLU output
    w2 = torch.from_numpy((rng.standard_normal((K2, N2)) * 0.06).astype(np.float32)).to(dev).to(t16)
    w3 = torch.from_numpy((rng.standard_normal((N2, N3)) * 0.08).astype(np.float32)).to(dev).to(t16)
    b2 = torch.from_numpy((rng.standard_normal(N2) * 0.1).astype(np.float32)).to(dev)
    b3 = torch.from_numpy((rng.standard_normal(N3) * 0.1).astype(np.float32)).to(dev)
    w5 = torch.from_numpy((rng.standard_normal(N3) * 0.1).astype(np.float32)).to(dev)
    b5 = torch.from_numpy(rng.standard_normal(1).astype(np.float32) * 0.1).to(dev)
    label = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
    if F:
        wide = torch.zeros((B, F, 2), dtype=torch.float32, device=dev)
        wide[:, :, 0] = torch.from_numpy((rng.standard_normal((B, F)) * 0.05).astype(np.float32)).to(dev)
        wb = torch.from_numpy(rng.standard_normal(1).astype(np.float32) * 0.1).to(dev)
    else:
        wide = torch.from_numpy((rng.standard_normal(B) * 0.2).astype(np.float32)).to(dev)
        wb = None
    return x, w2, b2, w3, b3, w5, b5, wide, wb, label


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,F,keep", [(16384, 26, None), (16384, 0, None), (64, 0, None), (4096, 26, 0.5), (1024, 39, 0.8)])
def test_tail_equals_the_separate_launches(dev, dt, B, F, keep):
    from mindrec_amd import ops
    x, w2, b2, w3, b3, w5, b5, wide, wb, label = _inputs(dev, dt, B, seed=B + F, F=F)
    assert ops.tail_supported(B, K2, N2, N3)
    dscale = 1024.0 / B
    d = (lambda layer: ops.Dropout(keep, 1004, layer, step=3, row0=128)) if keep else (lambda layer: None)
    if keep:                               # x is then the dropped-out input of the first tail layer (layer index 2)
        x = ops.dropout_(x.clone(), d(2))
    dhs = d(2).scale if keep else 1.0
    # --- the five launches
    y2 = ops.dense_fwd(x, w2, b2, relu=True, drop_next=d(3))
    y3 = ops.dense_fwd(y2, w3, b3, relu=True, drop_next=d(4))
    dw5 = torch.empty(N3, device=dev); db4 = torch.empty(N3, device=dev); db5 = torch.empty(1, device=dev)
    if F:
        loss, logit, dlogit, dz4 = ops.head_fwd_bwd_wide(y3, w5, b5, wide, wb, label, dscale, dw5, db4, db5, dh_scale=dhs)
    else:
        loss, logit, dlogit, dz4 = ops.head_fwd_bwd(y3, w5, b5, wide, label, dscale, dw5, db4, db5, dh_scale=dhs)
    db3 = torch.empty(N2, device=dev); db2 = torch.empty(K2, device=dev)
    dz3 = ops.dense_bwd_input(dz4, w3, h=y2, db_out=db3, drop_in=d(3))
    dz2 = ops.dense_bwd_input(dz3, w2, h=x, db_out=db2, drop_in=d(2))
    # --- the one launch
    packed = ops.tail_pack_weights(w2, w3)
    tdw5 = torch.empty(N3, device=dev); tdb4 = torch.empty(N3, device=dev); tdb5 = torch.empty(1, device=dev)
    s3 = torch.empty(N2, device=dev); s2 = torch.empty(K2, device=dev)
    out = {}
    tloss, tdlogit, ty2, tdz4, tdz3, tdz2 = ops.tail_fwd_bwd(x, packed, b2, b3, w5, b5, wide, wb, label, dscale, tdw5, tdb4, tdb5,
                                                              s3, s2, drop_in=d(2), out=out)
    torch.cuda.synchronize()
    assert torch.equal(ty2, y2), "second tail layer's input"
    assert torch.equal(out["logit"], logit) and torch.equal(tdlogit, dlogit)
    assert torch.equal(tdz4, dz4) and torch.equal(tdz3, dz3) and torch.equal(tdz2, dz2)
    assert abs(float(tloss) - float(loss)) <= 2e-6 * abs(float(loss))

    def close(a, b, what):
        a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-9, what
    close(tdw5, dw5, "dw5"); close(tdb4, db4, "db4"); close(tdb5, db5, "db5")
    close(s3, db3, "bias gradient of the first tail layer"); close(s2, db2, "bias gradient of the layer below")
    if keep:
        assert float((ty2 == 0).float().mean()) > 0.3          # the masks act


def test_tail_rejects_other_shapes(dev):
    from mindrec_amd import ops
    assert not ops.tail_supported(100, K2, N2, N3) and not ops.tail_supported(64, 256, N2, N3) and not ops.tail_supported(1 << 20, K2, N2, N3)
    assert ops.tail_supported(32768, K2, N2, N3)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_extra_weight_gradients_ride_the_backward_launch(dev, dt):
    """ops.dense_bwd(..., extra=[...]): the weight gradients of two other layers computed by the workgroups behind a layer's own
    backward launch must equal the separate mrec_dense_bwd_weight_* launches slab by slab, and leave the launch's own results alone."""
    from mindrec_amd import ops
    torch.manual_seed(0)
    M = 16384
    t16 = T16[dt]

    def r(*s):
        return (torch.randn(*s, device=dev) * 0.1).to(t16)

    def slabs(K, N):
        return torch.empty((ops.dense_bwd_weight_slabs(M, K, N), K, N), device=dev)
    dy, w, x = r(M, 512), r(1024, 512), torch.relu(r(M, 1024))
    x2, dy2, x3, dy3 = torch.relu(r(M, 512)), r(M, 256), torch.relu(r(M, 256)), r(M, 128)
    s1, s2, s3, r2, r3 = slabs(1024, 512), slabs(512, 256), slabs(256, 128), slabs(512, 256), slabs(256, 128)
    db = torch.empty((ops.dense_bwd_bias_slabs(M, 1024, 512), 1024), device=dev)
    dx_a = ops.dense_bwd(dy, w, x, s1, mask=True, db_slabs=db).clone()
    s1a, dba = s1.clone(), db.clone()
    ops.dense_bwd_weight(x2, dy2, r2)
    ops.dense_bwd_weight(x3, dy3, r3)
    s1.zero_(); db.zero_()
    dx_b = ops.dense_bwd(dy, w, x, s1, mask=True, db_slabs=db, extra=[(x2, dy2, s2), (x3, dy3, s3)])
    assert torch.equal(dx_a, dx_b) and torch.equal(s1a, s1) and torch.equal(dba, db)
    assert torch.equal(r2, s2) and torch.equal(r3, s3)
    with pytest.raises(ValueError):
        ops.dense_bwd(dy, w, x, s1, extra=[(x2, dy2, s2)] * 3)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(16384, 2080, 1024), (16384, 1024, 512), (300, 264, 136), (1000, 3120, 264)])
def test_forward_with_transposed_weight_is_identical(dev, dt, M, K, N):
    """mrec_dense_fwd_wt_*: the weight given as [N, K] (written by mrec_dense_operand_copies) -- same products in the same order as
    the kernel that reads W [K, N] as stored, so the outputs must be bit-identical (ragged edges, K tails, both tile shapes, Dropout)."""
    from mindrec_amd import ops
    t16 = T16[dt]
    rng = np.random.default_rng(K + N)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev).to(t16)
    w = torch.from_numpy((rng.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev).to(t16)
    b = torch.from_numpy((rng.standard_normal(N) * 0.1).astype(np.float32)).to(dev)
    wt = torch.empty((N, K), dtype=t16, device=dev)
    other = torch.from_numpy(rng.standard_normal((64, 40)).astype(np.float32)).to(dev).to(t16)
    other_t = torch.empty((40, 64), dtype=t16, device=dev)
    ops.operand_copies([(w, wt), (other, other_t)])
    assert torch.equal(wt, w.t().contiguous()) and torch.equal(other_t, other.t().contiguous())
    for drop in (None, ops.Dropout(0.5, 9, 1, step=2)):
        y0 = ops.dense_fwd(x, w, b, relu=True, drop_next=drop)
        y1 = ops.dense_fwd(x, None, b, relu=True, drop_next=drop, wt=wt)
        assert torch.equal(y0, y1)


def test_operand_copies_with_the_tail_pack(dev):
    from mindrec_amd import ops
    _, w2, _, w3, *_ = _inputs(dev, "bf16", 64, 3)
    ref = ops.tail_pack_weights(w2, w3)
    a = torch.randn(2080, 1024, device=dev).to(torch.bfloat16)
    at = torch.empty((1024, 2080), dtype=torch.bfloat16, device=dev)
    packed = torch.empty_like(ref)
    ops.operand_copies([(a, at)], tail=(w2, w3, packed))
    assert torch.equal(packed, ref) and torch.equal(at, a.t().contiguous())


@pytest.mark.parametrize("B", [192, 160])
def test_engine_with_and_without_the_fused_tail(dev, B):
    """The engine with the tail launch (B = 192) against the engine that runs the same net layer by layer: identical 16-bit
    tensors, so everything downstream agrees to the accuracy of the few fp32 batch sums that are grouped differently (loss,
    dW5, bias gradients).  B = 160 is not a multiple of 64: both engines then run layer by layer and must agree bit for bit."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=100_000, emb_dim=16, field_size=26, batch_size=B, deep_layer_dim=[1024, 512, 256, 128], mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(fused_tail=True, **kw), dev)
    b = WideDeepEngine(WideDeepConfig(fused_tail=False, **kw), dev)
    assert a._tail_ok and not b._tail_ok and a._tail_now(B) == (B % 64 == 0)
    for s in range(6):
        ids, wts, label = synthetic_batch(a.cfg, dev, "zipf", seed=900 + s)
        la, lb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        assert abs(la - lb) <= (2e-6 * abs(lb) if B % 64 == 0 else 0.0), (s, la, lb)
    if B % 64:
        assert torch.equal(a.deep, b.deep) and torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
    else:
        assert float((a.deep - b.deep).abs().max()) <= 2 * a.cfg.adam_lr * 6
        frac = float(((a.dense_flat.detach() - b.dense_flat.detach()).abs() <= 0.05 * a.cfg.adam_lr).float().mean())
        assert frac >= 0.999, frac
        assert float(((a.deep - b.deep).abs() <= 0.05 * a.cfg.adam_lr).float().mean()) >= 0.9999
