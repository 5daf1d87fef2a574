"""Dropout on the DenseLayer inputs (models/wide_deep/src/wide_and_deep.py:98,117-118; switched on by dropout_flag,
benchmarks/wide_deep/default_config.yaml:15) -- the counter-based mask of csrc/mrec_dropout.h against the oracle's own
statement of it (oracle.dropout_mask), as a pass of its own, inside the MFMA GEMM epilogues, through the output head, and in
the engine: fp32 net against the oracle-driven engine, 16-bit net against the mixed-precision oracle, graph replay == eager."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

T16 = {"bf16": torch.bfloat16, "f16": torch.float16}


@pytest.mark.parametrize("M,W,keep,seed,step,layer,row0", [
    (1000, 2080, 0.5, 1004, 0, 0, 0), (777, 1024, 0.5, 1004, 3, 1, 16384), (64, 128, 0.8, 5, 100000, 4, 7), (3, 4, 0.25, 2 ** 63 + 11, 1, 15, 0),
    (513, 520, 0.999, 9, 2, 2, 0), (100, 256, 1.0, 9, 2, 2, 0), (0, 256, 0.5, 9, 2, 2, 0)])
def test_mask_matches_oracle(dev, oracle, M, W, keep, seed, step, layer, row0):
    from mindrec_amd import ops
    d = ops.Dropout(keep, seed, layer, step=step, row0=row0)
    got = ops.dropout_mask(M, W, d, dev).cpu().numpy()
    ref = oracle.dropout_mask(M, W, seed, step, layer, keep, row0)
    assert np.array_equal(got, ref)
    if M * W >= 10000 and keep < 1:
        n = M * W
        assert abs((ref > 0).mean() - keep) <= 5 * np.sqrt(keep * (1 - keep) / n) + 2.0 ** -16      # Bernoulli(keep) (+ threshold grain)
    # the step read from a device-side step state draws the same mask
    ss = ops.StepState(dev, step=step)
    d2 = ops.Dropout(keep, seed, layer, step=12345, row0=row0, step_state=ss)
    assert np.array_equal(ops.dropout_mask(M, W, d2, dev).cpu().numpy(), ref)


def test_mask_matches_golden(dev):
    import json
    from mindrec_amd import ops
    for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dropout_masks.json"))):
        mk = ops.dropout_mask(c["M"], c["W"], ops.Dropout(c["keep"], c["seed"], c["layer"], step=c["step"], row0=c["row0"]), dev).cpu().numpy()
        assert "".join("1" if x > 0 else "0" for x in mk.ravel()) == c["kept"]


def test_mask_rows_do_not_depend_on_the_split(dev, oracle):
    """N data-parallel ranks draw the mask of the one big batch: rows [row0, row0 + m) of the [M, W] mask."""
    from mindrec_amd import ops
    whole = oracle.dropout_mask(4096, 512, 21, 7, 1, 0.5)
    for row0, m in ((0, 1024), (1024, 1024), (3000, 1096)):
        part = ops.dropout_mask(m, 512, ops.Dropout(0.5, 21, 1, step=7, row0=row0), dev).cpu().numpy()
        assert np.array_equal(part, whole[row0: row0 + m])
    assert not np.array_equal(oracle.dropout_mask(64, 512, 21, 8, 1, 0.5), whole[:64])        # another step: another mask
    assert not np.array_equal(oracle.dropout_mask(64, 512, 21, 7, 2, 0.5), whole[:64])        # another layer: another mask


@pytest.mark.parametrize("kind", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("keep", [0.5, 0.8])
def test_dropout_pass_matches_oracle(dev, oracle, kind, keep):
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    M, W = 1000, 2080
    x = rng.standard_normal((M, W)).astype(np.float32)
    if kind != "f32":
        x = oracle.round16(x, kind)
    t = torch.from_numpy(x).to(dev)
    if kind != "f32":
        t = t.to(T16[kind])
    d = ops.Dropout(keep, 77, 0, step=5)
    mask = oracle.dropout_mask(M, W, 77, 5, 0, keep)
    ref = oracle.dropout(x, mask, None if kind == "f32" else kind)
    out = torch.empty_like(t)
    ops.dropout_(t, d, out=out)
    assert np.array_equal(out.float().cpu().numpy(), ref)
    # strided rows, in place
    big = torch.zeros((M, W + 8), dtype=t.dtype, device=dev)
    big[:, :W] = t
    ops.dropout_(big[:, :W], d)
    assert np.array_equal(big[:, :W].float().cpu().numpy(), ref) and float(big[:, W:].abs().sum()) == 0.0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("keep", [0.5, 0.8])
@pytest.mark.parametrize("M,K,N", [(1000, 2080, 1024), (300, 264, 136), (16384, 256, 128)])
def test_dense_forward_epilogue(dev, oracle, dt, keep, M, K, N):
    """y = Dropout(round16(relu(x . w + b))): bit-exactly the oracle's dropout of the kernel's own no-dropout output (whose
    parity with the oracle's DenseLayer is tests/test_dense_gpu.py's subject) -- both tile configurations, ragged edges."""
    from mindrec_amd import ops
    rng = np.random.default_rng(M + N)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev).to(T16[dt])
    w = torch.from_numpy((rng.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev).to(T16[dt])
    b = torch.from_numpy((rng.standard_normal(N) * 0.1).astype(np.float32)).to(dev)
    y0 = ops.dense_fwd(x, w, b, relu=True).float().cpu().numpy()
    y1 = ops.dense_fwd(x, w, b, relu=True, drop_next=ops.Dropout(keep, 31, 2, step=9, row0=5)).float().cpu().numpy()
    mask = oracle.dropout_mask(M, N, 31, 9, 2, keep, row0=5)
    assert np.array_equal(y1, oracle.dropout(y0, mask, dt))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(1000, 2080, 1024), (300, 200, 72), (4096, 512, 256)])
def test_dense_backward_epilogues(dev, oracle, dt, M, K, N):
    """Input gradient in front of a Dropout: keep 0.5 scales by 2 -- exact in any binary format, so the result must be
    bit for bit twice the no-dropout kernel's, masked; keep 0.8 against the oracle's restatement within one 16-bit ulp.
    Both the separate and the fused (input + weight gradient) launch; with the activation as mask source (hidden layers:
    its zeros carry the mask) and with the hash mask (first layer)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(K + N)
    ulp = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}[dt]
    dy = torch.from_numpy((rng.standard_normal((M, N)) * 0.1).astype(np.float32)).to(dev).to(T16[dt])
    w = torch.from_numpy((rng.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev).to(T16[dt])
    for keep in (0.5, 0.8):
        mask = oracle.dropout_mask(M, K, 11, 4, 1, keep)
        h_act = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
        h_np = oracle.dropout(oracle.round16(h_act, dt), mask, dt)                       # the dropped-out activation the layer saw
        h = torch.from_numpy(h_np).to(dev).to(T16[dt])
        d = ops.Dropout(keep, 11, 1, step=4)
        S = ops.dense_bwd_weight_slabs(M, K, N)
        for use_h in (True, False):
            base = ops.dense_bwd_input(dy, w, h=h if use_h else None).float().cpu().numpy()
            got = ops.dense_bwd_input(dy, w, h=h if use_h else None, drop_in=d).float().cpu().numpy()
            slabs = torch.empty((S, K, N), dtype=torch.float32, device=dev)
            fused = ops.dense_bwd(dy, w, h, slabs, mask=use_h, drop_in=d).float().cpu().numpy()
            assert np.array_equal(got, fused)
            if use_h:
                assert np.array_equal(got == 0, (h_np <= 0) | (got == 0)) and not got[h_np <= 0].any()
            else:
                assert not got[mask == 0].any()
            if keep == 0.5:
                ref = np.where(mask > 0, 2.0 * base, 0.0) if not use_h else 2.0 * base
                big = np.abs(base) >= 2.0 ** -13                                          # (in the f16 subnormal range rounding does not commute with doubling)
                assert np.array_equal(got[big], ref[big].astype(np.float32))
            ref, _ = oracle.dense_bwd_input(dy.float().cpu().numpy(), w.float().cpu().numpy(), h_np if use_h else None, dt,
                                            scale=float(np.float32(1) / np.float32(keep)), mask=None if use_h else mask)
            tol = ulp * np.abs(ref) + 1e-3 * ulp * np.abs(ref).max() + (2.0 ** -24 if dt == "f16" else 0.0)
            assert (np.abs(got - ref) <= tol).all() and np.mean(got == ref) >= 0.97


def test_head_scales_the_gradient(dev, oracle):
    from mindrec_amd import ops
    rng = np.random.default_rng(5)
    B, K5 = 4096, 128
    mask = oracle.dropout_mask(B, K5, 3, 2, 4, 0.5)
    h4_np = oracle.dropout(oracle.round16(np.maximum(rng.standard_normal((B, K5)), 0).astype(np.float32), "bf16"), mask, "bf16")
    h4 = torch.from_numpy(h4_np).to(dev).to(torch.bfloat16)
    w5 = torch.from_numpy((rng.standard_normal(K5) * 0.1).astype(np.float32)).to(dev)
    b5 = torch.zeros(1, device=dev)
    wide = torch.from_numpy(rng.standard_normal(B).astype(np.float32) * 0.1).to(dev)
    label = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
    outs = []
    for sc in (1.0, 2.0):
        dw5 = torch.empty(K5, device=dev); db4 = torch.empty(K5, device=dev); db5 = torch.empty(1, device=dev)
        loss, logit, dlogit, dh4 = ops.head_fwd_bwd(h4, w5, b5, wide, label, 1024.0 / B, dw5, db4, db5, dh_scale=sc)
        outs.append((float(loss), dlogit.cpu().numpy(), dh4.float().cpu().numpy(), db4.cpu().numpy(), dw5.cpu().numpy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][4], outs[1][4])
    assert np.array_equal(outs[1][2], 2.0 * outs[0][2]) and not outs[1][2][h4_np <= 0].any()
    assert np.allclose(outs[1][3], 2.0 * outs[0][3], rtol=1e-6)
    ref = oracle.head_fwd_bwd(h4_np, w5.cpu().numpy(), 0.0, wide.cpu().numpy(), label.cpu().numpy(), 1024.0 / B, dh_scale=2.0)
    assert np.abs(outs[1][2] - ref["dh4"]).max() <= 2.0 ** -7 * np.abs(ref["dh4"]).max()


def _row_rel(a, b):
    den = np.maximum(np.abs(b).max(axis=1), 1e-30)
    return float((np.abs(a.astype(np.float64) - b).max(axis=1) / den).max())


def test_fp32_engine_with_dropout_matches_oracle_engine(dev, oracle):
    """Dropout on the fp32 net (the reference's published benchmark configuration, benchmarks/wide_deep/default_config.yaml:15-16):
    the hand-written fp32 MFMA net -- masks applied in place to the stored activations by mrec_dropout, 1 / keep folded into the
    backward -- against the oracle-side engine on the CPU, whose masks come from oracle.dropout_mask through autograd."""
    from _oracle_engine import OracleWideDeepEngine
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=50_000, emb_dim=80, field_size=26, batch_size=256, deep_layer_dim=[64, 32], mlp_dtype="fp32",
                         dropout_flag=True)
    g = WideDeepEngine(cfg, dev)
    assert g._f32net
    c = OracleWideDeepEngine(cfg, "cpu")
    n = WideDeepEngine(WideDeepConfig(**{**cfg.__dict__, "dropout_flag": False}), dev)
    for s in range(3):
        ids, wts, label = synthetic_batch(cfg, "cpu", "zipf", seed=7 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        ln = float(n.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 1e-5 * max(abs(lc), 1e-3)
        assert lg != ln                                                   # the masks do act
    a, b = g.deep.cpu().numpy(), c.deep.numpy()
    assert _row_rel(a, b) <= 2e-5
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=1e-4, atol=1e-6)
    # evaluation has no dropout: predictions of both engines agree
    ids, wts, _ = synthetic_batch(cfg, "cpu", "zipf", seed=99)
    pg = g.predict(ids.to(dev), wts.to(dev))[1].cpu().numpy()
    pc = c.predict(ids, wts)[1].numpy()
    assert np.allclose(pg, pc, rtol=1e-4, atol=1e-6)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_mixed_engine_with_dropout_vs_mixed_oracle(dev, oracle, dt):
    """The 16-bit MFMA net with Dropout (mask in the GEMM epilogues, step read from device memory inside the whole-step
    graph) free-running against the mixed-precision oracle: 6 steps, graphs replayed from step 4 on."""
    from _oracle_mixed import OracleMixedEngine
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=500_000, emb_dim=80, field_size=26, batch_size=4096, deep_layer_dim=[1024, 512, 256, 128],
                         mlp_dtype={"bf16": "bf16", "f16": "fp16"}[dt], dropout_flag=True)
    g = WideDeepEngine(cfg, dev)
    assert g._mfma and g._dropout
    o = OracleMixedEngine(cfg, dt)
    steps = 6
    lg, lo = [], []
    for s in range(steps):
        ids, wts, label = synthetic_batch(cfg, "cpu", "uniform" if s % 2 == 0 else "zipf", seed=3000 + s)
        if s == 0:
            # kernel level on the first step: the engine's dropped-out looked-up rows and first activation against the oracle's
            r = o.forward_backward(ids.numpy(), wts.numpy(), label.numpy().ravel())
            g._training = True                        # (what train_step sets; no device-side step state yet: the step goes by value)
            g.step_count += 1
            emb, wide, _ = g.lookup(ids.to(dev), wts.to(dev))
            hs = g._mlp_fwd(emb)
            g._training = False
            g.step_count -= 1
            assert np.array_equal(hs[0].float().cpu().numpy(), r["emb"]), "Dropout of the looked-up rows must be bit-exact"
            h1 = hs[1].float().cpu().numpy()
            assert np.array_equal(h1 == 0, r["hs"][1] == 0) or np.mean((h1 == 0) == (r["hs"][1] == 0)) >= 0.9999
            assert abs(np.mean(h1 == 0) - np.mean(r["hs"][1] == 0)) <= 1e-4
        lg.append(float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev))))
        lo.append(o.train_step(ids.numpy(), wts.numpy(), label.numpy().ravel()))
    assert g._step_graph is not None, "the whole-step graph must have been captured and replayed"
    print("  losses gpu   ", lg)
    print("  losses oracle", lo)
    assert np.allclose(lg, lo, rtol=2e-4)
    a, b = g.deep.cpu().numpy(), o.deep
    touched = (o.deep_m != 0).any(axis=1)
    assert np.array_equal((g.deep_m.cpu().numpy() != 0).any(axis=1), touched)
    assert np.array_equal(a[~touched], b[~touched])
    d = np.abs(a[touched].astype(np.float64) - b[touched])
    print(f"  deep rows: max |diff| = {d.max():.3e} (lr = {cfg.adam_lr}), fraction within 5 % of lr: {np.mean(d <= 0.05 * cfg.adam_lr):.5f}")
    assert d.max() <= 2 * cfg.adam_lr * steps and np.mean(d <= 0.05 * cfg.adam_lr) >= 0.999
    dd = np.abs(g.dense_flat.detach().cpu().numpy().astype(np.float64) - o.flat)
    assert dd.max() <= 2 * cfg.adam_lr * steps and np.mean(dd <= 0.05 * cfg.adam_lr) >= 0.99


def test_dropout_graph_replay_equals_eager(dev):
    """The captured whole-step graph reads the step from device memory: 7 steps with graphs must equal 7 eager steps bit for bit
    (a mask frozen at capture time would not)."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    base = dict(vocab_size=200_000, emb_dim=80, field_size=26, batch_size=2048, deep_layer_dim=[256, 128], mlp_dtype="bf16", dropout_flag=True)
    a = WideDeepEngine(WideDeepConfig(**base), dev)
    b = WideDeepEngine(WideDeepConfig(**base, graphs="none"), dev)
    cfg = a.cfg
    for s in range(7):
        ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=40 + s)
        la = float(a.train_step(ids, wts, label))
        lb = float(b.train_step(ids, wts, label))
        assert la == lb, (s, la, lb)
    assert a._step_graph is not None and b._step_graph is None
    assert torch.equal(a.deep, b.deep) and torch.equal(a.wide, b.wide) and torch.equal(a.dense_flat, b.dense_flat)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("keep", [0.5, 0.8])
def test_lookup_with_dropout_equals_lookup_then_dropout(dev, oracle, dt, keep):
    """mrec_gather_rows_wide(drop=...): the first layer's Dropout applied to the looked-up rows on their way out of the lookup must
    equal the lookup followed by the Dropout pass (which is checked against the oracle above), wide products untouched."""
    from mindrec_amd import ops
    B, F, D, V = 1000, 26, 80, 5000
    rows = torch.randn(V, 3 * D + 4, device=dev)
    table = rows[:, :D + 4]                          # fused rows: the wide word right behind the deep columns
    ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
    wts = torch.rand(B, F, device=dev)
    d = ops.Dropout(keep, 1004, 0, step=6, row0=4096)
    e0, w0 = ops.gather_rows_wide(table[:, :D], ids, wts, D, out_dtype=T16[dt])
    e1, w1 = ops.gather_rows_wide(table[:, :D], ids, wts, D, out_dtype=T16[dt], drop=d)
    ref = ops.dropout_(e0.reshape(B, F * D).clone(), d)
    assert torch.equal(e1.reshape(B, F * D), ref) and torch.equal(w0, w1)
    mask = oracle.dropout_mask(B, F * D, 1004, 6, 0, keep, row0=4096)
    assert np.array_equal(e1.reshape(B, F * D).float().cpu().numpy(), oracle.dropout(e0.reshape(B, F * D).float().cpu().numpy(), mask, dt))
