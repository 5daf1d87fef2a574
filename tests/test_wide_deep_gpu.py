"""The whole Wide&Deep step on the GPU (HIP kernels + torch GEMMs, fp32 MLP) against the same
engine driven by the oracle on the CPU, on identical seeded batches."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def row_rel(a, b):
    den = np.maximum(np.abs(b).max(axis=1), 1e-30)
    return float((np.abs(a.astype(np.float64) - b).max(axis=1) / den).max())


@pytest.mark.parametrize("fields,dist_kind", [(39, "zipf"), (26, "uniform")])
def test_engine_matches_oracle_engine(dev, oracle, fields, dist_kind):
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=50_000, emb_dim=80, field_size=fields, batch_size=256, deep_layer_dim=[64, 32],
                         mlp_dtype="fp32")
    g = WideDeepEngine(cfg, dev)
    c = OracleWideDeepEngine(cfg, "cpu")
    assert np.array_equal(g.deep.cpu().numpy(), c.deep.numpy())                 # same init, bit for bit
    assert np.array_equal(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy())
    for s in range(3):
        ids, wts, label = synthetic_batch(cfg, "cpu", dist_kind, seed=7 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 1e-5 * max(abs(lc), 1e-3)
    # embedding rows: untouched rows identical; touched rows within 1e-5 row-relative plus what the
    # GEMM's fp32 summation order (hipBLASLt vs CPU BLAS) feeds into the gradients
    a, b = g.deep.cpu().numpy(), c.deep.numpy()
    assert row_rel(a, b) <= 2e-5, row_rel(a, b)
    # FTRL's weight is a ratio of cancelling terms, so the 1e-6 GEMM-order noise in d(logit) is
    # amplified: bound the wide table against its own scale instead of element by element
    a, b = g.wide.cpu().numpy(), c.wide.numpy()
    assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()
    untouched = (c.deep_m.numpy() == 0).all(axis=1)
    assert untouched.sum() > 1000
    assert np.array_equal(g.deep.cpu().numpy()[untouched], c.deep.numpy()[untouched])
    assert np.array_equal((g.deep_m.cpu().numpy() == 0).all(axis=1), untouched)
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=1e-4, atol=1e-6)


def test_dense_gradient_mode_matches_oracle_engine(dev, oracle):
    """sparse=False, the reference's default and its CPU-runnable configuration (BASELINE configs[0]: vocab 2 M, dim 16,
    batch 1024, 39 fields): dense [V, D] embedding gradients with the L2 term, nn.Adam / nn.FTRL over every row.  The engine on
    the GPU against the same engine driven by the oracle on the CPU: every row moves every step (Adam on l2_coef * E), the
    untouched wide weights collapse onto FTRL's fixed point, and both sides must agree on all of it."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=2_000_000, emb_dim=16, field_size=39, batch_size=1024, mlp_dtype="fp32", sparse=False)
    g = WideDeepEngine(cfg, dev)
    c = OracleWideDeepEngine(cfg, "cpu")
    d0 = c.deep.numpy().copy()
    w0 = c.wide.numpy().copy()
    untouched = np.ones(cfg.vocab_size, bool)
    for s in range(3):
        ids, wts, label = synthetic_batch(cfg, "cpu", "zipf", seed=17 + s)
        untouched[ids.numpy().reshape(-1)] = False
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 1e-5 * max(abs(lc), 1e-3)
    a, b = g.deep.cpu().numpy(), c.deep.numpy()
    assert row_rel(a, b) <= 2e-5, row_rel(a, b)
    assert (np.abs(b - d0).max(axis=1) > 0).all()                    # dense Adam: no row stays where it was
    assert untouched.sum() > 1_000_000                               # rows that saw only the L2 pull
    assert np.array_equal(a[untouched], b[untouched])                 # ... and those agree bit for bit
    aw, bw = g.wide.cpu().numpy(), c.wide.numpy()
    assert np.abs(aw - bw).max() <= 1e-4 * np.abs(w0).max()
    assert (bw[untouched] == 0).all() and (w0[untouched] != 0).any()   # dense FTRL re-derives w from linear = 0
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=1e-4, atol=1e-6)
    # the deep optimizer's loss: + l2_coef * sum(E^2) / 2 at the last step's STARTING values (the forward's, wide_and_deep.py:356-360),
    # a by-product of that step's Adam pass; the oracle-side engine states the same over numpy
    dl, dlc = g.deep_loss(lg), c.deep_loss(lc)
    assert dl > lg and abs(dl - dlc) <= 2e-6 * dl
    assert abs((dl - lg) - cfg.l2_coef * 0.5 * float((b.astype(np.float64) ** 2).sum())) <= 0.1 * (dl - lg)      # (close to the final table's)


def test_predict_and_lookup(dev, oracle):
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=10_000, emb_dim=16, field_size=39, batch_size=64, deep_layer_dim=[32], mlp_dtype="fp32")
    g = WideDeepEngine(cfg, dev)
    ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=3)
    emb, wide, _ = g.lookup(ids, wts)
    table = oracle.fill_normal(cfg.seed, cfg.vocab_size, 16, 0.01)
    assert np.array_equal(emb.cpu().numpy().reshape(64, 39, 16), oracle.gather_rows(table, ids.cpu().numpy(), wts.cpu().numpy()))
    logit, prob = g.predict(ids, wts)
    assert logit.shape == (64, 1) and float(prob.min()) > 0 and float(prob.max()) < 1


def test_deep_cross_engine_matches_oracle_engine(dev, oracle):
    """DCN step (gather, 6 fused cross layers, MLP, dense Adam over the table) vs the oracle-driven engine."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.deep_cross import DeepCrossConfig, DeepCrossEngine
    from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
    cfg = DeepCrossConfig(vocab_size=5000, emb_dim=30, field_size=39, batch_size=128, deep_layer_dim=[64, 32])
    g = DeepCrossEngine(cfg, dev)
    c = OracleDeepCrossEngine(cfg, "cpu")
    assert np.array_equal(g.table.cpu().numpy(), c.table.numpy())
    bcfg = WideDeepConfig(vocab_size=5000, emb_dim=30, field_size=39, batch_size=128)
    for s in range(3):
        ids, wts, label = synthetic_batch(bcfg, "cpu", "zipf", seed=21 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 2e-5 * max(abs(lc), 1e-3)
    a, b = g.table.cpu().numpy(), c.table.numpy()
    assert row_rel(a, b) <= 5e-5, row_rel(a, b)
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=2e-4, atol=2e-6)
    logit, prob = g.predict(ids.to(dev), wts.to(dev))
    assert logit.shape == (128, 1)


def test_auc_parity_on_planted_signal(dev, oracle):
    """BASELINE "AUC parity": the GPU engine and the oracle-driven engine, trained on the same
    synthetic stream with a planted signal, reach the same held-out AUC (and both learn)."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from sklearn.metrics import roc_auc_score
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=3000, emb_dim=16, field_size=39, batch_size=1024, deep_layer_dim=[64, 32],
                         mlp_dtype="fp32", adam_lr=3e-3)          # cfg1-like shape, larger lr so 40 steps suffice
    g = WideDeepEngine(cfg, dev)
    c = OracleWideDeepEngine(cfg, "cpu")
    for s in range(40):
        ids, wts, label = synthetic_batch(cfg, "cpu", "uniform", seed=300 + s, signal=True)
        c.train_step(ids, wts, label)
        g.train_step(ids.to(dev), wts.to(dev), label.to(dev))
    ids, wts, label = synthetic_batch(cfg, "cpu", "uniform", seed=999, signal=True)
    _, pc = c.predict(ids, wts)
    _, pg = g.predict(ids.to(dev), wts.to(dev))
    y = label.numpy().ravel()
    auc_c, auc_g = roc_auc_score(y, pc.numpy().ravel()), roc_auc_score(y, pg.cpu().numpy().ravel())
    assert auc_c > 0.6 and auc_g > 0.6, (auc_c, auc_g)
    assert abs(auc_c - auc_g) < 2e-3, (auc_c, auc_g)


def test_mfma_mlp_step_matches_torch_fp32_reference(dev):
    """Second opinion for the hand-written mixed-precision MLP step (the oracle checks are tests/test_dense_gpu.py and
    tests/test_bench_shape_gpu.py): a plain PyTorch fp32 restatement with the SAME rounding points -- 16-bit operands
    widened to fp32, fp32 matmul + fp32 bias, one cast of every activation to 16 bits (whose autograd casts the gradient
    back at the same point) -- run through autograd on the same GPU."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    for dt in ("bf16", "fp16"):
        cfg = WideDeepConfig(vocab_size=20_000, emb_dim=80, field_size=26, batch_size=2048, mlp_dtype=dt)
        e = WideDeepEngine(cfg, dev)
        assert e._mfma
        amp = e._amp
        ids, wts, label = synthetic_batch(cfg, dev, "zipf", seed=5)
        emb, wide, _ = e.lookup(ids, wts)
        loss, g_emb, g_wide = e._mlp_step_eager(emb, wide, label)
        e._sum_dw_slabs()
        gd1 = e.dense_grad_flat.detach().clone()
        n = len(e.dims) - 1
        params = [p.detach().clone().requires_grad_(True) for p in e.dense]
        x = emb.detach().clone().requires_grad_(True)
        wd = (wide.prod[..., 0].sum(dim=1) + e.wide_b).detach().clone().requires_grad_(True)      # the fused lookup's per-field products
        with torch.enable_grad():
            h = x
            for i in range(n - 1):
                W16 = params[2 * i].to(amp)                         # the operand shadow: one rounding of the fp32 master weight
                h = torch.relu(h.float() @ W16.float() + params[2 * i + 1]).to(amp)
            logit = h.float() @ params[2 * (n - 1)] + params[2 * (n - 1) + 1] + wd.view(-1, 1)
            loss2 = torch.nn.functional.binary_cross_entropy_with_logits(logit, label)
            (loss2 * cfg.sens).backward()
        assert abs(float(loss) - float(loss2.detach())) <= 2e-5 * abs(float(loss2.detach())), dt
        assert torch.allclose(g_wide, wd.grad, rtol=2e-3, atol=1e-6 * float(wd.grad.abs().max())), dt
        ge1, ge2 = g_emb.float(), x.grad.float()
        # row gradients: 16-bit values, a few differ by an ulp (and what that does downstream)
        assert float((ge1 - ge2).abs().max()) <= 0.1 * float(ge2.abs().max()), dt
        assert float((ge1 == ge2).float().mean()) >= 0.9, dt
        for i, p in enumerate(params):
            ref = p.grad
            got = gd1[e.dense_grad[i].storage_offset(): e.dense_grad[i].storage_offset() + ref.numel()].view_as(ref)
            assert float((got - ref).abs().max()) <= 5e-3 * float(ref.abs().max()) + 1e-12, (dt, i)


def test_deepfm_engine_matches_oracle_engine(dev, oracle):
    """DeepFM step (two gathers, FM term, MLP, L2 over both whole tables, dense Adam) vs the oracle-driven engine."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.deepfm import DeepFMConfig, DeepFMEngine
    from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
    cfg = DeepFMConfig(data_vocab_size=4000, data_emb_dim=16, data_field_size=39, batch_size=128, deep_layer_dims=[64, 32], mlp_dtype="fp32")
    g = DeepFMEngine(cfg, dev)
    c = OracleDeepFMEngine(cfg, "cpu")
    assert np.array_equal(g.V_l2.cpu().numpy(), c.V_l2.numpy())
    bcfg = WideDeepConfig(vocab_size=4000, emb_dim=16, field_size=39, batch_size=128)
    for s in range(3):
        ids, wts, label = synthetic_batch(bcfg, "cpu", "zipf", seed=40 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 2e-5 * max(abs(lc), 1e-3)
    assert row_rel(g.V_l2.cpu().numpy(), c.V_l2.numpy()) <= 5e-5
    assert np.abs(g.W_l2.cpu().numpy() - c.W_l2.numpy()).max() <= 1e-4 * np.abs(c.W_l2.numpy()).max()
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=2e-4, atol=2e-6)
    logit, prob = g.predict(ids.to(dev), wts.to(dev))
    lc2, _ = c.predict(ids, wts)
    assert np.allclose(logit.cpu().numpy(), lc2.numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("level", ["step", "front", "mlp"])
def test_graph_replay_is_bit_identical_to_eager(dev, level):
    """The captured graphs (the whole step with the optimizers, the front of the step, or the MLP alone) hold the same
    kernels in the same order on the same buffers as the eager path: losses, tables and dense parameters must agree bit
    for bit over several steps (capture on step 3, replay from then on, a different batch every step).  In the whole-step
    graph the Adam step size comes from the device-side step state, advanced by a kernel inside the graph."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=50000, emb_dim=16, field_size=26, batch_size=2048, deep_layer_dim=[256, 128, 64, 32],
              mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(graphs=level, **kw), dev)
    b = WideDeepEngine(WideDeepConfig(graphs="none", **kw), dev)
    assert a._mfma and a._fold_wide, "MFMA MLP path with the folded wide branch expected"
    for s in range(9):
        ids, wts, label = synthetic_batch(a.cfg, dev, "zipf", seed=70 + s)
        la, lb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        assert la == lb, (s, la, lb)
    got = {"step": a._step_graph, "front": a._front_graph, "mlp": a._mlp_graph}
    assert got[level] is not None and all(v is None for k, v in got.items() if k != level), {k: v is not None for k, v in got.items()}
    assert b._mlp_graph is None and b._front_graph is None and b._step_graph is None
    assert torch.equal(a.deep, b.deep) and torch.equal(a.wide, b.wide)
    assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
    # device-side step scalars == the host mirrors (same fp32 products in the same order)
    st = a._step_state.read()
    assert int(st["step"]) == a.step_count == 9
    assert np.float32(st["beta1_power"]) == a.beta1_power and np.float32(st["beta2_power"]) == a.beta2_power
    lr_t = np.float32(a.cfg.adam_lr) * np.sqrt(np.float32(1) - a.beta2_power) / (np.float32(1) - a.beta1_power)
    assert np.float32(st["lr_t"]) == np.float32(lr_t)
    if level == "step":
        ms = a._step_state.apply_ms(range(4, 10))
        assert len(ms) == 6 and all(0.0 < x < 50.0 for x in ms), ms
        # the stamps' runtime switch (mrec_step_state_t.stamps_off): the captured graph reads it when it runs; same results, no stamps
        a._step_state.set_stamps(False)
        for s in range(9, 12):
            ids, wts, label = synthetic_batch(a.cfg, dev, "zipf", seed=70 + s)
            assert float(a.train_step(ids, wts, label)) == float(b.train_step(ids, wts, label)), s
        assert torch.equal(a.deep, b.deep) and torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
        assert a._step_state.apply_ms(range(10, 13)) == []
        a._step_state.set_stamps(True)
        ids, wts, label = synthetic_batch(a.cfg, dev, "zipf", seed=90)
        a.train_step(ids, wts, label)
        assert len(a._step_state.apply_ms(range(13, 14))) == 1


def test_step_graph_survives_checkpoint_restore(dev, tmp_path):
    """load_checkpoint moves the host-side step count and beta powers; the device-side step state follows on the next step
    (graph replay included): a restored engine continues bit-identically to the one that never stopped."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, load_checkpoint, save_checkpoint, synthetic_batch
    kw = dict(vocab_size=20000, emb_dim=16, field_size=26, batch_size=1024, deep_layer_dim=[64, 32], mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    batches = [synthetic_batch(a.cfg, dev, "uniform", seed=5 + s) for s in range(10)]
    for s in range(5):
        a.train_step(*batches[s])
    save_checkpoint(a, str(tmp_path / "ck.pt"))
    b = WideDeepEngine(WideDeepConfig(**kw), dev)
    for s in range(4):                    # b has its own history (and its step graph) before the restore
        b.train_step(*batches[9 - s])
    load_checkpoint(b, str(tmp_path / "ck.pt"))
    for s in range(5, 9):
        la, lb = float(a.train_step(*batches[s])), float(b.train_step(*batches[s]))
        assert la == lb, (s, la, lb)
    assert a._step_graph is not None and b._step_graph is not None
    assert torch.equal(a.deep, b.deep) and torch.equal(a.dense_flat.detach(), b.dense_flat.detach())


@pytest.mark.parametrize("graph", [False, True])
def test_dynamic_embedding_engine_equals_dense_table_engine(dev, graph):
    """--dynamic_embedding=True (hash tables keyed by the raw ids, rows created on first sight with their default
    values) must train exactly like the dense-table engine when the keys happen to be valid row numbers: default
    rows are the same counter-based N(0, 0.01) values keyed by (seed, id, column), every kernel downstream of the
    index probe is shared, and the segment sums are grouped in the same first-occurrence order."""
    from mindrec_amd import ops
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=50000, emb_dim=16, field_size=39, batch_size=1024, deep_layer_dim=[128, 64, 32],
              mlp_dtype="bf16", graphs="step" if graph else "none",
              const_columns=False)      # (two code paths compared bit for bit: the dense-table engine's constant-column sums are another order)
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    b = WideDeepEngine(WideDeepConfig(dynamic_embedding=True, hash_capacity=1 << 16, **kw), dev)
    seen = []
    for s in range(6):
        ids, wts, label = synthetic_batch(a.cfg, dev, "zipf", seed=90 + s)
        la, lb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        assert la == lb, (s, la, lb)
        seen.append(ids.reshape(-1))
    keys = torch.unique(torch.cat(seen)).to(torch.int64)
    assert len(b.index) == keys.numel()
    rows, _ = b.index.find_or_insert(keys.contiguous(), insert=False)
    assert bool((rows >= 0).all())
    r = rows.long()
    for name in ("deep", "deep_m", "deep_v", "wide", "wide_accum", "wide_linear"):
        assert torch.equal(getattr(a, name)[keys], getattr(b, name)[r]), name
    assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
    la, _ = a.predict(ids, wts)
    lb, _ = b.predict(ids, wts)
    assert torch.equal(la, lb)


def test_dynamic_embedding_checkpoint_roundtrip(dev, tmp_path):
    """A hash-table engine saved after three steps and restored into a fresh engine (which numbers its rows its own
    way) continues bit-identically."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, load_checkpoint, save_checkpoint, synthetic_batch
    kw = dict(vocab_size=50000, emb_dim=16, field_size=26, batch_size=512, deep_layer_dim=[64, 32], mlp_dtype="bf16",
              dynamic_embedding=True, hash_capacity=1 << 15)
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    for s in range(3):
        a.train_step(*synthetic_batch(a.cfg, dev, "zipf", seed=300 + s))
    save_checkpoint(a, tmp_path / "dyn.pt")
    b = WideDeepEngine(WideDeepConfig(**kw), dev)
    # give b's index a different history first?  no: it must be fresh -- a used engine is refused
    load_checkpoint(b, tmp_path / "dyn.pt")
    assert len(b.index) == len(a.index)
    for s in range(3, 6):
        batch = synthetic_batch(a.cfg, dev, "zipf", seed=300 + s)
        assert float(a.train_step(*batch)) == float(b.train_step(*batch))
    with pytest.raises(ValueError):
        load_checkpoint(b, tmp_path / "dyn.pt")


def test_host_cached_tables_engine_equals_resident_engine(dev):
    """host_cache_rows > 0 (the reference's vocab_cache_size): both tables live in pinned host DRAM behind a device
    cache far smaller than the vocabulary, rows are evicted, written back and re-fetched as the Zipf stream moves --
    and the engine trains bit-identically to the fully resident one; the flushed host table equals the resident tables."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=30000, emb_dim=16, field_size=26, batch_size=256, deep_layer_dim=[64, 32], mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(fold_wide=False, **kw), dev)      # the cache tier runs the separate wide kernels: same sum trees
    b = WideDeepEngine(WideDeepConfig(host_cache_rows=8000, **kw), dev)
    for s in range(12):
        ids, wts, label = synthetic_batch(a.cfg, dev, "uniform" if s % 3 == 0 else "zipf", seed=500 + s)
        la, lb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        assert la == lb, (s, la, lb)
    st = b.hb.stats
    assert st["evictions"] > 0 and st["misses"] > 8000 and st["hits"] > 0
    full = b.hb.full_table()                                       # [V, 3D + 4] on the host: p | m | v | w accum linear pad
    D = 16
    ref = torch.cat([a.deep, a.deep_m, a.deep_v, a.wide, a.wide_accum, a.wide_linear], dim=1).cpu()
    assert torch.equal(full[:, : 3 * D + 3], ref)
    assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())


def test_host_cached_engine_matches_oracle_engine(dev, oracle):
    """The cache tier against the ORACLE (not against the resident HIP engine): the engine whose tables live in host DRAM
    behind a small device cache -- rows evicted, written back and fetched again as the stream moves -- against the engine
    driven by the oracle on the CPU, which knows nothing of caches."""
    from _oracle_engine import OracleDeepCrossEngine, OracleDeepFMEngine, OracleWideDeepEngine  # noqa: F401
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=40_000, emb_dim=16, field_size=26, batch_size=256, deep_layer_dim=[64, 32], mlp_dtype="fp32")
    g = WideDeepEngine(WideDeepConfig(host_cache_rows=9000, **kw), dev)
    c = OracleWideDeepEngine(WideDeepConfig(**kw), "cpu")
    for s in range(8):
        ids, wts, label = synthetic_batch(c.cfg, "cpu", "uniform" if s % 2 else "zipf", seed=300 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 1e-5 * max(abs(lc), 1e-3), (s, lc, lg)
    st = g.hb.stats
    assert st["evictions"] > 0 and st["hits"] > 0 and st["misses"] > 9000
    full = g.hb.full_table().numpy()                                   # [V, 3D + 4]: p | m | v | w accum linear pad
    D = 16
    assert row_rel(full[:, :D], c.deep.numpy()) <= 2e-5
    assert np.abs(full[:, 3 * D] - c.wide.numpy()[:, 0]).max() <= 1e-4 * np.abs(c.wide.numpy()).max()
    untouched = (c.deep_m.numpy() == 0).all(axis=1)
    assert np.array_equal(full[untouched, :D], c.deep.numpy()[untouched])
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=1e-4, atol=1e-6)


def test_hash_tables_behind_the_host_cache_equal_resident_hash_tables(dev):
    """dynamic_embedding + host_cache_rows (BASELINE configs[4]: MapParameter tables larger than HBM): keys -> host rows by
    a second device key index, the cache tier below unchanged; trains bit-identically to the resident hash-table engine and
    ends with the same value for every key."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine
    kw = dict(vocab_size=1, emb_dim=16, field_size=26, batch_size=256, deep_layer_dim=[64, 32], mlp_dtype="bf16",
              dynamic_embedding=True, hash_capacity=1 << 16)
    a = WideDeepEngine(WideDeepConfig(fold_wide=False, **kw), dev)
    b = WideDeepEngine(WideDeepConfig(host_cache_rows=7000, **kw), dev)
    pool = torch.randint(-2 ** 62, 2 ** 62, (30000,), dtype=torch.int64, generator=torch.Generator().manual_seed(1))
    for s in range(10):
        g = torch.Generator().manual_seed(900 + s)
        ids = pool[(torch.rand(256, 26, generator=g) ** (3 if s % 3 else 1) * 30000).long().clamp_(0, 29999)].to(dev)
        wts = (torch.rand(256, 26, generator=g) > 0.1).float().to(dev)
        label = (torch.rand(256, 1, generator=g) < 0.3).float().to(dev)
        la, lb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        assert la == lb, (s, la, lb)
    st = b.hb.stats
    assert st["evictions"] > 0 and st["hits"] > 0
    keys, rows = b.hb.export_hashed()
    ka, ra = a.index.export()
    assert set(keys.tolist()) == set(ka.cpu().tolist())
    order_b, order_a = torch.argsort(keys), torch.argsort(ka.cpu())
    ref = torch.cat([a.deep, a.deep_m, a.deep_v, a.wide, a.wide_accum, a.wide_linear], dim=1)[ra.long()].cpu()[order_a]
    assert torch.equal(rows[order_b][:, : 3 * 16 + 3], ref)
    assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())


def test_weight_gradient_slabs_and_graph_switching(dev):
    """One GPU: the weight gradients stay fp32 batch slabs and the dense-Adam kernel adds them up.  The slabs are
    persistent engine buffers, so switching between the whole-step graph, the MLP graphs (phase timers on) and
    eager steps must not change which buffers the Adam reads (round-1 ADVICE: stale split-K partials after a switch):
    an engine that toggles its timers mid-run trains bit-identically to an eager one."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=50000, emb_dim=80, field_size=26, batch_size=4096, mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    b = WideDeepEngine(WideDeepConfig(graphs="none", **kw), dev)
    for s in range(12):
        if s == 5:
            a.timers = {}              # leaves the step graph: MLP graphs get captured
        if s == 8:
            a.timers = None            # back to the step graph
        batch = synthetic_batch(a.cfg, dev, "zipf", seed=700 + s)
        la, lb = float(a.train_step(*batch)), float(b.train_step(*batch))
        assert la == lb, (s, la, lb)
    assert a._step_graph is not None and a._mlp_graph is not None
    assert len(a._dw) == 4 and a._dw[0].dtype == torch.float32          # four hidden layers, fp32 slabs
    assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
    assert torch.equal(a.deep, b.deep)


def test_folded_wide_branch_equals_separate_wide_kernels(dev):
    """fold_wide (default on one GPU): the wide lookup rides the deep gather, the per-sample sum is taken inside the head and
    the wide FTRL rides the deep LazyAdam apply.  Same adds in the same order as the separate kernels wherever an id's run
    of duplicates fits the short-run path, a different (fixed) tree above it: losses and deep tables bit-identical on uniform
    ids, wide table within 1e-6 of its scale on Zipf ids with hot rows."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=40000, emb_dim=80, field_size=26, batch_size=2048, deep_layer_dim=[128, 64], mlp_dtype="bf16")
    for dist_kind in ("uniform", "zipf"):
        a = WideDeepEngine(WideDeepConfig(fold_wide=True, **kw), dev)
        b = WideDeepEngine(WideDeepConfig(fold_wide=False, **kw), dev)
        assert a._fold_wide and not b._fold_wide
        for s in range(6):
            batch = synthetic_batch(a.cfg, dev, dist_kind, seed=40 + s)
            la, lb = float(a.train_step(*batch)), float(b.train_step(*batch))
            if dist_kind == "uniform":
                assert la == lb, (s, la, lb)
            else:
                assert abs(la - lb) <= 1e-6 * abs(lb)
        if dist_kind == "uniform":
            assert torch.equal(a.deep, b.deep) and torch.equal(a.wide, b.wide) and torch.equal(a.wide_accum, b.wide_accum)
            assert torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
        else:
            assert float((a.wide - b.wide).abs().max()) <= 1e-5 * float(b.wide.abs().max())
            assert float((a.deep - b.deep).abs().max()) <= 1e-5 * float(b.deep.abs().max())


def test_sink_of_steps_equals_step_by_step(dev):
    """train_steps (the reference's dataset_sink_mode / sink_size): S steps replayed as ONE graph must equal S train_step calls
    bit for bit -- losses, tables, dense parameters, the device-side step state; a sink of another size, a shorter remainder and
    a step-by-step call in between keep working; Dropout's masks move with the step inside the sink."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=50000, emb_dim=16, field_size=26, batch_size=2048, deep_layer_dim=[256, 128, 64, 32], mlp_dtype="bf16",
              dropout_flag=True)
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    b = WideDeepEngine(WideDeepConfig(**kw), dev)
    bs = [synthetic_batch(a.cfg, dev, "zipf", seed=600 + s) for s in range(24)]
    la, lb = [], []
    i = 0
    for chunk in (4, 1, 3, 3, 1, 3, 4, 2, 3):                     # the first sinks run step by step (no graph yet), then 2-, 3- and 4-step graphs
        la += [float(x) for x in a.train_steps(bs[i:i + chunk])]
        lb += [float(b.train_step(*bs[i + j])) for j in range(chunk)]
        i += chunk
    assert la == lb, (la, lb)
    assert set(k[0] for k, v in a._sink_graphs.items() if v) == {2, 3, 4} and not b._sink_graphs
    assert torch.equal(a.deep, b.deep) and torch.equal(a.wide, b.wide) and torch.equal(a.dense_flat.detach(), b.dense_flat.detach())
    sa, sb = a._step_state.read(), b._step_state.read()
    assert int(sa["step"]) == int(sb["step"]) == a.step_count == b.step_count == 24
    assert float(sa["beta1_power"]) == float(sb["beta1_power"]) == float(a.beta1_power)


def test_fm_kernels_with_the_models_other_terms(dev, oracle):
    """mrec_fm_fwd_add_f32 / mrec_fm_bwd_mix_f32 (the DeepFM step's glue around the 16-bit net) against numpy: fm + linear, and
    widen(g16) + dout * (colsum - vx), bit-exact (same fp32 operations in the same order)."""
    from mindrec_amd import ops
    rng = np.random.default_rng(4)
    B, Fd, D = 700, 13, 80
    vx = rng.standard_normal((B, Fd, D)).astype(np.float32)
    lin = rng.standard_normal(B).astype(np.float32)
    tv = torch.from_numpy(vx).to(dev)
    fm0, cs0 = ops.fm_forward(tv)
    fm1, cs1 = ops.fm_forward(tv, add=torch.from_numpy(lin).to(dev))
    assert torch.equal(cs0, cs1) and np.array_equal(fm1.cpu().numpy(), lin + fm0.cpu().numpy())
    dout = rng.standard_normal(B).astype(np.float32)
    for tdt, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        g16 = oracle.round16(rng.standard_normal((B, Fd, D)).astype(np.float32), name)
        got = ops.fm_backward_mix(torch.from_numpy(g16).to(dev).to(tdt), tv, cs0, torch.from_numpy(dout).to(dev)).cpu().numpy()
        cs = cs0.cpu().numpy()
        ref = g16 + dout[:, None, None] * (cs[:, None, :] - vx)
        assert np.array_equal(got, ref.astype(np.float32))


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_deepfm_engine_in_the_references_precision(dev, dt):
    """DeepFM with convert_dtype (the reference's default: DenseLayer in float16, models/deepfm/default_config.yaml:27) runs its
    dense net on the hand-written MFMA kernels -- no library GEMM, the MLP step replayed as HIP graphs -- and tracks the fp32
    oracle-side engine to 16-bit accuracy: losses, both tables, the dense parameters."""
    from _oracle_engine import OracleDeepFMEngine
    from mindrec_amd.deepfm import DeepFMConfig, DeepFMEngine
    from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
    kw = dict(data_vocab_size=4000, data_emb_dim=16, data_field_size=39, batch_size=256, deep_layer_dims=[128, 64, 32, 16])
    g = DeepFMEngine(DeepFMConfig(mlp_dtype=dt, **kw), dev)
    c = OracleDeepFMEngine(DeepFMConfig(mlp_dtype="fp32", **kw), "cpu")
    assert g._mfma and torch.equal(g.dense_flat.detach().cpu()[:c.dense_flat.numel()], c.dense_flat.detach()[:g.dense_flat.numel()])
    bcfg = WideDeepConfig(vocab_size=4000, emb_dim=16, field_size=39, batch_size=256)
    for s in range(6):
        ids, wts, label = synthetic_batch(bcfg, "cpu", "zipf", seed=70 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lg - lc) <= 3e-3 * abs(lc), (s, lg, lc)
    assert g._mlp_graph is not None
    for a, b, name in ((g.V_l2, c.V_l2, "V"), (g.W_l2, c.W_l2, "W")):
        a, b = a.cpu().numpy(), b.numpy()
        assert np.abs(a - b).max() <= 2 * g.cfg.learning_rate * 6, name          # Adam: a flipped sign of a ~0 gradient is +-lr per step
        assert np.mean(np.abs(a - b) <= 0.1 * g.cfg.learning_rate) > 0.97, name
    n = c.dense_flat.numel()
    d = np.abs(g.dense_flat.detach().cpu().numpy()[:n] - c.dense_flat.detach().numpy()[:n])
    assert d.max() <= 2 * g.cfg.learning_rate * 6 and np.mean(d <= 0.1 * g.cfg.learning_rate) > 0.9
    ids, wts, _ = synthetic_batch(bcfg, "cpu", "zipf", seed=99)
    pg, pc = g.predict(ids.to(dev), wts.to(dev))[1].cpu().numpy(), c.predict(ids, wts)[1].numpy()
    assert np.abs(pg - pc).max() <= 2e-2


def test_engine_refuses_a_net_without_hip_path(dev):
    """No library-GEMM / autograd fallback in the product: a dense net the hand-written kernels do not cover (here a 16-bit net
    whose widths are not multiples of 8: rows that are not 16-byte aligned) raises UnsupportedNet (MREC_EUNSUPPORTED)."""
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    from mindrec_amd.wide_deep_mlp import UnsupportedNet
    cfg = WideDeepConfig(vocab_size=5000, emb_dim=16, field_size=5, batch_size=64, deep_layer_dim=[36, 20], mlp_dtype="fp16")
    eng = WideDeepEngine(cfg, dev)
    assert not eng._mfma and not eng._f32net
    with pytest.raises(UnsupportedNet, match="MREC_EUNSUPPORTED"):
        eng.train_step(*synthetic_batch(cfg, dev, "uniform", seed=1))
    with pytest.raises(UnsupportedNet, match="MREC_EUNSUPPORTED"):
        eng.predict(*synthetic_batch(cfg, dev, "uniform", seed=1)[:2])


@pytest.mark.parametrize("dropout", [False, True])
def test_fp32_net_with_a_wide_last_layer_matches_oracle_engine(dev, oracle, dropout):
    """The reference's benchmark net ends 1024 -> 1 (benchmarks/wide_deep/default_config.yaml:12): wider than the output-head
    kernel's 512 columns, so the fp32 engine takes its output end through the Deep&Cross head kernel ([h | wide, 0] . [W5 | 1, 0]).
    Against the oracle-side engine (torch autograd on the CPU), with and without Dropout."""
    from _oracle_engine import OracleWideDeepEngine
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    cfg = WideDeepConfig(vocab_size=20_000, emb_dim=16, field_size=39, batch_size=128, deep_layer_dim=[64, 1024], mlp_dtype="fp32",
                         dropout_flag=dropout)
    g, c = WideDeepEngine(cfg, dev), OracleWideDeepEngine(cfg, "cpu")
    assert g._f32net and not g.k.head_supported(1024)
    for s in range(3):
        ids, wts, label = synthetic_batch(cfg, "cpu", "zipf", seed=31 + s)
        lc = float(c.train_step(ids, wts, label))
        lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
        assert abs(lc - lg) <= 1e-5 * max(abs(lc), 1e-3)
    assert row_rel(g.deep.cpu().numpy(), c.deep.numpy()) <= 2e-5
    assert np.allclose(g.dense_flat.detach().cpu().numpy(), c.dense_flat.detach().numpy(), rtol=1e-4, atol=1e-6)
