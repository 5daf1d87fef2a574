"""Constant columns (mrec_const_cols_detect + mrec_sparse_apply_next_const_cols, include/mrec.h): the fields of a batch whose id is the
same in every sample -- what the reference's Criteo pipeline gives the 13 dense features (datasets/criteo_1tb/process_data.py:138-147)
-- are summed sample by sample by the folded apply's own launch instead of through the inverted index.  Checked: the detection against
a numpy restatement (ragged cases: a column equal in all but one sample, an id that also occurs in another field, an id outside the
table), and the apply against the oracle's LazyAdam / FTRL restatements and against the same call without the constant-column path
(every other row bit-identical; the constant columns' rows equal to rounding: a different fixed order of additions)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ids(rng, B, F, V, nconst, dtype):
    ids = np.minimum(rng.zipf(1.1, size=(B, F)) + 64, V - 1).astype(dtype)
    ids[:, :nconst] = np.arange(nconst, dtype=dtype)[None, :] + 7          # constant columns 0 .. nconst - 1 (ids 7 ..)
    return ids


def _detect_ref(ids, V, min_count):
    """{field: hot id}: the most frequent id of the field's first 16 samples (the earliest on a tie), in at least a quarter of them, a row
    of the table, in at least min_count of the field's samples and in no other field."""
    B, F = ids.shape
    S = min(16, B)
    out = {}
    for f in range(F):
        col = ids[:S, f]
        best, bc = None, 0
        for i in range(S):
            ci = int((col == col[i]).sum())
            if ci > bc:
                best, bc = int(col[i]), ci
        if 4 * bc < S or not (0 <= best < V):
            continue
        if int((ids[:, f] == best).sum()) < min_count or (np.delete(ids, f, axis=1) == best).any():
            continue
        out[f] = best
    return out


@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_detection_matches_the_restatement(dev, idt):
    from mindrec_amd import ops
    rng = np.random.default_rng(5)
    V, B, F = 5000, 300, 39
    ids = _ids(rng, B, F, V, 13, idt)
    ids[B - 1, 3] = 4000                      # column 3: constant in all samples but the last
    ids[17, 30] = ids[0, 5]                   # column 5's id also occurs in field 30
    ids[:, 9] = V + 2                         # column 9: one id, but not a row of the table
    ids[:, 38] = 11                           # a constant column that is not among the first fields ... and the id of column 4
    ids[rng.random(B) < 0.6, 20] = 3210       # field 20: a dominant id in ~60 % of the samples
    state = ops.const_cols_state(dev)
    tid = torch.from_numpy(ids).to(dev)
    for _ in range(3):                        # (the state is reused batch after batch: the launch's last workgroup clears it)
        got = ops.const_cols_ids(ops.const_cols_detect(tid, V, state))                     # constant columns only
        assert got == _detect_ref(ids, V, B) and sorted(got) == [0, 1, 2, 6, 7, 8, 10, 11, 12]
        assert (state[:4] == 0).all() and (state[136:] == 0).all()      # (word, ticket and the eight copies of the counters: cleared)
        got = ops.const_cols_ids(ops.const_cols_detect(tid, V, state, min_count=B // 8))   # ... and dominant ids
        assert got == _detect_ref(ids, V, B // 8) and sorted(got) == [0, 1, 2, 3, 6, 7, 8, 10, 11, 12, 20] and got[20] == 3210
    # no hot column at all (same state); samples agree by chance in a field that is not constant; more than 64 fields
    ids2 = rng.integers(0, V, size=(64, 26)).astype(idt)
    assert ops.const_cols_mask(ops.const_cols_detect(torch.from_numpy(ids2).to(dev), V, state)) == 0
    ids2[1:5, 4] = ids2[0, 4]
    ids2[:, 20] = 123
    assert ops.const_cols_ids(ops.const_cols_detect(torch.from_numpy(ids2).to(dev), V, state)) == _detect_ref(ids2, V, 64)
    ids3 = _ids(rng, 50, 39, V, 20, idt)
    assert ops.const_cols_mask(ops.const_cols_detect(torch.from_numpy(ids3).to(dev), V, state)) == (1 << 20) - 1
    assert ops.const_cols_detect(torch.zeros((4, 65), dtype=torch.int32, device=dev), V) is None


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("idt,gdt,B,nconst,dom", [(torch.int32, torch.float16, 1024, 13, 0), (torch.int64, torch.bfloat16, 515, 13, 0),
                                                 (torch.int32, torch.float16, 2000, 18, 3), (torch.int32, torch.float32, 96, 1, 0),
                                                 (torch.int32, torch.bfloat16, 777, 0, 5)])
def test_apply_with_constant_columns(dev, oracle, defer, idt, gdt, B, nconst, dom):
    from mindrec_amd import ops
    rng = np.random.default_rng(B + nconst)
    V, D, F = 5000, 80, 39
    ld = 256
    ids = _ids(rng, B, F, V, nconst, np.int64)
    for d in range(dom):                                              # `dom` more fields with a dominant id (40-90 % of the samples, ids 30 ..)
        ids[rng.random(B) < 0.4 + 0.1 * d, 20 + d] = 30 + d
    wts = rng.random((B, F)).astype(np.float32)
    g = (rng.standard_normal((B * F, D)) * 1.024).astype(np.float32)
    dname = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: None}[gdt]
    g16 = oracle.round16(g, dname) if dname else g
    gw = (rng.standard_normal(B) * 1.024).astype(np.float32)
    tid = torch.from_numpy(ids).to(dev, idt)
    twt, tg, tgw = torch.from_numpy(wts).to(dev), torch.from_numpy(g16).to(dev, gdt), torch.from_numpy(gw).to(dev)
    st0 = (rng.standard_normal((V, ld)) * 0.01).astype(np.float32)
    st0[:, D + 1] = 1.0                                              # FTRL accum
    st0[:, D + 2] = 0.0
    st0[:, 2 * D + 4:3 * D + 4] = np.abs(st0[:, 2 * D + 4:3 * D + 4])
    n_dense = 4096
    kw = dict(lr=3.5e-4, beta1_power=0.9, beta2_power=0.999, grad_scale=1 / 1024)

    def run(cc):
        st = torch.from_numpy(st0.copy()).to(dev)
        plan = ops.sparse_plan(tid)
        const = (ops.const_cols_detect(tid, V, min_count=B // 8 if dom else None), tid) if cc else None
        fin = ops.sparse_lazy_adam_wide_(st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4], plan, tg, twt, tgw, F, D, defer=defer,
                                         const_cols=const, **kw)
        if defer:
            dp, dm, dv, dg = (torch.zeros(n_dense, device=dev) for _ in range(4))
            ops.dense_adam_slabs_(dp, dm, dv, dg, [], finish=fin, **kw)
        torch.cuda.synchronize()
        return st.cpu().numpy()

    a, b = run(True), run(False)
    crows = np.concatenate([np.arange(nconst) + 7, np.arange(dom) + 30]).astype(np.int64)       # the rows that take the new path
    other = np.ones(V, bool)
    other[crows] = False
    assert np.array_equal(a[other], b[other]), "rows outside the constant columns must not change"
    assert not np.array_equal(a[crows, :D], st0[crows, :D])
    # the constant columns' rows against the oracle (sequential sums) and against the windows' tree order
    rp, rm, rv = st0[:, :D].copy(), st0[:, D + 4:2 * D + 4].copy(), st0[:, 2 * D + 4:3 * D + 4].copy()
    rw, rwa, rwl = st0[:, D:D + 1].copy(), st0[:, D + 1:D + 2].copy(), st0[:, D + 2:D + 3].copy()
    oracle.sparse_lazy_adam(rp, rm, rv, ids, g16, wts, lr=3.5e-4, b1_pow=0.9, b2_pow=0.999, grad_scale=1 / 1024)
    oracle.sparse_ftrl(rw, rwa, rwl, ids, np.repeat(gw, F).reshape(B * F, 1), wts, grad_scale=1 / 1024)
    for got in (a, b):
        den = np.maximum(np.abs(rp[crows]).max(axis=1), 1e-30)
        assert float((np.abs(got[crows, :D] - rp[crows]).max(axis=1) / den).max()) <= 1e-5
        # (sums of up to 2000 signed terms: compared on the scale of the row, not of an element that cancelled to nearly nothing)
        def close(x, y, tol=1e-5):
            return float((np.abs(x - y).max(axis=-1) / np.maximum(np.abs(y).max(axis=-1), 1e-30)).max()) <= tol
        assert close(got[crows, D + 4:2 * D + 4], rm[crows]) and close(got[crows, 2 * D + 4:3 * D + 4], rv[crows])
        assert close(got[crows, D], rw[crows, 0], 1e-4) and close(got[crows, D + 1], rwa[crows, 0]) and close(got[crows, D + 2], rwl[crows, 0], 1e-4)
    assert (a[:, D + 3] == st0[:, D + 3]).all() and (a[:, 3 * D + 4:] == st0[:, 3 * D + 4:]).all()      # pad words untouched


def test_apply_without_constant_columns_is_unchanged(dev):
    """A batch without a constant column: arming the path changes nothing, bit for bit."""
    from mindrec_amd import ops
    rng = np.random.default_rng(9)
    V, D, B, F = 4000, 80, 700, 26
    ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
    tid = torch.from_numpy(ids).to(dev)
    twt = torch.from_numpy(rng.random((B, F)).astype(np.float32)).to(dev)
    tg = torch.from_numpy(rng.standard_normal((B * F, D)).astype(np.float32)).to(dev, torch.float16)
    tgw = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(dev)
    st0 = torch.from_numpy((rng.standard_normal((V, 256)) * 0.01).astype(np.float32)).to(dev)
    st0[:, D + 1] = 1.0
    st0[:, 2 * D + 4:3 * D + 4].abs_()
    outs = []
    for cc in (False, True):
        st = st0.clone()
        plan = ops.sparse_plan(tid)
        ops.sparse_lazy_adam_wide_(st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4], plan, tg, twt, tgw, F, D, grad_scale=1 / 1024,
                                   const_cols=(ops.const_cols_detect(tid, V), tid) if cc else None)
        outs.append(st)
    assert torch.equal(outs[0], outs[1])


def test_engine_takes_the_constant_columns_path(dev):
    """The engine end to end on Criteo-shaped batches (39 fields, the 13 dense fields on one id each): the first, eager steps find the
    constant columns, the captured step and a sink of steps carry the detection launch and the HOT apply kernel; against the same engine
    with the path off: the same training run to rounding (the constant ids' sums are taken in another fixed order)."""
    from mindrec_amd import ops
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch
    kw = dict(vocab_size=40000, emb_dim=80, field_size=39, batch_size=2048, deep_layer_dim=[128, 64], mlp_dtype="bf16")
    a = WideDeepEngine(WideDeepConfig(**kw), dev)
    b = WideDeepEngine(WideDeepConfig(const_columns=False, **kw), dev)
    assert a._fold_wide and b._fold_wide
    batches = [synthetic_batch(a.cfg, dev, "zipf", seed=70 + s) for s in range(10)]
    la = [float(a.train_step(*x)) for x in batches[:6]] + [float(x) for x in a.train_steps(batches[6:])]
    lb = [float(b.train_step(*x)) for x in batches[:6]] + [float(x) for x in b.train_steps(batches[6:])]
    assert a._hot_seen and a._cc is not None and not b._hot_seen and getattr(b, "_cc", None) is None
    assert ops.const_cols_mask(a._cc[0]) == (1 << 13) - 1 and ops.const_cols_ids(a._cc[0]) == {f: f for f in range(13)}
    assert a._step_graph is not None and any(v for v in a._sink_graphs.values())          # (graphs replayed: the path is inside them)
    assert la[0] == lb[0]                                                                  # (the first loss: before any update)
    assert max(abs(x - y) / abs(y) for x, y in zip(la, lb)) <= 1e-4, (la, lb)
    for x, y in ((a.deep, b.deep), (a.wide, b.wide), (a.dense_flat.detach(), b.dense_flat.detach())):
        assert float((x - y).abs().max()) <= 2e-4 * float(y.abs().max())
    assert not torch.equal(a.deep[:13], torch.zeros_like(a.deep[:13]))
    # the stream changes under the captured graphs: a batch without a constant column (empty mask: the HOT kernel's windows do all
    # the work), one whose field 0 moves to field 5's id (that id then sits in two fields: neither is hot), the old shape again
    g = torch.Generator(device="cpu").manual_seed(3)
    for kind in ("none", "moved", "same"):
        ids, wts, label = (t.clone() for t in batches[2])
        if kind == "none":
            ids[:, :13] = torch.randint(13, a.cfg.vocab_size, (ids.shape[0], 13), generator=g, dtype=torch.int32).to(dev)
        elif kind == "moved":
            ids[:, 0] = 5
        xa, xb = float(a.train_step(ids, wts, label)), float(b.train_step(ids, wts, label))
        want = {"none": 0, "moved": ((1 << 13) - 1) & ~0b100001, "same": (1 << 13) - 1}[kind]
        assert ops.const_cols_mask(a._cc[0]) == want, kind
        assert abs(xa - xb) <= 1e-4 * abs(xb), (kind, xa, xb)
    assert float((a.deep - b.deep).abs().max()) <= 2e-4 * float(b.deep.abs().max())
