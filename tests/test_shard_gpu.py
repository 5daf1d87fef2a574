"""The kernels of the fixed-capacity shard protocol (csrc/mrec_route.hip, the skip-negative plan, the clamped apply, the
strided gather) through the C ABI against the oracle's restatements (oracle/oracle.py: shard_route_slots, unique_skip_negative,
gather_rows, sparse_lazy_adam, sparse_ftrl).  Integer results bit-exact; fp32 rows bit-exact where the summation order is the
oracle's (runs inside one apply window), 1e-5 row-relative otherwise."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("world,hashed,factor", [(1, False, 1.25), (2, False, 1.25), (8, False, 1.25), (5, True, 1.5), (4, False, 0.9)])
def test_route_slots_matches_oracle(dev, idt, world, hashed, factor):
    """Slots, padding, both index maps and the dropped-position count (factor 0.9: buckets overflow), bit-exact."""
    from mindrec_amd import ops
    rng = np.random.default_rng(7)
    n = 5000
    ids = rng.integers(0, 1 << 40 if (hashed and idt == np.int64) else 100_000, size=n).astype(idt)
    ids[::7] = ids[0]                                    # a hot id: its owner's bucket fills first
    if hashed:
        ids[5] = -123456                                 # hash tables take any key but -1 / -2
    wts = rng.random(n).astype(np.float32)
    cap = O.shard_capacity(n, world, factor)
    assert cap == ops.shard_capacity(n, world, factor)
    rid, rw, sop, pos, dropped = O.shard_route_slots(ids, wts, world, cap, hashed)
    ov = torch.zeros(1, dtype=torch.int64, device=dev)
    req, g_sop, g_pos = ops.shard_route_slots(T(ids, dev), T(wts, dev), world, cap, hashed=hashed, overflow=ov)
    g_ids, g_wts = ops.shard_unpack_req(req)
    assert np.array_equal(g_sop.cpu().numpy(), sop) and np.array_equal(g_pos.cpu().numpy(), pos)
    assert np.array_equal(g_ids.cpu().numpy().astype(np.int64), rid) and np.array_equal(g_wts.cpu().numpy(), rw)
    assert int(ov.item()) == dropped and (dropped > 0 or factor >= 1.0)       # (factor 0.9: some bucket must overflow; the hot id may fill one at 1.25 too)
    # second call accumulates (sticky counter)
    ops.shard_route_slots(T(ids, dev), T(wts, dev), world, cap, hashed=hashed, overflow=ov)
    assert int(ov.item()) == 2 * dropped
    # every valid slot points back at its position, owners are right
    ok = pos >= 0
    assert np.array_equal(sop[pos[ok]], np.nonzero(ok)[0])
    assert np.array_equal(np.nonzero(ok)[0] // cap, O.shard_owner(ids[pos[ok]], world, hashed))


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 0), (2, 1), (5, 3), (8, 7)])
def test_route_slots_with_rotated_chunks(dev, world, rank):
    """chunk_rot = rank + 1 (what the engine passes so that a rank's own chunk comes last): against the oracle, written into a
    caller-provided window of a larger buffer, and the own chunk is the last one."""
    from mindrec_amd import ops
    rng = np.random.default_rng(3 + world)
    n = 4000
    ids = rng.integers(0, 100_000, size=n).astype(np.int32)
    wts = rng.random(n).astype(np.float32)
    cap = O.shard_capacity(n, world, 1.5)
    rot = (rank + 1) % world
    rid, rw, sop, pos, dropped = O.shard_route_slots(ids, wts, world, cap, False, rot=rot)
    ov = torch.zeros(1, dtype=torch.int64, device=dev)
    X = torch.full(((2 * world - 1) * cap, 2), 77, dtype=torch.int32, device=dev)
    req, g_sop, g_pos = ops.shard_route_slots(T(ids, dev), T(wts, dev), world, cap, overflow=ov, rot=rot, out=X[: world * cap])
    assert req.data_ptr() == X.data_ptr() and bool((X[world * cap:] == 77).all())
    g_ids, g_wts = ops.shard_unpack_req(req)
    assert np.array_equal(g_sop.cpu().numpy(), sop) and np.array_equal(g_pos.cpu().numpy(), pos) and int(ov.item()) == dropped == 0
    assert np.array_equal(g_ids.cpu().numpy().astype(np.int64), rid) and np.array_equal(g_wts.cpu().numpy(), rw)
    ok = pos >= 0
    own = O.shard_owner(ids[pos[ok]], world, False)
    assert np.array_equal(np.nonzero(ok)[0] // cap, (own - rot) % world)
    assert set((np.nonzero(ok)[0] // cap)[own == rank]) <= {world - 1}          # the rank's own rows: the last chunk
    with pytest.raises(RuntimeError):
        ops.shard_route_slots(T(ids, dev), T(wts, dev), world, cap, overflow=ov, rot=world)


@pytest.mark.parametrize("act", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_answer_message_unroute_and_gradient_message(dev, act, idt):
    """Owner: the strided in-place gather writes [row | wide product, 0 | pad] for the valid slots only.  Requester: unroute ==
    the one-GPU fused lookup on the original ids; the gradient message carries g[pos] and dlogit[pos // F] slot by slot."""
    from mindrec_amd import ops
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[act]
    rng = np.random.default_rng(11)
    V, D, B, F, world = 3000, 80, 64, 13, 1          # one shard: the owner's rows are the table itself
    n = B * F
    ld = 256
    state = torch.zeros((V, ld), dtype=torch.float32, device=dev)
    ops.fill_normal_(state[:, :D], 5, 0.01)
    ops.fill_normal_(state[:, D:D + 1], 6, 0.01)
    table = state[:, :D]
    ids = rng.integers(0, V, size=(B, F)).astype(idt)
    ids[0, 0] = V + 5                                    # out of range: a zero row, as EmbeddingLookup gives
    wts = rng.random((B, F)).astype(np.float32)
    cap = ops.shard_capacity(n, world)
    assert cap == n
    ov = torch.zeros(1, dtype=torch.int64, device=dev)
    req, sop, pos = ops.shard_route_slots(T(ids, dev), T(wts, dev), world, cap, overflow=ov)
    Dw, W = ops.shard_msg_words(D, tdt)
    if idt == np.int32:
        ans = ops.gather_rows_req(table, req, 2, req.view(torch.float32).view(-1)[1:], 2, n, D, tdt)
    else:
        ans = ops.gather_rows_req(table, req, 2, req.view(torch.float32).view(-1)[2:], 4, n, D, tdt)
    emb, wprod = ops.shard_unroute_slots(ans, sop, D, tdt)
    tp = state.cpu().numpy()
    ref_rows = O.gather_rows(tp[:, :D].copy(), ids, wts).reshape(n, D)
    if act != "fp32":
        ref_rows = O.round16(ref_rows, act)
    ref_w = O.gather_rows(tp[:, D:D + 1].copy(), ids, wts).reshape(n)
    assert np.array_equal(emb.float().cpu().numpy(), ref_rows)
    assert np.array_equal(wprod.cpu().numpy()[:, 0], ref_w) and not wprod.cpu().numpy()[:, 1].any()
    if act != "fp32":
        # ... and it is what the one-GPU fused lookup hands out, bit for bit
        e1, w1 = ops.gather_rows_wide(table, T(ids, dev), T(wts, dev), D, out_dtype=tdt)
        assert torch.equal(e1.view(n, D), emb) and torch.equal(w1.view(n, 2), wprod)
    # gradient message
    g = torch.randn((n, D), device=dev).to(tdt)
    dl = torch.randn(B, device=dev)
    msg = ops.shard_route_grads(g, dl, F, pos)
    rows = (msg[:, :D] if act == "fp32" else msg.view(tdt)[:, :D]).float().cpu().numpy()
    p = pos.cpu().numpy()
    assert np.array_equal(rows, g.float().cpu().numpy()[p])
    assert np.array_equal(msg[:, Dw].cpu().numpy(), dl.cpu().numpy()[p // F])


def test_gather_req_skips_padding_slots(dev):
    from mindrec_amd import ops
    V, D = 500, 16
    state = torch.zeros((V, 64), dtype=torch.float32, device=dev)
    ops.fill_normal_(state[:, :D], 5, 0.01)
    rows = torch.tensor([3, -1, 7, -1], dtype=torch.int32, device=dev)
    wts = torch.ones(4, device=dev)
    Dw, W = ops.shard_msg_words(D, torch.bfloat16)
    msg = torch.full((4, W), float("nan"), dtype=torch.float32, device=dev)       # poison: what the kernel must not touch
    ops.gather_rows_req(state[:, :D], rows, 1, wts, 1, 4, D, torch.bfloat16, out=msg)
    m = msg.cpu().numpy()
    assert np.isnan(m[1]).all() and np.isnan(m[3]).all()            # padding slots: left alone
    assert not np.isnan(m[0, :Dw + 2]).any() and not np.isnan(m[2, :Dw + 2]).any()


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("n,frac_pad,hot", [(6000, 0.25, True), (6000, 0.0, True), (300, 0.5, False), (40_000, 0.2, True), (257, 1.0, False)])
def test_plan_skip_negative_matches_oracle(dev, idt, n, frac_pad, hot):
    """Unique over the non-negative ids in first-occurrence order, inverse -1 at the padding, the index proper = the valid
    positions grouped (ascending position inside a group), n_valid on the device; the padding follows as pseudo-group U."""
    from mindrec_amd import ops
    rng = np.random.default_rng(n)
    ids = rng.integers(0, max(n // 3, 2), size=n).astype(idt)
    if hot:
        ids[rng.random(n) < 0.3] = 5                     # a long run (crosses many apply windows)
    pad = rng.random(n) < frac_pad
    ids[pad] = -1
    u, inv = O.unique_skip_negative(ids)
    plan = ops.sparse_plan(T(ids, dev), skip_negative=True)
    U, nv = int(plan.n_uniq_dev.item()), int(plan.n_valid_dev.item())
    assert U == u.size and nv == int((~pad).sum())
    assert np.array_equal(plan.uniq_buf[:U].cpu().numpy().astype(np.int64), u)
    assert np.array_equal(plan.inv.cpu().numpy(), inv)
    sp, ss, so = plan.sorted_pos.cpu().numpy(), plan.sorted_seg.cpu().numpy(), plan.seg_offsets.cpu().numpy()
    order = np.lexsort((np.arange(n)[~pad], inv[~pad]))           # by group, then position
    assert np.array_equal(sp[:nv], np.arange(n)[~pad][order]) and np.array_equal(ss[:nv], inv[~pad][order])
    assert so[U] == nv and np.array_equal(so[:U], np.searchsorted(ss[:nv], np.arange(U)))
    if nv < n:
        assert sorted(sp[nv:n]) == list(np.nonzero(pad)[0]) and (ss[nv:n] == U).all() and int(plan.uniq_buf[U]) == -1


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_apply_reads_the_gradient_message_in_place_and_stops_at_n_valid(dev, act):
    """The owner's apply: LazyAdam on the deep columns + FTRL on the wide record over a plan with padding, gradients read from
    the [ns, W] message (row stride), one wide gradient per position (F = 1), against the oracle's two restatements."""
    from mindrec_amd import ops
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16}[act]
    rng = np.random.default_rng(3)
    V, D, ns = 2000, 80, 9000
    ids = rng.integers(0, V, size=ns).astype(np.int32)
    ids[rng.random(ns) < 0.25] = 17                      # long run
    pad = np.zeros(ns, bool)
    pad[6500:] = True                                    # the unused tail of the message ...
    pad[rng.random(ns) < 0.05] = True                    # ... and holes (several senders: every block has its own tail)
    ids[pad] = -1
    wts = rng.random(ns).astype(np.float32)
    Dw, W = ops.shard_msg_words(D, tdt)
    g = (rng.standard_normal((ns, D)) * 1.024).astype(np.float32)
    if act != "fp32":
        g = O.round16(g, act)
    gw = (rng.standard_normal(ns) * 1.024).astype(np.float32)
    msg = torch.full((ns, W), float("nan"), dtype=torch.float32, device=dev)       # padding rows: garbage the kernel must not use
    ok = ~pad
    if act == "fp32":
        msg[T(ok, dev), :D] = T(g[ok], dev)
    else:
        msg.view(tdt)[T(ok, dev), :D] = T(g[ok], dev).to(tdt)
    msg[T(ok, dev), Dw] = T(gw[ok], dev)
    ld = 256
    state = torch.zeros((V, ld), dtype=torch.float32, device=dev)
    p, w = state[:, :D], state[:, D:D + 1]
    m, v = state[:, D + 4:2 * D + 4], state[:, 2 * D + 4:3 * D + 4]
    ops.fill_normal_(p, 1000, 0.01)
    ops.fill_normal_(w, 1001, 0.01)
    state[:, D + 1] = 1.0                                # FTRL accum
    rp, rw_ = p.cpu().numpy().copy(), w.cpu().numpy().copy()
    rm, rv = np.zeros_like(rp), np.zeros_like(rp)
    ra, rl = np.ones_like(rw_), np.zeros_like(rw_)
    plan = ops.sparse_plan(T(ids, dev), skip_negative=True)
    rows = msg[:, :D] if act == "fp32" else msg.view(tdt)[:, :D]
    ops.sparse_lazy_adam_wide_(p, m, v, plan, rows, T(wts, dev), msg[:, Dw:Dw + 1], 1, D, grad_scale=1 / 2048)
    torch.cuda.synchronize()
    vi = ids[ok]
    O.sparse_lazy_adam(rp, rm, rv, vi, g[ok], wts[ok], grad_scale=1 / 2048)
    O.sparse_ftrl(rw_, ra, rl, vi, gw[ok].reshape(-1, 1), wts[ok], grad_scale=1 / 2048)
    gp = p.cpu().numpy()
    assert not np.isnan(gp).any() and not np.isnan(state.cpu().numpy()).any()
    den = np.maximum(np.abs(rp).max(axis=1), 1e-30)
    assert float((np.abs(gp - rp).max(axis=1) / den).max()) <= 1e-5
    assert float(np.abs(w.cpu().numpy() - rw_).max() / np.abs(rw_).max()) <= 1e-4
    assert float(np.abs(m.cpu().numpy() - rm).max()) <= 1e-5 * max(np.abs(rm).max(), 1e-30) + 1e-12
    untouched = np.setdiff1d(np.arange(V), vi)
    assert np.array_equal(gp[untouched], O.fill_normal(1000, V, D, 0.01)[untouched])


def test_map_lookup_skips_the_pad_key(dev):
    """Key -1 (reserved by the reference: "any integers except -1, -2", embedding.py:53) is a padding slot: row -1, never
    inserted, and the other keys are numbered as if it were not there."""
    from mindrec_amd import ops
    keys = np.array([50, -1, 7, 50, -1, 9, 7, -1], np.int64)
    ki = ops.KeyIndex(64, dev)
    vals = torch.zeros((64, 8), dtype=torch.float32, device=dev)
    rows = ki.lookup(T(keys, dev), insert=True, tables=[(vals, 0.01, None, 7)], skip_pad=True).cpu().numpy()
    assert list(rows) == [0, -1, 1, 0, -1, 2, 1, -1] and len(ki) == 3
    rows2 = ki.lookup(T(keys, dev), insert=True, tables=[(vals, 0.01, None, 7)], skip_pad=True).cpu().numpy()
    assert np.array_equal(rows, rows2) and len(ki) == 3
