"""The host-DRAM feature-cache tier is transparent: a table driven through a small device cache (with
evictions, write-backs, re-fetches and first-touch initialisation) ends bit-identical to a fully
device-resident table, and every lookup along the way returns the same rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dist_kind", ["uniform", "zipf"])
def test_cache_tier_is_transparent(dev, dist_kind):
    from mindrec_amd import ops
    from mindrec_amd.feature_cache import HostBackedTable
    V, D, C, B, F = 6000, 16, 900, 64, 13
    rng = np.random.default_rng(0)
    full = torch.zeros((V, 3 * D), device=dev)
    fp, fm, fv = full[:, :D], full[:, D:2 * D], full[:, 2 * D:]
    ops.fill_normal_(fp, 1000, 0.01)
    hb = HostBackedTable(V, D, C, dev, seed=1000, sigma=0.01)
    b1p = b2p = 1.0
    for step in range(25):
        if dist_kind == "uniform":
            ids = rng.integers(0, V, size=(B, F))
        else:
            ids = np.minimum(rng.zipf(1.2, size=(B, F)) - 1 + (step % 5) * 400, V - 1)
        tid = torch.from_numpy(ids.astype(np.int64)).to(dev)
        wts = torch.from_numpy(rng.random((B, F)).astype(np.float32)).to(dev)
        g = torch.from_numpy(rng.standard_normal((B * F, D)).astype(np.float32)).to(dev)
        b1p *= 0.9; b2p *= 0.999
        # fully resident
        plan_f = ops.sparse_plan(tid)
        emb_f = ops.gather_rows(fp, tid, wts)
        ops.sparse_lazy_adam_(fp, fm, fv, plan_f, g, wts, beta1_power=b1p, beta2_power=b2p)
        # through the cache tier: not one host synchronisation (a synchronising call raises in this mode)
        torch.cuda.set_sync_debug_mode("error")
        try:
            plan_c, rows_pos = hb.prepare(tid)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        emb_c = hb.gather(rows_pos, wts.reshape(-1)).view(B, F, D)
        assert torch.equal(emb_c, emb_f), step
        ops.sparse_lazy_adam_(hb.p, hb.slots[0], hb.slots[1], plan_c, g, wts, beta1_power=b1p, beta2_power=b2p)
    assert hb.stats["evictions"] > 0 and hb.stats["first_touch"] > C       # the cache really cycled
    assert torch.equal(hb.full_table(), full.cpu())                          # weights AND both Adam moments
    assert hb.stats["hits"] > 0 and hb.stats["hits"] + hb.stats["misses"] > 0


def test_cache_too_small_is_reported(dev):
    from mindrec_amd.feature_cache import HostBackedTable
    hb = HostBackedTable(1000, 8, 16, dev)
    hb.prepare(torch.arange(0, 100, dtype=torch.int64, device=dev))      # latched on the device, raised at the next host read
    with pytest.raises(RuntimeError, match="unique ids"):
        hb.check()
    with pytest.raises(RuntimeError, match="unique ids"):
        hb.stats


def test_lru_eviction_by_the_count(dev):
    """The second batch keeps 20 rows, needs 40 more and finds 4 free: the 36 oldest rows not in it go (counters are device words)."""
    from mindrec_amd.feature_cache import HostBackedTable
    hb = HostBackedTable(1000, 8, 64, dev)
    hb.prepare(torch.arange(0, 60, dtype=torch.int64, device=dev))
    hb.check()
    hb.prepare(torch.arange(40, 100, dtype=torch.int64, device=dev))      # 20 hits + 40 misses: 4 free rows + 40 stale rows: fits
    hb.check()
    assert hb.resident == 64 and hb.stats == {"hits": 20, "misses": 100, "evictions": 36, "first_touch": 100}


@pytest.mark.parametrize("W", [240, 81, 4])
def test_move_rows_between_device_and_pinned_host(dev, W):
    """mrec_move_rows_f32 (the cache tier's write-backs and fetches): device-side list length, negative rows skipped, either side
    pinned host memory; against a loop."""
    from mindrec_amd import ops
    g = torch.Generator().manual_seed(3)
    src = torch.randn(1000, W, generator=g)
    host = src.clone().pin_memory()
    dst = torch.zeros(500, W, device=dev)
    n = 300
    sr, dr = torch.randperm(1000, generator=g)[:n], torch.randperm(500, generator=g)[:n]
    sr[::7] = -1
    dr[3::11] = -1
    nd = torch.tensor([250], dtype=torch.int64, device=dev)
    ops.move_rows_(dst, dr.to(dev), host, sr.to(dev), n_dev=nd)          # host -> device, the first 250 pairs
    ref = torch.zeros(500, W)
    for i in range(250):
        if sr[i] >= 0 and dr[i] >= 0:
            ref[dr[i]] = src[sr[i]]
    assert torch.equal(dst.cpu(), ref)
    back = torch.zeros(1000, W).pin_memory()
    ops.move_rows_(back, sr.to(dev), dst, dr.to(dev))                    # device -> host, all n
    torch.cuda.synchronize()
    ref2 = torch.zeros(1000, W)
    for i in range(n):
        if sr[i] >= 0 and dr[i] >= 0:
            ref2[sr[i]] = ref[dr[i]]
    assert torch.equal(back, ref2)
    with pytest.raises(TypeError):
        ops.move_rows_(dst, dr.to(dev).int(), host, sr.to(dev))
