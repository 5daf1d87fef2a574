"""MapParameter as kernels (mrec_map_lookup / evict / export_dirty / put_rows_last) against the oracle's map and a host
dictionary model: duplicates inside a call, first-appearance row numbering, last-wins put, device-side admission counters
and eviction, incremental export / import.  Reference surface: README.md:160-205, RELEASE.md:18, embedding.py:136-206."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
def test_lookup_positions_with_duplicates_matches_sequential_insert(dev, oracle, key_dtype):
    """Every position probes the index itself (no Unique in front); missing keys -- repeated inside the call or not -- get
    rows in order of first appearance, exactly as the oracle's sequential loop hands them out; values are the default rows."""
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    D, cap = 24, 60000
    ki = ops.KeyIndex(cap, dev)
    vals = torch.zeros((cap, D), device=dev)
    slot = torch.full((cap, D), 7.0, device=dev)
    om = oracle.Map(D, cap, seed=11, sigma=0.02)
    tables = [(vals, 0.02, None, 11), (slot, None, 0.5, 0)]
    hi = 2 ** 31 - 1 if key_dtype == np.int32 else 2 ** 50
    for step in range(5):
        n = [1, 63, 5000, 40001, 2049][step]
        keys = rng.integers(-hi, hi, size=n).astype(key_dtype)
        keys[rng.random(n) < 0.4] = 12345 + step                                                     # a hot key
        if n > 100:
            keys[: n // 3] = keys[n // 3: 2 * (n // 3)]                                               # repeats
        rows = ki.lookup(T(keys, dev), insert=True, tables=tables)
        ref = om.find_or_insert(keys.astype(np.int64), True)
        assert np.array_equal(rows.cpu().numpy(), ref), step
        assert len(ki) == om.size()
    live = om.size()
    ok, ov = om.export()
    got = vals[:live].cpu().numpy()
    assert np.array_equal(got, ov)
    assert float(slot[:live].min()) == 0.5 == float(slot[:live].max()) and float(slot[live:].min()) == 7.0
    # probe only: misses stay -1 and nothing is inserted
    probe = np.concatenate([ok[:100].astype(key_dtype), np.array([hi - 7, hi - 8], key_dtype)])
    r = ki.lookup(T(probe, dev), insert=False).cpu().numpy()
    assert np.array_equal(r[:100], np.arange(100)) and (r[100:] == -1).all() and len(ki) == live


def test_put_last_duplicate_wins_and_get_without_insert(dev, oracle):
    from mindrec_amd.experimental import MapParameter
    rng = np.random.default_rng(4)
    D = 8
    m = MapParameter(key_dtype=torch.int64, value_shape=(D,), capacity=4096, device=dev, seed=5)
    om = oracle.Map(D, 4096, seed=5, sigma=0.01)
    keys = rng.integers(0, 300, size=2000).astype(np.int64)          # heavy duplication: ~7 copies per key
    vals = rng.standard_normal((2000, D)).astype(np.float32)
    m.put(T(keys, dev), T(vals, dev)); om.put(keys, vals)
    uk = np.unique(keys)
    got = m.get(T(uk, dev), insert_default_value=False).cpu().numpy()
    assert np.array_equal(got, om.get(uk, False))
    last = {int(k): i for i, k in enumerate(keys)}
    assert np.array_equal(got, vals[[last[int(k)] for k in uk]])
    # second put over the same rows: the scratch word per row was handed back clean
    vals2 = rng.standard_normal((2000, D)).astype(np.float32)
    m.put(T(keys[::-1].copy(), dev), T(vals2, dev)); om.put(keys[::-1].copy(), vals2)
    assert np.array_equal(m.get(T(uk, dev), False).cpu().numpy(), om.get(uk, False))
    # get(insert=False) of unseen keys: default rows, table unchanged, no host round trip in between
    n0 = len(m)
    unseen = np.array([10 ** 12 + 3, 10 ** 12 + 4, int(uk[0])], np.int64)
    g = m.get(T(unseen, dev), insert_default_value=False).cpu().numpy()
    assert np.array_equal(g[:2], oracle.normal_rows(5, unseen[:2], D, 0.01)) and np.array_equal(g[2], om.get(uk[:1], False)[0]) and len(m) == n0


def test_admission_counters_and_device_eviction(dev):
    """hits / last-seen step are kept by the lookup kernels (one hit per key and training lookup, however often the key
    repeats in the batch); evict() runs on the device.  Model: a host dictionary."""
    from mindrec_amd.experimental import MapParameter
    rng = np.random.default_rng(6)
    m = MapParameter(key_dtype=torch.int64, value_shape=(4,), capacity=2048, device=dev, permit_filter_value=3,
                     evict_filter_value=2)
    m.add_slot("moment1", 0.0)
    hits, last, step = {}, {}, 0
    pools = [rng.integers(0, 10 ** 9, size=300), rng.integers(10 ** 10, 10 ** 11, size=300)]
    for phase in (0, 0, 0, 1, 1, 1, 1, 0, 0):
        step += 1
        keys = rng.choice(pools[phase], size=700).astype(np.int64)
        _, _, rows = m.lookup_rows(T(keys, dev), insert=True)
        rows = rows.cpu().numpy()
        for k in np.unique(keys):
            k = int(k)
            hits[k] = hits.get(k, 0) + 1
            last[k] = step
        h, ls = m.hits.cpu().numpy(), m.last_step.cpu().numpy()
        assert all(h[r] == hits[int(k)] and ls[r] == step for k, r in zip(keys, rows))
        adm = m.admitted_rows(T(rows, dev)).cpu().numpy()
        want = np.where(np.array([hits[int(k)] >= 3 for k in keys]), rows, -1)
        assert np.array_equal(adm, want)
        if step in (7,):
            n_ev = m.evict()
            dead = [k for k, s0 in last.items() if step - s0 > 2]
            assert n_ev == len(dead) and len(dead) > 0
            for k in dead:
                del hits[k], last[k]
            assert len(m) == len(last)
            gone = m.index.lookup(T(np.array(dead[:50], np.int64), dev), insert=False).cpu().numpy()
            assert (gone == -1).all()
    assert len(m) == len(last)


def test_incremental_export_import_roundtrip(dev):
    """export_data(incremental=True) returns only what changed since the previous incremental export: modified rows with
    their values (status 1) and erased keys (status 2); a second map that imports the increments ends equal to the first."""
    from mindrec_amd.experimental import MapParameter
    rng = np.random.default_rng(8)
    D = 6
    a = MapParameter(key_dtype=torch.int64, value_shape=(D,), capacity=4096, device=dev, seed=1)
    b = MapParameter(key_dtype=torch.int64, value_shape=(D,), capacity=4096, device=dev, seed=2)     # other defaults

    def state(m):
        k, v = m.get_data()
        o = np.argsort(k.cpu().numpy())
        return k.cpu().numpy()[o], v.cpu().numpy()[o]

    k1 = rng.choice(10 ** 6, size=500, replace=False).astype(np.int64)
    a.put(T(k1, dev), T(rng.standard_normal((500, D)).astype(np.float32), dev))
    inc = a.export_data(incremental=True)
    assert inc[0].numel() == 500 and bool((inc[2] == 1).all())
    b.import_data(inc)
    assert all(np.array_equal(x, y) for x, y in zip(state(a), state(b)))
    assert a.export_data(incremental=True)[0].numel() == 0                       # nothing changed since
    # change 40 rows, add 30 keys, erase 25 (5 of them come back afterwards)
    a.put(T(k1[:40], dev), T(rng.standard_normal((40, D)).astype(np.float32), dev))
    k2 = (10 ** 7 + np.arange(30)).astype(np.int64)
    a.get(T(k2, dev))                                                           # inserted with default rows
    a.erase(T(k1[100:125], dev))
    a.put(T(k1[100:105], dev), T(np.ones((5, D), np.float32), dev))
    ik, iv, ist = a.export_data(incremental=True)
    ik, ist = ik.cpu().numpy(), ist.cpu().numpy()
    assert set(ik[ist == 1]) == set(k1[:40]) | set(k2) | set(k1[100:105])
    assert set(ik[ist == 2]) == set(k1[105:125])
    assert float(iv[torch.from_numpy(ist == 2).to(dev)].abs().max()) == 0.0
    b.import_data((T(ik, dev), iv, T(ist, dev)))
    assert all(np.array_equal(x, y) for x, y in zip(state(a), state(b)))
    # the full export still lists everything with status 0
    fk, fv, fs = a.export_data()
    assert fk.numel() == len(a) == 500 + 30 - 20 and int(fs.abs().sum()) == 0
