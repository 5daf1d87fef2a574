"""The Criteo encoder against a line-by-line restatement of the reference's map_cat2id
(datasets/criteo_1tb/process_data.py:132-163) on a synthetic TSV."""
import numpy as np
import pytest

from mindrec_amd.criteo import CAT_COUNT_THRESHOLD, CriteoDataset, StatsDict


def _tsv(n, seed=0):
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(n):
        dense = ["" if rng.random() < 0.1 else str(int(rng.integers(0, 50))) for _ in range(13)]
        cats = ["%08x" % int(rng.zipf(1.5)) if rng.random() > 0.05 else "" for _ in range(26)]
        lines.append("\t".join([str(int(rng.random() < 0.25))] + dense + cats))
    return lines


def _restate(lines, threshold=CAT_COUNT_THRESHOLD):
    """The reference's two passes written out from the raw lines, sharing nothing with StatsDict: pass 1 (stats_vals / stats_cats,
    process_data.py:60-118) takes each integer column's maximum and each categorical column's value counts; the dictionary
    (:120-131) gives ids 0..12 to the integer columns, 13..38 to the per-column out-of-vocabulary slots and then one id per
    category seen MORE than `threshold` times, column by column in order of first appearance; pass 2 (map_cat2id, :132-163)
    encodes one value at a time."""
    rows = [line.split("\t") for line in lines]
    vmax = [0.0] * 13
    counts = [dict() for _ in range(26)]              # insertion-ordered: first appearance
    for parts in rows:
        for i, v in enumerate(parts[1:14]):
            if v != "":
                vmax[i] = max(vmax[i], float(v))
        for j, c in enumerate(parts[14:]):
            counts[j][c] = counts[j].get(c, 0) + 1
    cat2id, nxt = [dict() for _ in range(26)], 39
    for j in range(26):
        for c, k in counts[j].items():
            if k > threshold:
                cat2id[j][c] = nxt
                nxt += 1
    ids, wts = [], []
    for parts in rows:
        row_i, row_w = [], []
        for i, v in enumerate(parts[1:14]):
            row_i.append(i)
            row_w.append(0.0 if v == "" else float(v) / (vmax[i] if vmax[i] != 0 else 1.0))
        for j, c in enumerate(parts[14:]):
            row_i.append(cat2id[j][c] if c in cat2id[j] else 13 + j)
            row_w.append(1.0)
        ids.append(row_i); wts.append(row_w)
    return np.array(ids, np.int32), np.array(wts, np.float32), nxt


def test_encoder_matches_restatement_and_contract():
    lines = _tsv(600)
    st = StatsDict()
    st.update(lines[:300]); st.update(lines[300:])          # chunked first pass
    st.finalize()
    ids, wts, label = st.encode(lines)
    rid, rw, rvocab = _restate(lines)
    assert st.vocab_size == rvocab
    assert ids.dtype == np.int32 and wts.dtype == np.float32 and label.dtype == np.float32      # process_data.py:204-206
    assert ids.shape == (600, 39) and label.shape == (600, 1)
    assert np.array_equal(ids, rid) and np.array_equal(wts, rw)
    assert (ids[:, :13] == np.arange(13)).all() and (wts[:, 13:] == 1.0).all() and wts[:, :13].max() <= 1.0
    # vocabulary: only categories seen more than the threshold get their own id; the rest are per-column OOV
    for j in range(26):
        for cat, cnt in st.cat_counts[j].items():
            assert (cat in st.cat2id[j]) == (cnt > CAT_COUNT_THRESHOLD)
    assert ids.max() < st.vocab_size and len({v for d in st.cat2id for v in d.values()}) == st.vocab_size - 39
    unseen = st.encode(["0\t" + "\t".join(["7"] * 13) + "\t" + "\t".join(["zzzzzzzz"] * 26)])[0]
    assert (unseen[0, 13:] == 13 + np.arange(26)).all()


def test_dataset_batches():
    lines = _tsv(250, seed=3)
    st = StatsDict(); st.update(lines); st.finalize()
    ds = CriteoDataset(lines, st, batch_size=100)
    batches = list(ds)
    assert ds.get_dataset_size() == 2 and len(batches) == 2 and batches[0][0].shape == (100, 39)
    assert len(list(CriteoDataset(lines, st, 100, drop_remainder=False))) == 3


def test_record_files_roundtrip_and_rank_sharding(tmp_path):
    """Records of 1000 samples (the reference's MindRecord row), sharded record by record over the ranks, batch = whole
    records, remainder dropped (datasets.py:274-325)."""
    from mindrec_amd.criteo import RecordDataset, write_records
    rng = np.random.default_rng(0)
    n = 10 * 1000 + 377                                   # 10 full records; the tail does not make a record
    ids = rng.integers(0, 10 ** 6, size=(n, 39)).astype(np.int32)
    wts = rng.random((n, 39)).astype(np.float32)
    label = (rng.random(n) < 0.3).astype(np.float32)
    assert write_records(str(tmp_path), "train", ids, wts, label, records_per_file=4) == 10
    assert write_records(str(tmp_path), "test", ids[:3000], wts[:3000], label[:3000]) == 3
    ev = RecordDataset(str(tmp_path), train_mode=False, batch_size=2000)
    assert ev.get_dataset_size() == 1                      # 3 records -> one batch of 2, remainder dropped
    (bi, bw, bl), = list(ev)
    assert bi.shape == (2000, 39) and bl.shape == (2000, 1) and bi.dtype == np.int32 and bw.dtype == np.float32
    assert np.array_equal(bi, ids[:2000]) and np.array_equal(bw, wts[:2000]) and np.array_equal(bl[:, 0], label[:2000])
    seen, sizes = [], []
    for rank in range(3):
        ds = RecordDataset(str(tmp_path), train_mode=True, batch_size=1000, rank_size=3, rank_id=rank, seed=5)
        sizes.append(ds.get_dataset_size())
        got = []
        for bi, bw, bl in ds:
            r = int(np.nonzero((ids[::1000][:10] == bi[0]).all(axis=1))[0][0])       # which record this batch is
            assert np.array_equal(bi, ids[r * 1000:(r + 1) * 1000]) and np.array_equal(bl[:, 0], label[r * 1000:(r + 1) * 1000])
            got.append(r)
        # rank k reads records k, k + 3, ... and wraps around to the first records: 10 records on 3 ranks = 4 each
        assert sorted(got) == sorted((rank + 3 * j) % 10 for j in range(4))
        seen += got
    assert sizes == [4, 4, 4]                              # every rank runs the same number of steps per epoch
    assert set(seen) == set(range(10))                     # every record is read; 2 of the 12 reads are the wrapped-around ones
    # batches of 2 records: 4 records per rank -> 2 batches on every rank
    assert [RecordDataset(str(tmp_path), batch_size=2000, rank_size=3, rank_id=k).get_dataset_size() for k in range(3)] == [2, 2, 2]
    a = [b[0][0, 0] for b in RecordDataset(str(tmp_path), batch_size=1000, seed=1)]
    d = RecordDataset(str(tmp_path), batch_size=1000, seed=1)
    assert [b[0][0, 0] for b in d] == a                    # same seed, same epoch: same order
    d.reset()
    assert [b[0][0, 0] for b in d] != a                    # next epoch reshuffles
    with pytest.raises(ValueError):
        RecordDataset(str(tmp_path), batch_size=1500)
