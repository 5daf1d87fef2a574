"""Criteo TSV -> (ids, wts, label) batches: the id / weight encoding that defines the engine's input
contract.

Reference: datasets/criteo_1tb/process_data.py -- `StatsDict` (:43-163) and the record writer (:203-283):
  * 13 integer columns: id = column index 0..12, weight = value / column max; missing -> weight 0   (:138-147)
  * 26 categorical columns: id from a dictionary built over the data, keeping categories seen MORE than
    CAT_COUNT_THRESHOLD = 6 times (:120-123); anything else maps to the column's OOV id 13 + column   (:149-162);
    weight 1.0
  * dictionary ids start after the 13 + 26 reserved ones, assigned column by column (:120-123)
  * output dtypes int32 / float32 / float32 (:204-206); the reference packs 1000 samples per MindRecord row,
    here batches are yielded directly.
This is host-side data plumbing (numpy), not a kernel; it exists so real Criteo data can be fed to the
engine with the reference's encoding.
"""
import collections

import numpy as np

NUM_INTEGER_COLUMNS = 13
NUM_CATEGORICAL_COLUMNS = 26
CAT_COUNT_THRESHOLD = 6


def _parse_line(line):
    parts = line.rstrip("\n").split("\t")
    if len(parts) != 1 + NUM_INTEGER_COLUMNS + NUM_CATEGORICAL_COLUMNS:
        raise ValueError(f"expected 40 tab-separated fields, got {len(parts)}")
    return parts[0], parts[1:1 + NUM_INTEGER_COLUMNS], parts[1 + NUM_INTEGER_COLUMNS:]


class StatsDict:
    """Column statistics and the category -> id dictionary (process_data.py:43-131)."""

    def __init__(self, dense_dim=NUM_INTEGER_COLUMNS, slot_dim=NUM_CATEGORICAL_COLUMNS, threshold=CAT_COUNT_THRESHOLD):
        self.dense_dim, self.slot_dim, self.threshold = dense_dim, slot_dim, threshold
        self.field_size = dense_dim + slot_dim
        self.val_max = np.zeros(dense_dim, np.float64)
        self.val_min = np.zeros(dense_dim, np.float64)
        self.cat_counts = [collections.Counter() for _ in range(slot_dim)]
        self.cat2id = None

    def update(self, lines):
        """First pass over (a chunk of) the data: min / max of the integer columns, category counts."""
        for line in lines:
            _, vals, cats = _parse_line(line)
            for i, v in enumerate(vals):
                if v != "":
                    x = float(v)
                    self.val_max[i] = max(self.val_max[i], x)
                    self.val_min[i] = min(self.val_min[i], x)
            for j, c in enumerate(cats):
                self.cat_counts[j][c] += 1

    def finalize(self):
        """Builds the dictionary: ids 0..12 dense columns, 13..38 per-column OOV, then one id per category
        seen more than `threshold` times, column by column in first-seen order."""
        self.cat2id = [dict() for _ in range(self.slot_dim)]
        nxt = self.field_size
        for j in range(self.slot_dim):
            for cat, cnt in self.cat_counts[j].items():
                if cnt > self.threshold:
                    self.cat2id[j][cat] = nxt
                    nxt += 1
        self.vocab_size = nxt
        return self

    def encode(self, lines):
        """Second pass: lines -> (ids int32 [n, 39], wts float32 [n, 39], label float32 [n, 1])."""
        if self.cat2id is None:
            raise RuntimeError("call finalize() before encode()")
        n = len(lines)
        ids = np.empty((n, self.field_size), np.int32)
        wts = np.empty((n, self.field_size), np.float32)
        label = np.empty((n, 1), np.float32)
        ids[:, : self.dense_dim] = np.arange(self.dense_dim, dtype=np.int32)
        vmax = np.where(self.val_max != 0, self.val_max, 1.0)
        for r, line in enumerate(lines):
            lab, vals, cats = _parse_line(line)
            label[r, 0] = float(lab)
            for i, v in enumerate(vals):
                wts[r, i] = 0.0 if v == "" else float(v) / vmax[i]
            for j, c in enumerate(cats):
                ids[r, self.dense_dim + j] = self.cat2id[j].get(c, self.dense_dim + j)
                wts[r, self.dense_dim + j] = 1.0
        return ids, wts, label


class CriteoDataset:
    """Iterable of (ids, wts, label) batches over TSV lines, usable by RecModel.online_train:
    `get_dataset_size()`, `reset()`, optional `to_device` hook (e.g. lambda t: torch.from_numpy(t).cuda())."""

    def __init__(self, lines, stats, batch_size, drop_remainder=True, to_device=None):
        self.lines, self.stats, self.batch_size = list(lines), stats, int(batch_size)
        self.drop_remainder, self.to_device = drop_remainder, to_device

    def get_dataset_size(self):
        n = len(self.lines)
        return n // self.batch_size if self.drop_remainder else -(-n // self.batch_size)

    def reset(self):
        pass

    def __iter__(self):
        B = self.batch_size
        for s in range(0, len(self.lines), B):
            chunk = self.lines[s:s + B]
            if len(chunk) < B and self.drop_remainder:
                return
            out = self.stats.encode(chunk)
            yield tuple(self.to_device(a) for a in out) if self.to_device else out


# ---- record files: 1000 samples per record, sharded by rank ------------------------------------------------------
# Reference: the training scripts read MindRecord files whose every row packs `line_per_sample = 1000` samples
# (process_data.py:203-283 writes them; models/wide_deep/src/datasets.py:274-325 reads them: MindDataset(columns
# feat_ids / feat_vals / label, num_shards=rank_size, shard_id=rank_id) -> batch(batch_size / 1000, drop_remainder=True) ->
# reshape to [batch, 39] / [batch, 39] / [batch, 1]).  MindRecord itself is MindSpore's container format; the same records
# are kept here in plain .npz shards (`train_XXXX.npz` / `test_XXXX.npz`, arrays feat_ids [R, 1000*39] int32, feat_vals
# [R, 1000*39] float32, label [R, 1000] float32), and the reading contract is the reference's.
LINE_PER_SAMPLE = 1000
FIELD_SIZE = NUM_INTEGER_COLUMNS + NUM_CATEGORICAL_COLUMNS


def write_records(directory, prefix, ids, wts, label, records_per_file=64, line_per_sample=LINE_PER_SAMPLE):
    """Packs samples into records of `line_per_sample` (the tail that does not fill a record is dropped, as the reference's
    writer does, process_data.py:262-283) and writes `prefix_0000.npz`, ...  Returns the number of records written."""
    import os
    ids = np.asarray(ids, np.int32); wts = np.asarray(wts, np.float32); label = np.asarray(label, np.float32).reshape(-1)
    n, F = ids.shape
    R = n // line_per_sample
    os.makedirs(directory, exist_ok=True)
    fi = 0
    for r0 in range(0, R, records_per_file):
        r1 = min(R, r0 + records_per_file)
        sl = slice(r0 * line_per_sample, r1 * line_per_sample)
        np.savez(os.path.join(directory, f"{prefix}_{fi:04d}.npz"),
                 feat_ids=ids[sl].reshape(r1 - r0, line_per_sample * F), feat_vals=wts[sl].reshape(r1 - r0, line_per_sample * F),
                 label=label[sl].reshape(r1 - r0, line_per_sample))
        fi += 1
    return R


def write_tfrecords(directory, prefix, ids, wts, label, records_per_file=64, line_per_sample=LINE_PER_SAMPLE):
    """The same records as TFRecord files of tf.train.Example rows (`feat_ids` Int64List, `feat_vals` / `label` FloatList; 1000 samples
    per row), named `<prefix>_input_part.tfrecord-0000`, ...: the format the reference's OWN reader takes besides MindRecord
    (models/wide_deep/src/datasets.py:226-271, `dataset_type: tfrecord`; it picks up every file whose name contains `train` /
    `test` and `tfrecord`).  Returns the number of records written."""
    import os
    from . import tfrecord
    ids = np.asarray(ids, np.int32); wts = np.asarray(wts, np.float32); label = np.asarray(label, np.float32).reshape(-1)
    n, F = ids.shape
    R = n // line_per_sample
    os.makedirs(directory, exist_ok=True)
    for fi, r0 in enumerate(range(0, R, records_per_file)):
        rows = ({"feat_ids": ids[r * line_per_sample:(r + 1) * line_per_sample].reshape(-1),
                 "feat_vals": wts[r * line_per_sample:(r + 1) * line_per_sample].reshape(-1),
                 "label": label[r * line_per_sample:(r + 1) * line_per_sample]} for r in range(r0, min(R, r0 + records_per_file)))
        tfrecord.write_file(os.path.join(directory, f"{prefix}_input_part.tfrecord-{fi:04d}"), rows)
    return R


class RecordDataset:
    """The reading side: records of 1000 samples from `directory/{train,test}_*.npz`, sharded over the ranks the way
    MindDataset(num_shards, shard_id) shards a MindRecord file [EXT]: every rank gets the SAME number of records,
    ceil(R / rank_size) -- rank k reads records k, k + rank_size, ... and, where R is not a multiple of rank_size, wraps around to
    the first records -- so that all ranks run the same number of steps per epoch (a row-sharded job meets in an all-to-all and
    an all-reduce every step: a rank with one batch more would wait forever).  `batch_size / line_per_sample` records per
    batch, the remainder dropped, shuffled per epoch in train mode (seeded).  Yields (ids [B, 39] int32, wts [B, 39] float32,
    label [B, 1] float32) like the reference's padding function (datasets.py:210-217); `to_device` as in CriteoDataset."""

    def __init__(self, directory, train_mode=True, batch_size=1000, line_per_sample=LINE_PER_SAMPLE, rank_size=None, rank_id=None,
                 field_size=FIELD_SIZE, seed=0, to_device=None, cached_files=4):
        import glob
        import os
        if batch_size % line_per_sample:
            raise ValueError("batch_size must be a multiple of line_per_sample (datasets.py:319)")
        self.files = sorted(glob.glob(os.path.join(directory, ("train" if train_mode else "test") + "_*.npz")))
        if not self.files:
            raise FileNotFoundError(f"no record files in {directory}")
        self.shuffle, self.seed, self.epoch = bool(train_mode), int(seed), 0
        self.B, self.lps, self.F = int(batch_size), int(line_per_sample), int(field_size)
        self.rank_size, self.rank_id = (int(rank_size), int(rank_id)) if rank_size is not None and rank_id is not None else (1, 0)
        if not 0 <= self.rank_id < self.rank_size:
            raise ValueError("rank_id must be in [0, rank_size)")
        self.to_device = to_device
        self._cached_files = max(1, int(cached_files))
        counts = []
        for f in self.files:
            with np.load(f) as z:
                counts.append(z["label"].shape[0])
        self._index = [(fi, r) for fi, c in enumerate(counts) for r in range(c)]            # global record order
        R = len(self._index)
        per_rank = -(-R // self.rank_size)                                                  # the same on every rank
        self._mine = [(self.rank_id + j * self.rank_size) % R for j in range(per_rank)]

    def get_dataset_size(self):
        return len(self._mine) // (self.B // self.lps)

    def reset(self):
        self.epoch += 1

    def _file(self, cache, fi):
        """File fi as {name: ndarray}, read ONCE (np.load's NpzFile re-reads and decompresses a member on every access); a small
        LRU, because a shuffled epoch jumps between files."""
        z = cache.pop(fi, None)
        if z is None:
            with np.load(self.files[fi]) as f:
                z = {k: f[k] for k in ("feat_ids", "feat_vals", "label")}
            while len(cache) >= self._cached_files:
                cache.pop(next(iter(cache)))
        cache[fi] = z                    # most recently used last
        return z

    def __iter__(self):
        order = list(self._mine)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(order)
        rpb = self.B // self.lps
        cache = {}
        for b in range(len(order) // rpb):
            ids = np.empty((self.B, self.F), np.int32); wts = np.empty((self.B, self.F), np.float32)
            label = np.empty((self.B, 1), np.float32)
            for j, k in enumerate(order[b * rpb:(b + 1) * rpb]):
                fi, r = self._index[k]
                z = self._file(cache, fi)
                sl = slice(j * self.lps, (j + 1) * self.lps)
                ids[sl] = z["feat_ids"][r].reshape(self.lps, self.F)
                wts[sl] = z["feat_vals"][r].reshape(self.lps, self.F)
                label[sl, 0] = z["label"][r]
            out = (ids, wts, label)
            yield tuple(self.to_device(a) for a in out) if self.to_device else out
