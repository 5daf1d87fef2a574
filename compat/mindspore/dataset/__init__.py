"""`mindspore.dataset`: the one source the in-scope online-learning path uses -- `GeneratorDataset(source, column_names)`
over a random-access or iterable Python object, then `.batch(B)`
(examples/online_learning/online_train.py:30-46,71-72; ci/st/online_learning/test_online_learning.py:43-51,63-64).
`Schema` + `TFRecordDataset` read the TFRecord files the reference's training scripts take (models/wide_deep/src/datasets.py:226-271)
through the codec of `mindrec_amd/tfrecord.py`.  MindRecord is MindSpore's own container and cannot be restated: `MindDataset`
says so (the same records live in `.npz` shards here, `mindrec_amd.criteo.RecordDataset`)."""
import numpy as np

from . import config  # noqa: F401


class Dataset:
    """Iteration protocol the Model loops use: iterable of tuples of numpy arrays (one per column), `get_dataset_size()`,
    `reset()`."""

    def __init__(self, column_names):
        self.column_names = list(column_names) if column_names is not None else None

    def get_dataset_size(self):
        raise NotImplementedError

    def get_col_names(self):
        return list(self.column_names or [])

    def batch(self, batch_size, drop_remainder=False, num_parallel_workers=None, **kw):
        return BatchDataset(self, batch_size, drop_remainder)

    def repeat(self, count=None):
        return RepeatDataset(self, count)

    def map(self, operations=None, input_columns=None, output_columns=None, column_order=None, num_parallel_workers=None, **kw):
        return MapDataset(self, operations, input_columns, output_columns, column_order)

    def shuffle(self, buffer_size):
        return self

    def create_tuple_iterator(self, columns=None, num_epochs=-1, output_numpy=False, do_copy=True):
        return iter(self)

    def create_dict_iterator(self, num_epochs=-1, output_numpy=False, do_copy=True):
        names = self.get_col_names()
        return (dict(zip(names, row)) for row in self)

    def reset(self):
        return None

    def get_batch_size(self):
        return 1

    def get_repeat_count(self):
        return 1


class GeneratorDataset(Dataset):
    """GeneratorDataset(source, column_names=None, ..., shuffle=None, num_shards=None, shard_id=None).  `source`:
    an object with `__getitem__` + `__len__` (rows are read in index order -- a stream source ignores the index and hands
    out its next row, online_train.py:35-43), an iterable, or a callable returning an iterator.  A row is a tuple with one
    entry per column (a lone array counts as one column)."""

    def __init__(self, source, column_names=None, column_types=None, schema=None, num_samples=None, num_parallel_workers=1,
                 shuffle=None, sampler=None, num_shards=None, shard_id=None, python_multiprocessing=True, max_rowsize=6):
        super().__init__(column_names)
        if column_names is None and schema is None:
            raise ValueError("For 'GeneratorDataset', neither 'column_names' nor 'schema' are provided.")
        if isinstance(column_names, str):
            self.column_names = [c.strip() for c in column_names.split(",")]
        self.source, self.num_samples = source, num_samples
        self.num_shards, self.shard_id = (num_shards or 1), (shard_id or 0)
        if not 0 <= self.shard_id < self.num_shards:
            raise ValueError(f"For 'GeneratorDataset', 'shard_id' must be in [0, num_shards), but got {shard_id} of {num_shards}.")
        self._random_access = hasattr(source, "__getitem__") and hasattr(source, "__len__")
        if not self._random_access and not (callable(source) or hasattr(source, "__iter__")):
            raise TypeError("For 'GeneratorDataset', the 'source' must be a random-access object (__getitem__, __len__), an "
                            "iterable or a callable that returns an iterator.")

    def get_dataset_size(self):
        if self._random_access:
            n = len(self.source)
        elif hasattr(self.source, "__len__"):
            n = len(self.source)
        else:
            n = self.num_samples if self.num_samples is not None else sum(1 for _ in self._rows())
        n = (n + self.num_shards - 1) // self.num_shards if self.num_shards > 1 else n
        return min(n, self.num_samples) if self.num_samples is not None else n

    def _rows(self):
        if self._random_access:
            return (self.source[i] for i in range(self.shard_id, len(self.source), self.num_shards))
        it = self.source() if callable(self.source) else iter(self.source)
        return (r for i, r in enumerate(it) if i % self.num_shards == self.shard_id)

    def __iter__(self):
        ncol = len(self.column_names or [])
        for i, row in enumerate(self._rows()):
            if self.num_samples is not None and i >= self.num_samples:
                return
            if not isinstance(row, (tuple, list)):
                row = (row,)
            if ncol and len(row) != ncol:
                raise RuntimeError(f"GeneratorDataset: a row of the source has {len(row)} column(s) but 'column_names' "
                                   f"names {ncol}: {self.column_names}.")
            yield tuple(np.asarray(c) for c in row)


class BatchDataset(Dataset):
    """Stacks `batch_size` consecutive rows column by column; an unbounded stream simply keeps yielding batches."""

    def __init__(self, parent, batch_size, drop_remainder=False):
        super().__init__(parent.column_names)
        if isinstance(batch_size, bool) or not isinstance(batch_size, int) or batch_size <= 0:
            raise ValueError(f"For 'batch', the 'batch_size' must be int and must > 0, but got {batch_size!r}.")
        self.parent, self.batch_size, self.drop_remainder = parent, batch_size, bool(drop_remainder)

    def get_dataset_size(self):
        n = self.parent.get_dataset_size()
        return n // self.batch_size if self.drop_remainder else -(-n // self.batch_size)

    def get_batch_size(self):
        return self.batch_size

    def __iter__(self):
        cols = None
        for row in self.parent:
            if cols is None:
                cols = [[] for _ in row]
            for c, x in zip(cols, row):
                c.append(x)
            if len(cols[0]) == self.batch_size:
                yield tuple(np.stack(c) for c in cols)
                cols = None
        if cols is not None and not self.drop_remainder:
            yield tuple(np.stack(c) for c in cols)

    def reset(self):
        self.parent.reset()


class RepeatDataset(Dataset):
    def __init__(self, parent, count):
        super().__init__(parent.column_names)
        self.parent, self.count = parent, count

    def get_dataset_size(self):
        n = self.parent.get_dataset_size()
        return n * self.count if self.count and self.count > 0 else n

    def __iter__(self):
        i = 0
        while self.count is None or self.count < 0 or i < self.count:
            yield from self.parent
            self.parent.reset()
            i += 1

    def reset(self):
        self.parent.reset()


class MapDataset(Dataset):
    def __init__(self, parent, operations, input_columns, output_columns, column_order):
        super().__init__(output_columns or parent.column_names)
        self.parent = parent
        self.ops = operations if isinstance(operations, (list, tuple)) else [operations]
        names = parent.get_col_names()
        self.idx = [names.index(c) for c in (input_columns or names)]

    def get_dataset_size(self):
        return self.parent.get_dataset_size()

    def get_batch_size(self):
        return self.parent.get_batch_size()

    def reset(self):
        self.parent.reset()

    def __iter__(self):
        for row in self.parent:
            vals = [row[i] for i in self.idx]
            for op in self.ops:
                vals = op(*vals)
                vals = list(vals) if isinstance(vals, (tuple, list)) else [vals]
            if len(vals) == len(self.idx):
                row = list(row)
                for i, v in zip(self.idx, vals):
                    row[i] = v
                yield tuple(row)
            else:
                yield tuple(vals)


class NumpySlicesDataset(GeneratorDataset):
    def __init__(self, data, column_names=None, **kw):
        if isinstance(data, dict):
            column_names = column_names or list(data)
            data = tuple(data[k] for k in column_names)
        if not isinstance(data, (tuple, list)):
            data = (data,)
        arrs = [np.asarray(a) for a in data]

        class _Src:
            def __len__(self):
                return len(arrs[0])

            def __getitem__(self, i):
                return tuple(a[i] for a in arrs)

        super().__init__(_Src(), column_names or [f"column_{i}" for i in range(len(arrs))], **kw)


class Schema:
    """Schema().add_column(name, de_type, shape=None): the columns a TFRecordDataset parses, in this order."""

    def __init__(self, schema_file=None):
        if schema_file is not None:
            raise NotImplementedError("Schema files are not read; add the columns with add_column().")
        self.columns = []

    def add_column(self, name, de_type, shape=None):
        if not isinstance(name, str) or not name:
            raise ValueError("For 'Schema.add_column', the 'name' must be a non-empty string.")
        self.columns.append((name, de_type, tuple(shape) if shape is not None else None))


def _np_dtype(de_type):
    import torch
    t = getattr(de_type, "dtype", de_type)
    return {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32, torch.float16: np.float16,
            torch.float64: np.float64}.get(t, t if isinstance(t, type) else np.float32)


class TFRecordDataset(Dataset):
    """TFRecordDataset(dataset_files, schema=None, columns_list=None, num_samples=None, num_parallel_workers=None, shuffle=...,
    num_shards=None, shard_id=None, shard_equal_rows=False): rows of tf.train.Example records, one numpy array per column.
    Sharding as MindSpore states it: by FILE (file i belongs to shard i mod num_shards), or, with shard_equal_rows, by ROW with
    every shard getting the same number of rows (row i belongs to shard i mod num_shards; the rows past the last full round are
    dropped).  shuffle: a permutation of the shard's rows per epoch, seeded by `dataset.config`'s seed (+ the epoch)."""

    def __init__(self, dataset_files, schema=None, columns_list=None, num_samples=None, num_parallel_workers=None, shuffle=True,
                 num_shards=None, shard_id=None, shard_equal_rows=False, cache=None, compression_type=None):
        if isinstance(dataset_files, str):
            dataset_files = [dataset_files]
        self.files = sorted(dataset_files)
        if not self.files:
            raise ValueError("For 'TFRecordDataset', 'dataset_files' is empty.")
        if compression_type:
            raise NotImplementedError("compressed TFRecord files are not read")
        cols = list(schema.columns) if isinstance(schema, Schema) else None
        if columns_list is not None:
            cols = [c for c in (cols or [(n, None, None) for n in columns_list]) if c[0] in columns_list]
        self._cols = cols                     # None: every feature of the first record, by name
        super().__init__([c[0] for c in cols] if cols else None)
        self.num_samples = num_samples
        self.shuffle = bool(shuffle) and shuffle not in (0, "false")
        self.num_shards, self.shard_id = (num_shards or 1), (shard_id or 0)
        if not 0 <= self.shard_id < self.num_shards:
            raise ValueError(f"For 'TFRecordDataset', 'shard_id' must be in [0, num_shards), but got {shard_id} of {num_shards}.")
        self.shard_equal_rows = bool(shard_equal_rows)
        self._epoch = 0
        from mindrec_amd import tfrecord
        self._codec = tfrecord
        counts = [tfrecord.count_records(f) for f in self.files]
        index = [(fi, r) for fi, c in enumerate(counts) for r in range(c)]
        if self.num_shards > 1:
            if self.shard_equal_rows:
                per = len(index) // self.num_shards
                index = [index[i] for i in range(self.shard_id, per * self.num_shards, self.num_shards)]
            else:
                index = [(fi, r) for fi, r in index if fi % self.num_shards == self.shard_id]
        self._index = index

    def get_dataset_size(self):
        n = len(self._index)
        return min(n, self.num_samples) if self.num_samples is not None else n

    def reset(self):
        self._epoch += 1

    def __iter__(self):
        order = list(self._index)
        if self.shuffle:
            np.random.default_rng(config.get_seed() + self._epoch).shuffle(order)
        if self.num_samples is not None:
            order = order[:self.num_samples]
        by_file = {}
        for fi, r in order:
            by_file.setdefault(fi, set()).add(r)
        cache = {}
        for fi, r in order:
            if fi not in cache:               # (one pass over a file keeps the rows this epoch wants)
                want = by_file[fi]
                cache[fi] = {j: rec for j, rec in enumerate(self._codec.read_file(self.files[fi])) if j in want}
            ex = self._codec.decode_example(cache[fi].pop(r))
            if self._cols is None:
                self._cols = [(n, None, None) for n in sorted(ex)]
                self.column_names = [c[0] for c in self._cols]
            row = []
            for name, de_type, shape in self._cols:
                if name not in ex:
                    raise RuntimeError(f"TFRecordDataset: column {name!r} is not in the record (it has {sorted(ex)}).")
                a = np.asarray(ex[name])
                if de_type is not None:
                    a = a.astype(_np_dtype(de_type))
                row.append(a.reshape(shape) if shape else a)
            yield tuple(row)
            if not cache[fi]:
                del cache[fi]


class MindDataset(Dataset):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MindRecord is MindSpore's own container format and is not read here: write the records as TFRecord "
                                  "files (mindrec_amd.criteo.write_tfrecords; dataset_type 'tfrecord' in the reference's configs) or "
                                  "read this repo's .npz record shards with mindrec_amd.criteo.RecordDataset.")
