"""`mindspore.experimental.MapParameter` on MI355X.

Reference surface: built by HashEmbeddingLookup at mindspore_rec/ops/embedding.py:136-146, read by
MapTensorGet(insert_default_value=True) at :149,193,199, API by example at README.md:160-205
(`m[key] = val`, `m[key]`, `m.erase(key)`), semantics SURVEY.md Appendix A.6 [EXT].

Storage is MI355X-style: a device key -> row-number index (`ops.KeyIndex`, csrc/mrec_hash.hip) over
a dense fp32 row table [capacity, D]; optimizer slots (Adam m/v, FTRL accum/linear) are further
[capacity, D] tables with the same row numbering, allocated by the optimizer.  All lookups and
updates therefore run through the dense gather / sparse-apply kernels.
"""
import sys
import zlib

import torch

from . import ops

MAX_SIZE = sys.maxsize


def _seed_from_name(name):
    return zlib.crc32((name or "map_parameter").encode()) & 0x7FFFFFFF


class RowGrad:
    """RowTensor analogue: what SparseGatherV2 / MapTensorGet bprop hands the optimizer --
    (row indices, per-position value gradients), not yet deduplicated."""

    def __init__(self, plan, values, row_scale=None):
        self.plan = plan            # ops.SparsePlan whose uniq_buf holds TABLE ROW numbers
        self.values = values        # [n, D] fp32
        self.row_scale = row_scale  # optional [n] mask fused into the apply


class MapParameter:
    """Hash-table parameter: int keys -> fp32 rows of shape value_shape.

    Extra (MI355X) arguments: `capacity` rows are reserved in HBM up front (default 1 Mi rows);
    `device`.  Keys -1 and -2 are reserved by MindRec (embedding.py:55-56); this implementation
    reserves none.
    """

    def __init__(self, key_dtype=torch.int32, value_dtype=torch.float32, value_shape=1, key_tensor=None,
                 value_tensor=None, default_value="normal", permit_filter_value=1, evict_filter_value=MAX_SIZE,
                 name=None, requires_grad=True, capacity=1 << 20, device="cuda:0", seed=None):
        if key_dtype not in (torch.int32, torch.int64):
            raise TypeError(f"For 'MapParameter', the 'key_dtype' must be int32 or int64, but got {key_dtype}.")
        if value_dtype != torch.float32:
            raise TypeError(f"For 'MapParameter', the 'value_dtype' must be float32, but got {value_dtype}.")
        if isinstance(value_shape, int):
            value_shape = (value_shape,)
        value_shape = tuple(int(x) for x in value_shape)
        if len(value_shape) != 1 or value_shape[0] <= 0:
            raise ValueError(f"For 'MapParameter', 'value_shape' must be one positive dimension, but got {value_shape}.")
        if not isinstance(permit_filter_value, int) or permit_filter_value < 1:
            raise ValueError("For 'MapParameter', 'permit_filter_value' must be a positive int.")
        if not isinstance(evict_filter_value, int) or evict_filter_value < 1:
            raise ValueError("For 'MapParameter', 'evict_filter_value' must be a positive int.")
        self.key_dtype, self.value_dtype, self.value_shape = key_dtype, value_dtype, value_shape
        self.default_value = default_value
        self.permit_filter_value, self.evict_filter_value = permit_filter_value, evict_filter_value
        self.name = name or "map_parameter"
        self.requires_grad = requires_grad
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.seed = _seed_from_name(self.name) if seed is None else int(seed)
        if isinstance(default_value, str):
            if default_value not in ("normal", "zeros", "ones"):
                raise ValueError(f"For 'MapParameter', unsupported 'default_value' {default_value!r}.")
            self._sigma, self._fill = (0.01, None) if default_value == "normal" else (None, 0.0 if default_value == "zeros" else 1.0)
        else:
            self._sigma, self._fill = None, float(default_value)
        D = value_shape[0]
        self.index = ops.KeyIndex(self.capacity, self.device)
        self.values = torch.zeros((self.capacity, D), dtype=torch.float32, device=self.device)
        self.slots = {}              # optimizer state tables keyed by name, same row numbering
        self.sparse_grads = []       # RowGrad list filled by the lookup's backward
        self.unique = True           # set by HashEmbeddingLookup (embedding.py:146)
        self.cache_enable = False
        # admission / eviction counters (hits, last-seen step) live inside the index and are kept by the lookup kernels
        self._track = permit_filter_value > 1 or evict_filter_value < MAX_SIZE
        self.step = 0
        self._winner = None          # put(): one int32 per row, all -1 between calls
        if key_tensor is not None:
            self.put(key_tensor, value_tensor)

    # ---- helpers --------------------------------------------------------------------------
    def _keys(self, keys):
        if not torch.is_tensor(keys):
            keys = torch.as_tensor(keys, dtype=self.key_dtype)
        if keys.dtype != self.key_dtype:
            raise TypeError(f"For 'MapParameter', the key dtype must be {self.key_dtype}, but got {keys.dtype}.")
        return keys.to(self.device).reshape(-1).contiguous()

    def _init_kwargs(self):
        return dict(seed=self.seed, sigma=self._sigma if self._sigma is not None else 0.0, fill=self._fill)

    def _tables(self):
        """(tensor, sigma, fill, seed) of every table that shares this map's row numbering: the values, then the slots."""
        tabs = [(self.values, self._sigma, self._fill, self.seed)]
        tabs += [(t["table"], None, t["init"], 0) for t in self.slots.values()]
        return tabs

    def lookup_rows(self, keys_flat, insert=True, dedup=None, train=None, skip_pad=False):
        """(dedup, rows_uniq int32, rows_pos int32 [n]) for flat device keys, without host sync.  One chain of three launches
        (mrec_map_lookup): probe, rank + place the missing keys in first-occurrence order, default rows of the values and of
        every optimizer slot.  `dedup`: an ops.unique(keys_flat) result when the caller needs the Unique anyway (a training
        step: the optimizer's inverted index is built from it) -- the index is then probed per unique key and rows_pos is
        composed through the inverse; without it every position probes the index itself.
        train (default: insert and a filter is active): this lookup counts as one training step of the table."""
        train = (self._track and insert) if train is None else bool(train)
        if train:
            self.step += 1            # one inserting lookup = one training step of this table (evict threshold unit)
        kw = dict(insert=insert, train=train, step=self.step, permit=self.permit_filter_value, tables=self._tables())
        if dedup is not None:
            rows_u = self.index.lookup(dedup.uniq_buf, unique=True, n_dev=dedup.n_uniq_dev, **kw)
            return dedup, rows_u, ops.compose_i32(rows_u, dedup.inv)
        rows_pos = self.index.lookup(keys_flat, skip_pad=skip_pad, **kw)      # (skip_pad: key -1 = a padding slot of a shard's request)
        return None, None, rows_pos

    def admitted_rows(self, rows):
        """Row numbers with un-admitted keys (seen in fewer than permit_filter_value training lookups) replaced by -1,
        which the sparse-apply kernels skip: such keys read their default row and are not updated."""
        if self.permit_filter_value <= 1:
            return rows
        ok = (rows >= 0) & (rows < self.capacity) & (self.hits[rows.clamp(0, self.capacity - 1).long()] >= self.permit_filter_value)
        return torch.where(ok, rows, torch.full_like(rows, -1))

    @property
    def hits(self):
        return self.index.tracking()[0]

    @property
    def last_step(self):
        return self.index.tracking()[1]

    # ---- MapTensorGet / Put / Erase -----------------------------------------------------------
    def get(self, key_tensor, insert_default_value=True):
        """MapTensorGet: 4 launches (the lookup chain + the row gather), 3 without insertion (probe, gather, defaults)."""
        keys = self._keys(key_tensor)
        D = self.value_shape[0]
        if insert_default_value and D % 4 == 0 and D <= 256:
            # one pass over the rows of NEW keys less: the kernel that generates their default rows writes them to the table AND to
            # the output; the gather behind it moves the rows of the keys that were there (and of later positions of new keys)
            out = torch.empty((keys.numel(), D), dtype=torch.float32, device=self.device)
            train = self._track
            if train:
                self.step += 1
            _, rows_g = self.index.lookup(keys, insert=True, train=train, step=self.step, permit=self.permit_filter_value,
                                          tables=self._tables(), out=out, out_table=0)
            ops.gather_rows_skip_(self.values, rows_g, out)
            return out
        _, _, rows_pos = self.lookup_rows(keys, insert=insert_default_value)
        out = ops.gather_rows(self.values, rows_pos)
        if not insert_default_value:
            # missing keys read as their default row, without being inserted (no host round trip: a kernel overlays them)
            self.index.fill_missing(keys, rows_pos, out, self._sigma, self._fill, self.seed)
        return out

    def put(self, key_tensor, value_tensor):
        """MapTensorPut (README.md:188-190): upsert; with duplicate keys in one call the LAST one wins, as in a sequential
        loop over the pairs."""
        keys = self._keys(key_tensor)
        vals = value_tensor.to(self.device, torch.float32).reshape(keys.numel(), self.value_shape[0])
        _, _, rows_pos = self.lookup_rows(keys, insert=True, train=False)
        if self._winner is None:
            self._winner = torch.full((self.capacity,), -1, dtype=torch.int32, device=self.device)
        ops.put_rows_last_(self.values, rows_pos, vals, self._winner)
        self.index.mark_dirty(rows_pos)                     # for the next incremental export
        return self

    def erase(self, key_tensor):
        keys = self._keys(key_tensor)
        d = ops.unique(keys)
        self.index.erase(ops.widen_keys(d.uniq))
        return self

    def __getitem__(self, key_tensor):
        return self.get(key_tensor, True)

    def __setitem__(self, key_tensor, value_tensor):
        self.put(key_tensor, value_tensor)

    def __len__(self):
        return len(self.index)

    # ---- export / import ------------------------------------------------------------------
    def get_keys(self):
        k, _ = self.index.export()
        return k.to(self.key_dtype)

    def get_values(self):
        _, r = self.index.export()
        return ops.gather_rows(self.values, r)

    def get_data(self):
        k, r = self.index.export()
        return k.to(self.key_dtype), ops.gather_rows(self.values, r)

    def export_data(self, incremental=False):
        """(keys, values, statuses).  incremental=False: every live pair, status 0.  incremental=True (RELEASE.md:18): only
        what changed since the previous incremental export -- rows inserted, trained on or put since then (status 1, with
        their values) and keys erased or evicted since then (status 2, values zero); the marks are cleared."""
        if not incremental:
            k, v = self.get_data()
            return k, v, torch.zeros(k.numel(), dtype=torch.int32, device=self.device)
        k, r, status = self.index.export_dirty(clear=True)
        v = ops.gather_rows(self.values, r)               # negative rows (erased keys) read as zeros
        return k.to(self.key_dtype), v, status

    def import_data(self, data):
        """Full or incremental: pairs with status 2 are erased, the others upserted."""
        keys, values = data[0], data[1]
        status = data[2] if len(data) > 2 and data[2] is not None else None
        if status is not None and bool((status == 2).any()):
            gone = status.to(self.device) == 2
            keys, values = keys.to(self.device), values.to(self.device)
            self.erase(keys[gone])
            keys, values = keys[~gone], values[~gone]
        if keys.numel():
            self.put(keys, values)

    def add_slot(self, name, init=0.0):
        """Optimizer state table with this map's row numbering (Adam m/v, FTRL accum/linear)."""
        if name not in self.slots:
            t = torch.full((self.capacity, self.value_shape[0]), float(init), dtype=torch.float32, device=self.device)
            self.slots[name] = {"table": t, "init": float(init)}
            pend = self.__dict__.setdefault("_pending_slots", {}).pop(name, None)
            if pend is not None:                   # restored before the optimizer had created the slot (see import_slot)
                ops.scatter_rows_(t, pend[0], pend[1])
        return self.slots[name]["table"]

    def import_slot(self, name, rows, vals):
        """Restores rows of an optimizer slot.  A slot the optimizer has not created yet is NOT created here -- only the optimizer
        knows what an untouched row holds (FTRL's accumulator starts at initial_accum, not 0) -- its rows wait until add_slot."""
        vals = vals.to(self.device, torch.float32).reshape(rows.numel(), self.value_shape[0])
        if name in self.slots:
            ops.scatter_rows_(self.slots[name]["table"], rows, vals)
        else:
            self.__dict__.setdefault("_pending_slots", {})[name] = (rows.clone(), vals.clone())

    def slot_rows(self, rows):
        """{slot name: its values at `rows`}, restored-but-not-yet-created slots included."""
        out = {n: ops.gather_rows(t["table"], rows) for n, t in self.slots.items()}
        for n, (r, v) in self.__dict__.get("_pending_slots", {}).items():
            t = torch.zeros((self.capacity, self.value_shape[0]), dtype=torch.float32, device=self.device)
            ops.scatter_rows_(t, r, v)
            out[n] = ops.gather_rows(t, rows)
        return out

    # ---- eviction (README.md:182-183: thresholds in training steps) -----------------------------
    def evict(self):
        """Removes keys not seen in a training lookup for more than evict_filter_value steps (SURVEY A.6 definition), on the
        device (mrec_map_evict: one pass over the rows); returns how many went (one host read of the count)."""
        if not self._track or self.evict_filter_value >= MAX_SIZE:
            return 0
        return int(self.index.evict(self.step, self.evict_filter_value).item())
