"""Host-DRAM feature-cache tier.

Reference: MindRec's "embedding cache" (`vocab_cache_size`; mindspore_rec/ops/embedding.py:164-182,
models/wide_deep/src/wide_and_deep.py:215-265, scripts/run_parameter_server_standalone_train_terabyte_scale_model.sh:35-42):
the worker keeps `vocab_cache_size` rows on the device, a host-side map id -> slot, pulls misses from
the server and pushes evicted rows back [EXT].  Here, on one MI355X node, the backing store is pinned
host DRAM instead of a parameter server:

  * every row -- weights AND optimizer state, in the fused layout [p(D) | m(D) | v(D)] -- has a home in a
    pinned host array [V, W]; a row is only materialised there when it is first evicted;
  * the device holds `cache_rows` rows [cache_rows, W] plus a device key index (ops.KeyIndex) id -> cache row;
  * prepare(ids) makes every id of the batch resident: Unique -> probe -> if the misses do not fit, the
    least-recently-used rows that are not in this batch are written back to the host and their keys erased
    -> misses get cache rows; a row never seen before is initialised on the device by the same
    counter-based generator as a fully resident table (keyed by the global id), otherwise it is fetched
    from the host;
  * gather / sparse apply then run unchanged on cache rows (the plan's `uniq` holds cache rows).

All bookkeeping (residency flags, LRU stamps, victim selection, the miss and victim lists, the counters) lives on the
device, every array has a shape the host knows in advance (lists are compacted into buffers of the batch's length and
carry their true length in a device word), and rows move between the pinned host array and the cache by a row-copy kernel
that addresses host memory over PCIe and reads its list length on the device (ops.move_rows_) -- no host-side indexing,
no staging buffers, and NO host synchronisation inside a step: prepare() never reads a value back
(tests/test_feature_cache_gpu.py runs it under torch.cuda.set_sync_debug_mode("error")).  What cannot be raised without
reading back -- a batch with more unique ids than the cache has rows, a working set that leaves too few eviction
candidates -- is latched in a device word and raised by check() (called by flush(), stats and the engines' overflow
checks), like the shard exchange's overflow counter.

Victims are the least-recently-used rows that are not in this batch: ages (steps since last use, capped at 4095) are
sorted once, the age of the n_evict-th oldest row is the threshold, rows above it go, rows at it go in row order until the
count is met -- exact LRU below the cap, reproducible.

The tier is transparent: a table driven through it ends bit-identical to a fully device-resident table
(tests/test_feature_cache_gpu.py).
"""
import torch

from . import ops


class HostBackedTable:
    """columns: optional list of (name, width, init) describing a row's column groups, init = ("normal", seed, sigma)
    or ("fill", value); default: the fused LazyAdam row [p(D) normal | m(D) 0 | v(D) 0] (state_slots / state_init).
    `self.cols[name]` is the [cache_rows, width] view of a group."""

    def __init__(self, vocab_size, emb_dim, cache_rows, device, seed=1000, sigma=0.01, state_slots=2,
                 state_init=(0.0, 0.0), columns=None, hashed=False, key_scale=1, key_offset=0):
        """hashed=True: the table is a hash table keyed by arbitrary int keys (MapParameter under the cache tier, BASELINE
        configs[4]): a second device key index `home` gives every key a host row (0 .. vocab_size-1, in order of first
        appearance), the tier below runs unchanged on host-row numbers, and default values stay keyed by the KEY.
        key_scale / key_offset: a row shard of a dense table -- local row r is global id r * key_scale + key_offset, which is
        what its default values are keyed by."""
        if cache_rows <= 0 or vocab_size <= 0:
            raise ValueError("vocab_size and cache_rows must be positive")
        self.V, self.D, self.C = int(vocab_size), int(emb_dim), int(cache_rows)
        if columns is None:
            columns = [("p", self.D, ("normal", seed, sigma))] + [(f"slot{i}", self.D, ("fill", float(state_init[i])))
                                                                   for i in range(state_slots)]
        self.columns = [(str(n), int(w), tuple(i)) for n, w, i in columns]
        self.W = sum(w for _, w, _ in self.columns)
        self.device = torch.device(device)
        self.seed, self.sigma, self.state_init = seed, sigma, tuple(state_init)
        self.host = torch.zeros((self.V, self.W), dtype=torch.float32).pin_memory()
        C, dev = self.C, self.device
        # slot V / C of the flag arrays is a dummy that absorbs the writes of masked-out entries
        self.materialised = torch.zeros(self.V + 1, dtype=torch.bool, device=dev)      # host row holds real data
        self.cache = torch.zeros((C, self.W), dtype=torch.float32, device=dev)
        self.index = ops.KeyIndex(C, dev)
        self.hashed = bool(hashed)
        self.key_scale, self.key_offset = int(key_scale), int(key_offset)
        self.home = ops.KeyIndex(self.V, dev) if self.hashed else None        # key -> host row
        self.row_key = torch.full((C + 1,), -1, dtype=torch.int64, device=dev)
        self.stamp = torch.zeros(C + 1, dtype=torch.int64, device=dev)               # last step a row was used
        self.step = 0
        self._resident = torch.zeros(1, dtype=torch.int64, device=dev)               # keys in the index
        self._counts = torch.zeros(4, dtype=torch.int64, device=dev)                 # hits, misses, evictions, first touches
        self._err = torch.zeros(1, dtype=torch.int64, device=dev)                    # latched errors (check())
        self._iota_buf = torch.arange(C, dtype=torch.int64, device=dev)
        self.cols, off = {}, 0
        for name, w, _ in self.columns:
            self.cols[name] = self.cache[:, off:off + w]
            off += w
        first = self.columns[0][0]
        self.p = self.cols[first]
        self.slots = [self.cols[n] for n, _, _ in self.columns[1:]]

    @property
    def col_ranges(self):
        """{column group: (first column, one past the last)} of a row."""
        out, off = {}, 0
        for name, w, _ in self.columns:
            out[name] = (off, off + w)
            off += w
        return out

    _AGE_CAP = 4095

    def check(self):
        """Raises what prepare() latched on the device (host sync)."""
        e = int(self._err.item())
        dropped = self.index.counters()[2]
        if e & 1:
            raise RuntimeError(f"a batch had more unique ids than the device cache holds ({self.C} rows)")
        if (e & 2) or dropped:
            raise RuntimeError("device cache too small for this batch's working set")

    @property
    def resident(self):
        """Keys in the cache (host sync)."""
        return int(self._resident.item())

    @property
    def stats(self):
        self.check()
        h, m, e, f = (int(x) for x in self._counts.tolist())
        return {"hits": h, "misses": m, "evictions": e, "first_touch": f}

    @stats.setter
    def stats(self, d):
        self._counts.copy_(torch.tensor([int(d.get(k, 0)) for k in ("hits", "misses", "evictions", "first_touch")], dtype=torch.int64))

    def _iota(self, n):
        if self._iota_buf.numel() < n:
            self._iota_buf = torch.arange(n, dtype=torch.int64, device=self.device)
        return self._iota_buf[:n]

    @staticmethod
    def _compact(mask, values, length):
        """values[mask] in order, in a buffer of `length` entries padded with -1 (entries beyond it are dropped)."""
        pos = torch.cumsum(mask, 0) - 1
        out = torch.full((length + 1,), -1, dtype=values.dtype, device=values.device)
        out.scatter_(0, torch.where(mask & (pos < length), pos, length), values)
        return out[:length]

    # ------------------------------------------------------------------------------------------
    def prepare(self, ids, skip_negative=False):
        """Makes all ids resident; returns a SparsePlan whose groups map to CACHE rows (plan.uniq_buf) and
        the per-position cache rows (int32 [n]) for the gather.  skip_negative: ids of -1 are padding slots of a shard's
        request message (no group, cache row -1).  No host synchronisation (module docstring)."""
        self.step += 1
        C, V, dev, step = self.C, self.V, self.device, self.step
        if self.hashed:
            # keys -> host rows (every position probes `home`; new keys take the next host row): from here on `ids` are
            # host-row numbers and the tier is the dense-table one
            ids = self.home.lookup(ids, insert=True, skip_pad=skip_negative).view(ids.shape)
        plan = ops.sparse_plan(ids, skip_negative=skip_negative)
        n = plan.n
        U = plan.n_uniq_dev                                           # device word
        iota = self._iota(max(n, C))
        valid = iota[:n] < U
        keys = torch.where(valid, ops.widen_keys(plan.uniq_buf)[:n], -1)
        rows, _ = self.index.find_or_insert(keys, insert=False, n_dev=U)       # entries >= U: -1
        hit = rows >= 0
        self.stamp.scatter_(0, torch.where(hit, rows.long(), C), step)           # rows of this batch are not eviction candidates
        n_hit = hit.sum()
        n_miss = U[0] - n_hit
        n_evict = torch.clamp(n_miss - (C - self._resident[0]), min=0)
        self._err |= (U[0] > C).to(torch.int64)
        # ---- victims: the n_evict least-recently-used rows that are not in this batch
        live = self.row_key[:C] >= 0
        age = torch.where(live, torch.clamp(step - self.stamp[:C], max=self._AGE_CAP), 0)   # 0: free, or used this step
        by_age = torch.sort(age.to(torch.int16), descending=True).values
        thr = by_age.index_select(0, torch.clamp(n_evict - 1, min=0, max=C - 1).view(1))[0].to(torch.int64)   # age of the n_evict-th oldest row
        self._err |= ((n_evict > 0) & (thr == 0)).to(torch.int64) << 1
        above = age > thr
        at = (age == thr) & (thr > 0)
        need_at = n_evict - above.sum()
        sel = (above | (at & (torch.cumsum(at, 0) <= need_at))) & (n_evict > 0)
        vrows = self._compact(sel, iota[:C], n)                                            # cache rows, -1 padded
        vkeys = torch.where(vrows >= 0, self.row_key[torch.clamp(vrows, min=0)], -1)
        n_evict_w = n_evict.view(1)
        ops.move_rows_(self.host, vkeys, self.cache, vrows, n_dev=n_evict_w)                # device writes host memory (PCIe)
        self.materialised.scatter_(0, torch.where(vkeys >= 0, vkeys, V), True)
        self.materialised[V:].fill_(False)          # (`t[i] = value` stages the value through a synchronising copy)
        self.index.erase(vkeys)                                                             # (key -1: not there)
        self.row_key.scatter_(0, torch.where(vrows >= 0, vrows, C), -1)
        # ---- the misses take rows (fresh ones first, then the ones just freed)
        rows, is_new = self.index.find_or_insert(keys, insert=True, n_dev=U)
        new = valid & (is_new.view(-1)[:n] != 0)
        n_new = new.sum()
        self._resident += n_new - n_evict
        self._counts[:3] += torch.stack([n_hit, n_miss, n_evict])
        mkeys = self._compact(new, keys, n)
        mrows = self._compact(new, rows.long(), n)
        mslot = torch.where(mrows >= 0, mrows, C)
        self.row_key.scatter_(0, mslot, mkeys)
        self.row_key[C:].fill_(-1)
        self.stamp.scatter_(0, mslot, step)
        seen = self.materialised[torch.where(mkeys >= 0, mkeys, V)]
        # rows with a home on the host: fetched over PCIe; first-touch rows: initialised here
        ops.move_rows_(self.cache, torch.where(seen, mrows, -1), self.host, torch.where(seen, mkeys, -1), n_dev=n_new.view(1))
        fresh = (mkeys >= 0) & ~seen
        self._counts[3] += fresh.sum()
        self._init_groups(self.cols, mrows.to(torch.int32), mkeys, fresh.to(torch.uint8))
        plan.uniq_buf = torch.where(valid, rows, -1)
        rows_pos = ops.compose_i32(plan.uniq_buf, plan.inv)
        return plan, rows_pos

    def _init_groups(self, views, rows, keys, mask):
        """Default values of first-touch rows, column group by column group (the generator is keyed by the GLOBAL id)."""
        if self.hashed:
            keys = self.home.row_keys()[torch.clamp(keys, min=0)]                   # host row -> the key it belongs to
        elif self.key_scale != 1 or self.key_offset != 0:
            keys = keys * self.key_scale + self.key_offset
        for name, _, init in self.columns:
            if init[0] == "normal":
                ops.init_rows_(views[name], rows, keys, mask, seed=int(init[1]), sigma=float(init[2]))
            else:
                ops.init_rows_(views[name], rows, keys, mask, seed=0, sigma=None, fill=float(init[1]))

    # ------------------------------------------------------------------------------------------
    def gather(self, rows_pos, row_scale=None, out_dtype=torch.float32):
        return ops.gather_rows(self.p, rows_pos, row_scale, out_dtype=out_dtype)

    def flush(self):
        """Writes every resident row back to the host (checkpoint / end of training)."""
        C = self.C
        keys = self.row_key[:C]
        ops.scatter_rows_pinned_(self.host, keys, self.cache)                        # free rows carry key -1: skipped
        self.materialised.scatter_(0, torch.where(keys >= 0, keys, torch.full_like(keys, self.V)),
                                   torch.ones(C, dtype=torch.bool, device=self.device))
        self.materialised[self.V] = False
        torch.cuda.synchronize(self.device)
        self.check()

    def export_hashed(self):
        """hashed tables: (keys int64 [n], rows float32 [n, W]) of every key seen so far (host tensors)."""
        self.flush()
        k, r = self.home.export()
        return k.cpu(), self.host[r.cpu().long()]

    def full_table(self):
        """The whole table as a host tensor [V, W] (never-touched rows are generated on demand)."""
        if self.hashed:
            raise RuntimeError("a hashed table has no dense image; use export_hashed()")
        self.flush()
        out = self.host.clone()
        missing = (~self.materialised[: self.V]).nonzero().view(-1)
        if missing.numel():
            tmp = torch.zeros((missing.numel(), self.W), dtype=torch.float32, device=self.device)
            seq = torch.arange(missing.numel(), dtype=torch.int32, device=self.device)
            views, off = {}, 0
            for name, w, _ in self.columns:
                views[name] = tmp[:, off:off + w]
                off += w
            self._init_groups(views, seq, missing, None)
            out[missing.cpu()] = tmp.cpu()
        return out
