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

All bookkeeping (residency flags, LRU stamps, victim selection, the miss list) lives on the device, and rows move
between the pinned host array and the cache by the ordinary gather / scatter kernels addressing host memory over
PCIe -- no host-side indexing, no staging buffers.  The host takes part twice per step: it reads the number of
unique ids and the number of misses (two scalar syncs).

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
        self.resident = 0                                                            # keys in the index (host copy)
        self._hits = self._misses = self._evictions = 0
        self._first_touch = torch.zeros(1, dtype=torch.int64, device=dev)
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

    @property
    def stats(self):
        return {"hits": self._hits, "misses": self._misses, "evictions": self._evictions,
                "first_touch": int(self._first_touch.item())}

    @stats.setter
    def stats(self, d):
        self._hits, self._misses, self._evictions = int(d.get("hits", 0)), int(d.get("misses", 0)), int(d.get("evictions", 0))
        self._first_touch.fill_(int(d.get("first_touch", 0)))

    # ------------------------------------------------------------------------------------------
    def prepare(self, ids, skip_negative=False):
        """Makes all ids resident; returns a SparsePlan whose groups map to CACHE rows (plan.uniq_buf) and
        the per-position cache rows (int32 [n]) for the gather.  skip_negative: ids of -1 are padding slots of a shard's
        request message (no group, cache row -1)."""
        self.step += 1
        C, dev = self.C, self.device
        if self.hashed:
            # keys -> host rows (every position probes `home`; new keys take the next host row): from here on `ids` are
            # host-row numbers and the tier is the dense-table one
            ids = self.home.lookup(ids, insert=True, skip_pad=skip_negative).view(ids.shape)
        plan = ops.sparse_plan(ids, skip_negative=skip_negative)
        U = plan.U                                                    # host sync #1
        if U > C:
            raise RuntimeError(f"batch has {U} unique ids but the device cache holds {C} rows")
        keys = ops.widen_keys(plan.uniq_buf)[:U].contiguous()
        rows, _ = self.index.find_or_insert(keys, insert=False)
        hit = rows >= 0
        slot = torch.where(hit, rows, torch.full_like(rows, C)).long()
        self.stamp.scatter_(0, slot, torch.full_like(slot, self.step))   # rows of this batch are not eviction candidates
        n_miss = U - int(hit.sum())                                   # host sync #2
        self._hits += U - n_miss
        self._misses += n_miss
        if n_miss:
            free = C - self.resident
            if n_miss > free:
                self._evict(n_miss - free, n_hits=U - n_miss)
            rows, is_new = self.index.find_or_insert(keys, insert=True)
            self.resident += n_miss
            new = is_new.view(-1)[:U].bool()
            # the miss list, compacted on the device (its length is known on the host, so no sync)
            mpos = torch.argsort(new.to(torch.int8), descending=True, stable=True)[:n_miss]
            mkeys, mrows = keys[mpos].contiguous(), rows[mpos].contiguous()
            self.row_key[mrows.long()] = mkeys
            self.stamp[mrows.long()] = self.step
            seen = self.materialised[mkeys]
            # rows with a home on the host: fetched over PCIe by the gather kernel; first-touch rows: initialised here
            fetched = ops.gather_rows_pinned(self.host, torch.where(seen, mkeys, torch.full_like(mkeys, -1)))
            ops.scatter_rows_(self.cache, torch.where(seen, mrows, torch.full_like(mrows, -1)), fetched)
            fresh = (~seen).to(torch.uint8)
            self._first_touch += fresh.sum()
            self._init_groups(self.cols, mrows, mkeys, fresh)
        if plan.n > U:      # groups >= U do not exist; keep the buffer's length (n) with skipped rows
            rows = torch.cat([rows, torch.full((plan.n - U,), -1, dtype=torch.int32, device=dev)])
        plan.uniq_buf = rows
        rows_pos = ops.compose_i32(plan.uniq_buf, plan.inv)
        return plan, rows_pos

    def _init_groups(self, views, rows, keys, mask):
        """Default values of first-touch rows, column group by column group (the generator is keyed by the GLOBAL id)."""
        if self.hashed:
            keys = self.home.row_keys()[keys]                   # host row -> the key it belongs to
        elif self.key_scale != 1 or self.key_offset != 0:
            keys = keys * self.key_scale + self.key_offset
        for name, _, init in self.columns:
            if init[0] == "normal":
                ops.init_rows_(views[name], rows, keys, mask, seed=int(init[1]), sigma=float(init[2]))
            else:
                ops.init_rows_(views[name], rows, keys, mask, seed=0, sigma=None, fill=float(init[1]))

    def _evict(self, k, n_hits=0):
        """Writes the k least-recently-used rows (never rows stamped this step) back to the host and frees them."""
        C = self.C
        if self.resident - n_hits < k:
            raise RuntimeError("device cache too small for this batch's working set")
        live = self.row_key[:C] >= 0
        cand = live & (self.stamp[:C] < self.step)
        score = torch.where(cand, self.stamp[:C], torch.full_like(self.stamp[:C], torch.iinfo(torch.int64).max))
        victims = torch.topk(score, k, largest=False).indices                       # cache rows
        vkeys = self.row_key[victims].contiguous()
        data = ops.gather_rows(self.cache, victims.to(torch.int32))
        ops.scatter_rows_pinned_(self.host, vkeys, data)                             # device writes host memory (PCIe)
        self.materialised[vkeys] = True
        self.index.erase(vkeys)
        self.row_key[victims] = -1
        self.resident -= k
        self._evictions += int(k)

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
