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
    counter-based generator as a fully resident table (keyed by the global id), otherwise it is copied
    from the host;
  * gather / sparse apply then run unchanged on cache rows (the plan's `uniq` holds cache rows).

The tier is transparent: a table driven through it ends bit-identical to a fully device-resident table
(tests/test_feature_cache_gpu.py).  It trades host round trips (two small syncs per step for the miss
count and the victim list) for capacity; the all-resident engine needs none of this at V = 200 M.
"""
import torch

from . import ops


class HostBackedTable:
    def __init__(self, vocab_size, emb_dim, cache_rows, device, seed=1000, sigma=0.01, state_slots=2,
                 state_init=(0.0, 0.0)):
        if cache_rows <= 0 or vocab_size <= 0:
            raise ValueError("vocab_size and cache_rows must be positive")
        self.V, self.D, self.C = int(vocab_size), int(emb_dim), int(cache_rows)
        self.W = self.D * (1 + state_slots)
        self.device = torch.device(device)
        self.seed, self.sigma, self.state_init = seed, sigma, tuple(state_init)
        self.host = torch.zeros((self.V, self.W), dtype=torch.float32).pin_memory()
        self.materialised = torch.zeros(self.V, dtype=torch.bool)             # host row holds real data
        self.cache = torch.zeros((self.C, self.W), dtype=torch.float32, device=self.device)
        self.index = ops.KeyIndex(self.C, self.device)
        self.row_key = torch.full((self.C,), -1, dtype=torch.int64, device=self.device)
        self.stamp = torch.zeros(self.C, dtype=torch.int64, device=self.device)   # last step a row was used
        self.step = 0
        self.stats = {"hits": 0, "misses": 0, "evictions": 0, "first_touch": 0}
        # views of the cache in the layout the kernels expect
        D = self.D
        self.p = self.cache[:, :D]
        self.slots = [self.cache[:, (i + 1) * D:(i + 2) * D] for i in range(state_slots)]

    # ------------------------------------------------------------------------------------------
    def prepare(self, ids):
        """Makes all ids resident; returns a SparsePlan whose groups map to CACHE rows (plan.uniq_buf) and
        the per-position cache rows (int32 [n]) for the gather."""
        self.step += 1
        plan = ops.sparse_plan(ids)
        U = plan.U                                                    # host sync #1
        if U > self.C:
            raise RuntimeError(f"batch has {U} unique ids but the device cache holds {self.C} rows")
        keys = ops.widen_keys(plan.uniq_buf)[:U].contiguous()
        rows, _ = self.index.find_or_insert(keys, insert=False)
        hit = rows >= 0
        self.stamp[rows[hit].long()] = self.step                      # rows of this batch are not eviction candidates
        n_miss = int((~hit).sum())                                    # host sync #2
        self.stats["hits"] += U - n_miss
        self.stats["misses"] += n_miss
        if n_miss:
            free = self.C - len(self.index)
            if n_miss > free:
                self._evict(n_miss - free)
            rows, is_new = self.index.find_or_insert(keys, insert=True)
            new = is_new.bool()
            nrows, nkeys = rows[new], keys[new]
            self.row_key[nrows.long()] = nkeys
            self.stamp[nrows.long()] = self.step
            seen = self.materialised[nkeys.cpu()]
            # rows with a home on the host: copy them in; first-touch rows: initialise on the device
            if bool(seen.any()):
                sel = seen.to(self.device)
                back = self.host[nkeys[sel].cpu()].to(self.device, non_blocking=True)
                ops.scatter_rows_(self.cache, nrows[sel].contiguous(), back)
            fresh = ~seen
            if bool(fresh.any()):
                sel = fresh.to(self.device)
                fr, fk = nrows[sel].contiguous(), nkeys[sel].contiguous()
                self.stats["first_touch"] += int(fr.numel())
                ops.init_rows_(self.p, fr, fk, None, seed=self.seed, sigma=self.sigma)
                for slot, init in zip(self.slots, self.state_init):
                    ops.init_rows_(slot, fr, fk, None, seed=0, sigma=None, fill=float(init))
        if plan.n > U:      # groups >= U do not exist; keep the buffer's length (n) with skipped rows
            rows = torch.cat([rows, torch.full((plan.n - U,), -1, dtype=torch.int32, device=self.device)])
        plan.uniq_buf = rows
        rows_pos = ops.compose_i32(plan.uniq_buf, plan.inv)
        return plan, rows_pos

    def _evict(self, k):
        """Writes the k least-recently-used rows (never rows stamped this step) back to the host and frees them."""
        live = self.row_key >= 0
        cand = live & (self.stamp < self.step)
        if int(cand.sum()) < k:
            raise RuntimeError("device cache too small for this batch's working set")
        score = torch.where(cand, self.stamp, torch.full_like(self.stamp, torch.iinfo(torch.int64).max))
        victims = torch.topk(score, k, largest=False).indices                       # cache rows
        vkeys = self.row_key[victims]
        data = self.cache[victims].cpu()
        hk = vkeys.cpu()
        self.host[hk] = data
        self.materialised[hk] = True
        self.index.erase(vkeys.contiguous())
        self.row_key[victims] = -1
        self.stats["evictions"] += int(k)

    # ------------------------------------------------------------------------------------------
    def gather(self, rows_pos, row_scale=None, out_dtype=torch.float32):
        return ops.gather_rows(self.p, rows_pos, row_scale, out_dtype=out_dtype)

    def flush(self):
        """Writes every resident row back to the host (checkpoint / end of training)."""
        live = (self.row_key >= 0).nonzero().view(-1)
        if live.numel():
            hk = self.row_key[live].cpu()
            self.host[hk] = self.cache[live].cpu()
            self.materialised[hk] = True

    def full_table(self):
        """The whole table as a host tensor [V, W] (never-touched rows are generated on demand)."""
        self.flush()
        out = self.host.clone()
        missing = (~self.materialised).nonzero().view(-1)
        if missing.numel():
            tmp = torch.zeros((missing.numel(), self.W), dtype=torch.float32, device=self.device)
            seq = torch.arange(missing.numel(), dtype=torch.int32, device=self.device)
            ops.init_rows_(tmp[:, : self.D], seq, missing.to(self.device), None, seed=self.seed, sigma=self.sigma)
            for i, init in enumerate(self.state_init):
                tmp[:, (i + 1) * self.D:(i + 2) * self.D] = float(init)
            out[missing] = tmp.cpu()
        return out
