"""GRAPH_MODE's compile step on an MI355X: a recognised train step written against the `mindspore` API (compat/mindspore) is
lowered to the fused engine -- the whole step as ONE HIP graph over hand-written kernels -- instead of being executed primitive
by primitive.  This is what lets the reference's model scripts (models/wide_deep/src/wide_and_deep.py:136-492,
models/deep_and_cross/src/deep_and_cross.py:206-354) run unmodified AND at the engine's speed.

Recognition is structural, not by class or attribute name: a train cell owning one FTRL and one (Lazy)Adam optimizer over a
model with two embedding lookups sharing one id tensor (one of width 1: the wide table) and a chain of DenseLayer-shaped cells
(`weight` [in, out] + `bias` [out]) from F * D down to 1 is Wide&Deep; one Adam over one embedding lookup, a chain of
DenseLayer-shaped cells and a stack of cross layers (`[X, 1]` weight and bias pairs) is Deep&Cross.  Because structure alone
cannot prove the arithmetic (activation, loss, L2 term), a lowering is VERIFIED before it is used: the engine's logits on the
first batch must equal the cell's own eager forward, and the log loss of those logits the cell's own first loss; a mismatch
refuses the lowering (the cell then runs eagerly) -- it never silently computes something else.  Verification runs BEFORE
anything of the cell is touched: parameters are copied into the engine, checked, and only then re-bound (`commit`).

After lowering the cell's Parameters ARE the engine's memory (tables, dense weights, optimizer state are re-bound as views), so
an evaluation network over the same model, checkpoints and `asnumpy()` see what the engine trains.

**Distributed train cells** (models/wide_deep/src/wide_and_deep.py:458-470,487-489: `reducer_flag` + two
`DistributedGradReducer(mean, degree)`, set up by train_and_eval_distribute.py:123-140; models/deep_and_cross/train.py:57-62).
A cell that reduces its gradients over N > 1 ranks is NEVER lowered onto a one-rank engine.  Wide&Deep with row gradients
(`sparse=True` / hash tables: LazyAdam + FTRL on the union of the ranks' rows) is lowered onto the ROW-SHARDED engine
(`WideDeepEngine(rank, world)`: proven equal to the reference's data-parallel step, tests/test_wide_deep_2rank_gpu.py): each
rank keeps the rows it owns, requests / rows / row gradients move point to point, the dense net stays data parallel with one
all-reduce(mean).  The cell's tables are then MIRRORS of the shards, refreshed by `LoweredStep.sync()` (a collective
`mindspore.Model.train` calls on every rank at the steps at which some rank checkpoints, and at the end); evaluation through
the model cell is routed to the engine's collective `predict`.  Every other distributed cell (dense table gradients, Deep&Cross,
DeepFM) is REFUSED and runs primitive by primitive with its all-reduce.  The decision is taken from structure + world size and
agreed by all-reduce, so every rank takes the same branch.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .wide_deep_mlp import UnsupportedNet


class LoweringRefused(Exception):
    """Why a train cell was not lowered (kept on the cell as `_lowering_refused`; the cell runs eagerly)."""


def _cells(cell):
    return [c for _, c in cell.cells_and_names()]


def _opt_kind(c):
    n = type(c).__name__
    return n if n in ("FTRL", "Adam", "LazyAdam") and hasattr(c, "parameters") and hasattr(c, "loss_scale") else None


def _is_dense_layer(c):
    p = c.__dict__.get("_params", {})
    w, b = p.get("weight"), p.get("bias")
    return (isinstance(w, torch.Tensor) and isinstance(b, torch.Tensor) and w.dim() == 2 and b.dim() == 1 and w.shape[1] == b.shape[0])


def _is_cross_layer(c):
    ps = [q for q in c.__dict__.get("_params", {}).values() if isinstance(q, torch.Tensor)]
    return len(ps) == 2 and all(q.dim() == 2 and q.shape[1] == 1 for q in ps) and ps[0].shape == ps[1].shape


def _lookup_table(c):
    """(table, kind) of an embedding-lookup cell: kind 'dense' / 'sparse' (RowTensor gradient) / 'hash' (MapParameter)."""
    t = c.__dict__.get("_params", {}).get("embedding_table")
    if t is None:
        return None
    if hasattr(t, "_store"):
        return t, "hash"
    if not (isinstance(t, torch.Tensor) and t.dim() == 2):
        return None
    g = getattr(c, "gather", None) or getattr(c, "gatherv2", None)
    sparse = bool(getattr(c, "sparse", False)) or type(g).__name__ in ("SparseGatherV2", "EmbeddingLookup")
    return t, ("sparse" if sparse else "dense")


def _chain(layers, width_in):
    """Orders DenseLayer-shaped cells into a chain width_in -> ... by their weight shapes; None if they do not form one."""
    if layers and layers[0].weight.shape[0] == width_in and all(a.weight.shape[1] == b.weight.shape[0] for a, b in zip(layers, layers[1:])):
        return list(layers)                    # definition order already is the chain (the usual case; also resolves equal widths)
    rest, out, w = list(layers), [], width_in
    while rest:
        nxt = [c for c in rest if c.weight.shape[0] == w]
        if len(nxt) != 1:
            return None
        out.append(nxt[0])
        rest.remove(nxt[0])
        w = nxt[0].weight.shape[1]
    return out


def _rebind(param, view):
    """The Parameter becomes a view of engine memory (same shape / dtype); autograd leaf status is kept."""
    if tuple(param.shape) != tuple(view.shape) or param.dtype != view.dtype:
        raise LoweringRefused(f"cannot alias parameter {getattr(param, 'name', '?')}: {tuple(param.shape)} vs {tuple(view.shape)}")
    with torch.no_grad():
        torch.Tensor.data.__set__(param, view.detach())


def _as_t(x, cls):
    return x.as_subclass(cls) if isinstance(x, torch.Tensor) else x


class LoweredStep:
    """What Model.train / RecModel.online_train call instead of the cell."""

    def __init__(self, engine, out_cls, n_losses, deep_loss=None, optimizers=()):
        self.engine, self._cls, self._n, self._deep_loss, self._opts = engine, out_cls, n_losses, deep_loss, tuple(optimizers)
        self.sharded = False          # the cell's tables are mirrors of row shards (distributed lowering)
        self.dirty = False            # ... and the shards have been trained since the mirrors were last refreshed
        self._binds, self._after_commit, self._sync_fns = [], [], []
        self.committed = False

    # ---- build -> verify -> commit ---------------------------------------------------------------------------------
    def commit(self):
        """Re-binds the cell's Parameters as views of engine memory (nothing of the cell was touched before this)."""
        for param, view in self._binds:
            _rebind(param, view)
        for fn in self._after_commit:
            fn()
        self._binds, self._after_commit, self.committed = [], [], True

    def discard(self):
        """A refused lowering: the engine is dropped, the cell is as it was."""
        eng, self.engine = self.engine, None
        self._binds, self._after_commit, self._sync_fns = [], [], []
        if eng is not None and hasattr(eng, "release_graphs"):
            eng.release_graphs()

    def sync(self):
        """Row-sharded lowering: the cell's full-size tables (and their optimizer slots) are refreshed from all ranks' shards.
        A COLLECTIVE: every rank calls it (mindspore.Model.train does, at agreed steps); free when nothing was trained since."""
        if self.sharded and self.dirty:
            for fn in self._sync_fns:
                fn()
            self.dirty = False

    def _sync_back(self):
        """The cell's optimizers keep their step scalars on the host (and checkpoint them): they follow the engine's."""
        e = self.engine
        for o in self._opts:
            o.global_step = int(e.step_count)
            if hasattr(o, "beta1_power"):
                o.beta1_power, o.beta2_power = np.float32(e.beta1_power), np.float32(e.beta2_power)
            sc = o.__dict__.get("_scalars", {})
            for n, t in sc.items():
                t.as_subclass(torch.Tensor)[0] = float(getattr(o, n))

    def _outs(self, loss):
        self._sync_back()
        loss = _as_t(loss.reshape(()), self._cls)
        if self._n == 1:
            return loss
        return loss, (self._deep_loss(loss) if self._deep_loss is not None else loss)

    def __call__(self, ids, wts, label):
        self.dirty = True
        return self._outs(self.engine.train_step(_raw(ids), _raw(wts), _raw(label)))

    def run_sink(self, batches):
        """One sink of steps (dataset_sink_mode): ONE HIP graph launch where the step has a graph; the last step's outputs."""
        bs = [tuple(_raw(t) for t in b) for b in batches]
        self.dirty = True
        if hasattr(self.engine, "train_steps"):
            losses = self.engine.train_steps(bs)
            return self._outs(losses[-1])
        for b in bs:
            loss = self.engine.train_step(*b)
        return self._outs(loss)


def _raw(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t


class DistPlan:
    """How a train cell is distributed: `world` ranks of torch.distributed, `reduces` = it owns gradient reducers that act
    (world > 1).  `shard`: lower onto the row-sharded engine."""

    def __init__(self, world=1, rank=0, reduces=False):
        self.world, self.rank, self.reduces = int(world), int(rank), bool(reduces)

    @property
    def shard(self):
        return self.reduces and self.world > 1


def _reducers(cell):
    return [c for c in _cells(cell) if type(c).__name__ == "DistributedGradReducer"]


def dist_plan(cell):
    """The distribution of a train cell, from what every rank sees alike: the process group's size, the cell's own
    DistributedGradReducers / `reducer_flag` (wide_and_deep.py:458-470) and the parallel mode
    (train_and_eval_distribute.py:135-138).  Raises LoweringRefused for distributions no engine mode computes."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    reds = _reducers(cell)
    flag = any(bool(c.__dict__.get("reducer_flag", False)) for c in _cells(cell))
    mode = None
    try:
        from mindspore import context as _ctx                       # (the compat package, when the cell came from it)
        mode = _ctx.get_auto_parallel_context("parallel_mode")
    except Exception:       # noqa: BLE001
        pass
    if world == 1:
        return DistPlan()
    if mode in ("auto_parallel", "semi_auto_parallel"):
        raise LoweringRefused(f"distributed train cell: parallel_mode {mode!r} (operator sharding strategies) is not lowered")
    if not reds:
        if flag or mode in ("data_parallel", "hybrid_parallel"):
            raise LoweringRefused("distributed train cell: data-parallel context / reducer_flag without a DistributedGradReducer "
                                  "the lowering can read (mean, degree)")
        return DistPlan(world, rank, False)                  # independent replicas: what the cell itself computes
    for r in reds:
        if not bool(getattr(r, "mean", False)) or int(getattr(r, "degree", 0)) != world:
            raise LoweringRefused(f"distributed train cell: DistributedGradReducer(mean={getattr(r, 'mean', None)}, "
                                  f"degree={getattr(r, 'degree', None)}) over {world} ranks -- only mean over all ranks is lowered")
    return DistPlan(world, rank, True)


def _agree(ok, plan):
    """Every rank of a distributed cell takes the same branch: True only if `ok` on all of them."""
    if not plan.shard:
        return bool(ok)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


_comm_factory = None          # tests: several ranks on one GPU inject their staged communicator here


def lower_train_step(cell, first_batch=None):
    """-> LoweredStep (verified on `first_batch` and committed), or None when the cell is not lowered (the reason is left in
    cell._lowering_refused; the cell is untouched and runs primitive by primitive)."""
    d = cell.__dict__
    try:
        plan = dist_plan(cell)
    except LoweringRefused as e:
        d["_lowering_refused"] = str(e)
        return None
    low, err = None, None
    try:
        low = _lower_wide_deep(cell, plan) or _lower_deep_cross(cell, plan) or _lower_deepfm(cell, plan)
        if low is None:
            err = "not a Wide&Deep, Deep&Cross or DeepFM train step by structure"
    except (LoweringRefused, UnsupportedNet) as e:
        err = str(e)
    if not _agree(low is not None, plan) and low is not None:
        low.discard()
        low, err = None, "another rank refused the lowering"
    if low is None:
        d["_lowering_refused"] = err
        return None
    if first_batch is not None and len(first_batch) >= 2 and all(isinstance(t, torch.Tensor) for t in first_batch[:2]):
        try:
            low.verify(*first_batch[:3])
        except LoweringRefused as e:
            err = str(e)
        if not _agree(err is None, plan):
            err = err or "another rank's verification failed"
        if err is not None:
            low.discard()
            d["_lowering_refused"] = err
            return None
        low.verified = True
    low.commit()
    return low


class _EngineMapStore:
    """What stands behind a `MapParameter` once its model is lowered: the table is one of the engine's two row tables and its key
    index is the engine's -- ONE index for both tables, as both are looked up with the same ids (wide_and_deep.py:300-302).  Reads,
    inserts by lookup, puts, exports and the optimizer slots' export work on engine memory; what would let the two tables' key
    sets drift apart (erase, evict, clear on ONE of them) is refused -- the cell then needs its own two indexes back, i.e. the
    primitive-by-primitive path."""

    def __init__(self, eng, which, sigma, seed, key_dtype):
        self.eng, self.which, self.sigma, self.seed, self.key_dtype = eng, which, sigma, int(seed), key_dtype
        self._winner = None

    def _table(self):
        return self.eng.deep if self.which == "deep" else self.eng.wide

    def _slots(self):
        e = self.eng
        return ({"moment1": e.deep_m, "moment2": e.deep_v} if self.which == "deep" else {"accum": e.wide_accum, "linear": e.wide_linear})

    def _refuse(self, what):
        raise NotImplementedError(f"MapParameter.{what} on a table of a lowered (GRAPH_MODE) model: both tables share the engine's one "
                                  f"key index; run the model in PYNATIVE_MODE, or {what} before the first training step")

    def get(self, keys, insert):
        e = self.eng
        rows = e.index.lookup(keys, insert=bool(insert), tables=e._map_tables() if insert else ())
        out = ops.gather_rows(self._table(), rows)
        if not insert:
            e.index.fill_missing(keys, rows, out, self.sigma, None, self.seed)
        return out

    def put(self, keys, vals):
        e = self.eng
        rows = e.index.lookup(keys, insert=True, tables=e._map_tables())
        if self._winner is None:
            self._winner = torch.full((e.index.capacity,), -1, dtype=torch.int32, device=e.device)
        ops.put_rows_last_(self._table(), rows, vals, self._winner)

    def size(self):
        return len(self.eng.index)

    def export(self):
        k, r = self.eng.index.export()
        return k.to(self.key_dtype), ops.gather_rows(self._table(), r)

    def export_data(self, incremental):
        if incremental:
            self._refuse("export_data(incremental=True)")
        k, v = self.export()
        return k, v, torch.zeros(k.numel(), dtype=torch.int32, device=k.device)

    def import_data(self, data):
        if len(data) > 2 and data[2] is not None and bool((data[2] == 2).any()):
            self._refuse("import_data with erased keys")
        if data[0].numel():
            self.put(data[0].to(self.eng.device).reshape(-1), data[1].to(self.eng.device, torch.float32).reshape(data[0].numel(), -1))

    def export_slots(self):
        _, r = self.eng.index.export()
        return {n: ops.gather_rows(t, r).cpu().numpy() for n, t in self._slots().items()}

    def import_slots(self, keys, slots):
        e = self.eng
        rows = e.index.lookup(keys.reshape(-1).to(e.device), insert=True, tables=e._map_tables())
        mine = self._slots()
        for n, vals in slots.items():
            if n not in mine:
                raise KeyError(f"slot {n!r} does not belong to the {self.which} table's optimizer")
            ops.scatter_rows_(mine[n], rows, vals.to(e.device, torch.float32))

    def erase(self, keys):
        if keys.numel():
            self._refuse("erase")

    def evict(self):
        self._refuse("evict")

    def clear(self):
        if len(self.eng.index):
            self._refuse("clear")

    def apply_lazy_adam(self, *a, **k):
        self._refuse("optimizer apply outside the lowered step")

    apply_ftrl = apply_lazy_adam


class _ShardedMapStore(_EngineMapStore):
    """A MapParameter of a model lowered onto KEY-SHARDED hash tables: a key lives on the rank that owns it (hash(key) mod n), so
    what the cell's MapParameter can hand out locally is the snapshot of all ranks' shards the last `LoweredStep.sync()` took
    (mindspore.Model.train takes one on every rank wherever a rank checkpoints, and at the end).  Lookups outside the lowered
    step go through the engine's collective forward (the model cell's evaluation is routed there); single-rank reads of a stale
    snapshot are refused rather than answered with old rows."""
    snapshot = None

    def _snap(self, what):
        low = getattr(self, "low", None)
        if self.snapshot is None or (low is not None and low.dirty):
            raise RuntimeError(f"MapParameter.{what} on a key-sharded lowered model: the shards were trained since the last sync -- "
                               f"call LoweredStep.sync() (mindspore.Model.sync_parameters()) on every rank first")
        return self.snapshot

    def get(self, keys, insert):
        self._refuse("get outside the lowered step (keys live on their owner ranks: evaluate through the model cell)")

    def put(self, keys, vals):
        self._refuse("put")

    def size(self):
        return int(self._snap("size")["keys"].numel())

    def export(self):
        s = self._snap("export")
        return s["keys"].to(self.key_dtype), s["deep" if self.which == "deep" else "wide"]

    def export_slots(self):
        s = self._snap("export_slots")
        names = ("moment1", "moment2") if self.which == "deep" else ("accum", "linear")
        return {n: s[n].cpu().numpy() for n in names}

    def import_slots(self, keys, slots):
        self._refuse("import_slots")

    def import_data(self, data):
        self._refuse("import_data")


def _hash_spec(mp):
    """(sigma, seed, key dtype, capacity) of a MapParameter the engine's hash mode can stand in for, or a refusal."""
    from mindspore.experimental import MAX_SIZE          # (the compat package: this path only runs under it)
    st = getattr(mp._store, "m", None)
    if st is None or not hasattr(st, "index"):
        raise LoweringRefused("the MapParameter's store is not the HIP one")
    if mp.default_value != "normal" or st._fill is not None:
        raise LoweringRefused("hash tables are lowered with default_value='normal' only")
    if int(mp.permit_filter_value) != 1 or int(mp.evict_filter_value) != MAX_SIZE:
        raise LoweringRefused("hash tables with admission / eviction filters are not lowered")
    if mp.key_dtype not in (torch.int32, torch.int64):
        raise LoweringRefused("hash tables are lowered with int32 / int64 keys")
    return float(st._sigma), int(st.seed), mp.key_dtype, int(st.capacity)


# ---- Wide&Deep -------------------------------------------------------------------------------------------------------------------
def _lower_wide_deep(cell, plan):
    from .wide_deep import WideDeepConfig, WideDeepEngine
    opts = [(c, _opt_kind(c)) for c in cell.cells() if _opt_kind(c)]
    kinds = [k for _, k in opts]
    if len(opts) != 2 or kinds.count("FTRL") != 1:
        return None
    ftrl = next(c for c, k in opts if k == "FTRL")
    adam = next(c for c, k in opts if k != "FTRL")
    owner = None
    for c in _cells(cell):
        looks = [x for x in c.cells() if _lookup_table(x)]
        if len(looks) == 2 and sum(_is_dense_layer(x) for x in c.cells()) >= 2:
            owner, lookups = c, looks
            break
    if owner is None:
        return None
    wide_l = [x for x in lookups if _lookup_table(x)[0].shape[-1] == 1 or getattr(x, "embedding_size", 0) == 1]
    deep_l = [x for x in lookups if x not in wide_l]
    if len(wide_l) != 1 or len(deep_l) != 1:
        return None
    (deep_t, deep_kind), (wide_t, wide_kind) = _lookup_table(deep_l[0]), _lookup_table(wide_l[0])
    if deep_kind != wide_kind:
        raise LoweringRefused("the two tables are looked up differently")
    hashed = deep_kind == "hash"
    if hashed:
        (sig_d, seed_d, kd, cap_d), (sig_w, seed_w, kw_, cap_w) = _hash_spec(deep_t), _hash_spec(wide_t)
        if sig_d != sig_w or seed_w != seed_d + 1 or kd != kw_:
            raise LoweringRefused("the two hash tables' default-value streams / key types are not the engine's (same sigma, "
                                  "consecutive seeds, one key dtype)")
        if deep_t.value_shape[1:] or wide_t.value_shape != (1,):
            raise LoweringRefused("hash tables are lowered with value_shape (D,) and (1,)")
        V, D = max(cap_d, cap_w), int(deep_t.value_shape[0])
    else:
        V, D = int(deep_t.shape[0]), int(deep_t.shape[1])
    B, F = int(getattr(owner, "batch_size", 0) or getattr(owner, "B", 0)), int(getattr(owner, "field_size", 0) or getattr(owner, "F", 0))
    if B <= 0 or F <= 0:
        raise LoweringRefused("the model does not state batch_size / field_size")
    layers = _chain([x for x in owner.cells() if _is_dense_layer(x)], F * D)
    if layers is None or layers[-1].weight.shape[1] != 1 or len(layers) < 2:
        raise LoweringRefused("the DenseLayer cells do not form a chain F * D -> ... -> 1")
    extra = [p for p in owner.__dict__["_params"].values() if isinstance(p, torch.Tensor) and p.numel() == 1]
    if len(extra) != 1:
        raise LoweringRefused("expected exactly one scalar parameter (the wide bias) on the model")
    wide_b = extra[0]
    in_ftrl = any(p is wide_b for p in ftrl.parameters)
    if not in_ftrl and not any(p is wide_b for p in adam.parameters):
        raise LoweringRefused("the wide bias belongs to neither optimizer")
    if not any(p is wide_t for p in ftrl.parameters) or not any(p is deep_t for p in adam.parameters):
        raise LoweringRefused("unexpected parameter split: the wide table must belong to FTRL, the deep table to (Lazy)Adam")
    lazy = type(adam).__name__ == "LazyAdam"
    if hashed:
        deep_kind = "sparse"                   # MapTensorGet's gradient is a row gradient by construction
    if deep_kind == "sparse" and not lazy:
        raise LoweringRefused("sparse lookups under a non-lazy Adam (RowTensor gradients densified every step) have no engine mode")
    if deep_kind == "dense" and lazy:
        lazy = False               # LazyAdam on dense gradients IS Adam
    loss_cell = next((c for c in _cells(cell) if any(x is owner for x in c.cells()) and hasattr(c, "l2_coef")), None)
    l2 = float(getattr(loss_cell, "l2_coef", 0.0)) if loss_cell is not None else 0.0
    no_l2 = bool(getattr(loss_cell, "no_l2loss", not getattr(loss_cell, "with_l2", False))) if loss_cell is not None else True
    if deep_kind == "sparse" and not no_l2:
        raise LoweringRefused("an L2 term over a sparsely-updated table has no engine mode")
    half = bool(getattr(layers[0], "convert_dtype", getattr(layers[0], "half", False)))
    drop = bool(getattr(layers[0], "drop_out", False))
    keep = float(getattr(getattr(layers[0], "dropout", None), "keep_prob", 0.5)) if drop else 0.5
    if float(adam.loss_scale) != float(ftrl.loss_scale) or float(getattr(cell, "sens", adam.loss_scale)) != float(adam.loss_scale):
        raise LoweringRefused("sens and the optimizers' loss_scale differ")
    if any(getattr(o, "weight_decay", 0.0) for o in (adam, ftrl)) or getattr(adam, "use_nesterov", False):
        raise LoweringRefused("weight decay / Nesterov are not lowered")
    if abs(float(getattr(ftrl, "lr_power", -0.5)) + 0.5) > 1e-9:
        raise LoweringRefused(f"FTRL lr_power {ftrl.lr_power} is not lowered (the engine computes lr_power = -0.5)")
    if plan.shard and deep_kind == "dense":
        raise LoweringRefused("distributed train cell with dense table gradients (sparse=False: every rank all-reduces [V, D]): no "
                              "engine mode -- the step runs primitive by primitive with its all-reduce")
    cfg = WideDeepConfig(vocab_size=V, emb_dim=D, field_size=F, batch_size=B, deep_layer_dim=[int(l.weight.shape[1]) for l in layers[:-1]],
                         sens=float(adam.loss_scale), adam_lr=adam.get_lr(), adam_eps=float(adam.eps), ftrl_lr=ftrl.get_lr(), ftrl_l1=ftrl.l1,
                         ftrl_l2=ftrl.l2, ftrl_initial_accum=ftrl.initial_accum, mlp_dtype="fp16" if half else "fp32",
                         sparse=deep_kind == "sparse", l2_coef=l2 if not no_l2 else 0.0, dropout_flag=drop, dropout_keep_prob=keep,
                         id_dtype="int64" if hashed and kd == torch.int64 else "int32", wide_b_optimizer="ftrl" if in_ftrl else "adam",
                         **(dict(dynamic_embedding=True, hash_capacity=V, init_sigma=sig_d, seed=seed_d) if hashed else {}))
    if not cfg.sparse and no_l2:
        cfg.l2_coef = 0.0
    dev = torch.device(deep_t.device)
    if dev.type != "cuda":
        raise LoweringRefused("parameters are not on an MI355X")
    if abs(float(adam.beta1) - 0.9) > 1e-6 or abs(float(adam.beta2) - 0.999) > 1e-6:
        raise LoweringRefused("non-default Adam betas are not lowered")
    if plan.shard:
        # The reference's data-parallel step over row gradients == the row-sharded step (owner = id mod n / hash(key) mod n; the
        # ranks' row gradients meet at the owner, gradients_mean is the appliers' 1 / n; the dense net all-reduces its mean)
        comm = _comm_factory() if _comm_factory is not None else None
        eng = WideDeepEngine(cfg, dev, rank=plan.rank, world=plan.world, comm=comm)
    else:
        eng = WideDeepEngine(cfg, dev)
    n, me = plan.world if plan.shard else 1, plan.rank if plan.shard else 0
    ref_t = layers[0].weight
    cls = type(ref_t).__mro__[1] if type(ref_t).__name__ == "Parameter" else type(ref_t)

    def deep_loss(loss):
        if cfg.sparse or cfg.l2_coef == 0.0:
            return loss
        return _as_t(_raw(loss) + (eng._l2_sumsq * (cfg.l2_coef * 0.5)).to(torch.float32).view(()), cls)

    low = LoweredStep(eng, cls, 2, deep_loss, optimizers=(adam, ftrl))
    low.kind, low.sharded = "wide_deep", bool(plan.shard)
    bind = low._binds.append
    mirrors = []                  # row-sharded dense-storage tables: (the cell's full [V, .] tensor, this rank's rows of it)

    def table(cell_t, eng_t):
        """A table or one of its optimizer slots moves into the engine: whole (one rank: the cell's tensor becomes a view of engine
        memory) or this rank's rows r with r mod n == rank (shards: the cell's tensor stays a mirror, refreshed by sync())."""
        if n == 1:
            eng_t.copy_(_raw(cell_t))
            bind((cell_t, eng_t))
        else:
            eng_t.copy_(_raw(cell_t)[me::n])
            mirrors.append((cell_t, eng_t))

    # parameters and optimizer state are COPIED into the engine here; the cell's Parameters are re-bound as views of engine memory
    # by commit(), after the verification
    with torch.no_grad():
        if hashed:
            _move_hash_tables(eng, deep_t, wide_t, n, me)
        else:
            table(deep_t, eng.deep)
            table(wide_t, eng.wide)
        eng.load_dense_parameters([_raw(l.weight) for l in layers], [_raw(l.bias) for l in layers], extra=_raw(wide_b))
        st = adam.__dict__.get("_state", {})
        for (prefix, pid), s in st.items():
            tgt = None
            if pid == id(deep_t):
                if not hashed:
                    table(s, eng.deep_m if prefix == "moment1" else eng.deep_v)
                continue
            for i, l in enumerate(layers):
                if pid == id(l.weight):
                    tgt = (eng.dense_m if prefix == "moment1" else eng.dense_v)[_off(eng, 2 * i)].view(l.weight.shape)
                elif pid == id(l.bias):
                    tgt = (eng.dense_m if prefix == "moment1" else eng.dense_v)[_off(eng, 2 * i + 1)].view(l.bias.shape)
            if tgt is not None:
                tgt.copy_(_raw(s))
                bind((s, tgt))
        for (prefix, pid), s in ftrl.__dict__.get("_state", {}).items():
            if pid == id(wide_t):
                if not hashed:
                    table(s, eng.wide_accum if prefix == "accum" else eng.wide_linear)
            elif pid == id(wide_b) and in_ftrl:
                tgt = (eng.dense_m if prefix == "accum" else eng.dense_v)[eng._wb_off:eng._wb_off + 1]
                tgt.copy_(_raw(s).reshape(1))
                bind((s, tgt.view(s.shape)))
    eng.beta1_power, eng.beta2_power = np.float32(adam.beta1_power), np.float32(adam.beta2_power)
    eng.step_count = int(adam.global_step)
    for i, l in enumerate(layers):
        bind((l.weight, eng.dense[2 * i]))
        bind((l.bias, eng.dense[2 * i + 1]))
    bind((wide_b, eng.wide_b.view(wide_b.shape)))
    if hashed:
        store = _ShardedMapStore if n > 1 else _EngineMapStore
        sd, sw = store(eng, "deep", sig_d, seed_d, kd), store(eng, "wide", sig_w, seed_w, kd)

        def swap_stores():
            deep_t._store, wide_t._store = sd, sw
        low._after_commit.append(swap_stores)
        if n > 1:
            sd.low = sw.low = low
            low._sync_fns.append(lambda: _sync_hash_mirrors(eng, sd, sw))
    elif n > 1:
        low._sync_fns.append(lambda: _sync_row_mirrors(eng, mirrors, n))
    if n > 1:
        # evaluation through the model cell (PredictWithSigmoid and the like call `owner(ids, wts)`): the engine's collective
        # forward over the shards instead of eager primitives over the (stale) mirrors
        def fwd(ids, wts):
            logit = eng.predict(_raw(ids), _raw(wts))[0]
            like = low._eval_like                       # what the model cell's own construct returned during the verification
            if isinstance(like, (tuple, list)):
                return type(like)([_as_t(logit.view(like[0].shape), cls)] + list(like[1:]))
            return _as_t(logit.view(like.shape) if like is not None else logit, cls)
        low._eval_like = None
        low._after_commit.append(lambda: owner.__dict__.__setitem__("_lowered_forward", fwd))
    low.verify = lambda ids, wts, label=None: _verify(low, eng, owner, loss_cell, ids, wts, label, half)
    return low


def _sync_row_mirrors(eng, mirrors, n):
    """Dense-storage tables under row sharding: all ranks' rows back into the cell's full-size tensors (row id = local * n + rank)."""
    with torch.no_grad():
        for full, shard in mirrors:
            rows, cnts = eng.comm.all_gather_rows(shard)
            off = 0
            for r, c in enumerate(cnts):
                _raw(full)[r::n] = rows[off:off + c].to(full.device)
                off += c


def _sync_hash_mirrors(eng, sd, sw):
    """Hash tables under key sharding: every rank's (keys, rows, optimizer slots) gathered into the snapshot the cell's two
    MapParameters export from (every key lives on one rank: the snapshot is the concatenation)."""
    with torch.no_grad():
        keys, rows = eng.index.export()
        snap = {"keys": eng.comm.all_gather_rows(keys.contiguous())[0]}
        for name, t in (("deep", eng.deep), ("wide", eng.wide), ("moment1", eng.deep_m), ("moment2", eng.deep_v),
                        ("accum", eng.wide_accum), ("linear", eng.wide_linear)):
            snap[name] = eng.comm.all_gather_rows(ops.gather_rows(t, rows))[0]
    sd.snapshot = sw.snapshot = snap


def _move_hash_tables(eng, deep_mp, wide_mp, n=1, me=0):
    """Keys, rows and optimizer slots of the cell's two MapParameters into the engine's one key index and its row tables.
    n > 1: only the keys this rank owns (owner = hash(key) mod n, the routing kernel's own function)."""
    kd, vd = deep_mp._store.export()
    kw, vw = wide_mp._store.export()
    if kd.numel() != kw.numel():
        raise LoweringRefused("the two hash tables hold different key sets")
    if kd.numel() == 0:
        return
    od, ow = torch.argsort(kd), torch.argsort(kw)
    if not torch.equal(kd[od], kw[ow]):
        raise LoweringRefused("the two hash tables hold different key sets")
    sd = {n_: torch.from_numpy(v).to(eng.device) for n_, v in deep_mp._store.export_slots().items()}
    sw = {n_: torch.from_numpy(v).to(eng.device) for n_, v in wide_mp._store.export_slots().items()}
    w_of_d = ow[torch.searchsorted(kw[ow], kd)]                                               # wide export position of each deep key
    pick = None
    if n > 1:
        _, perm, counts = ops.shard_route(kd.contiguous(), n, hashed=True)
        c = counts.cpu().tolist()
        lo = sum(c[:me])
        pick = perm[lo:lo + c[me]].long()
        if pick.numel() == 0:
            return
        kd, vd, w_of_d = kd[pick], vd[pick], w_of_d[pick]
        sd = {k: v[pick] for k, v in sd.items()}
    rows = eng.index.lookup(kd.contiguous(), insert=True, tables=eng._map_tables())          # the deep table's row order
    ops.scatter_rows_(eng.deep, rows, vd.to(torch.float32))
    ops.scatter_rows_(eng.wide, rows, vw[w_of_d].to(torch.float32).reshape(-1, 1))
    for n_, tgt in (("moment1", eng.deep_m), ("moment2", eng.deep_v)):
        if n_ in sd:
            ops.scatter_rows_(tgt, rows, sd[n_].to(torch.float32))
    for n_, tgt in (("accum", eng.wide_accum), ("linear", eng.wide_linear)):
        if n_ in sw:
            ops.scatter_rows_(tgt, rows, sw[n_][w_of_d].to(torch.float32).reshape(-1, 1))


def _off(eng, k):
    t = eng.dense[k]
    return slice(t.storage_offset(), t.storage_offset() + t.numel())


def _log_loss(logit, label):
    """mean sigmoid cross-entropy in float64 on the host (a check of two scalars, not a step's arithmetic)."""
    z, y = logit.detach().double().cpu().reshape(-1), label.detach().double().cpu().reshape(-1)
    return float((torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs()))).mean())


def _verify(low, eng, owner, loss_cell, ids, wts, label, half, loss_extra=None):
    """Before anything of the cell is re-bound: the engine's inference logits against the model cell's own eager forward on the
    same batch, and -- where the loss cell is known -- the log loss of the engine's logits against the loss cell's own first
    output (the loss function is part of what the engine computes).  Raises LoweringRefused."""
    with torch.no_grad():
        mode = owner.training
        owner.set_train(False)
        out = owner(ids, wts)
        ref = _raw(out[0] if isinstance(out, (tuple, list)) else out).reshape(-1).float()
        got = eng.predict(_raw(ids), _raw(wts))[0].reshape(-1).float()
        loss_ref = None
        if loss_cell is not None and label is not None:
            lmode = loss_cell.training
            loss_cell.set_train(False)
            lo = loss_cell(ids, wts, label)
            loss_cell.set_train(lmode)
            loss_ref = float(_raw(lo[0] if isinstance(lo, (tuple, list)) else lo).reshape(-1)[0])
        owner.set_train(mode)
    low._eval_like = out
    tol = 2e-2 if half else 1e-4
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    if not err <= tol:
        raise LoweringRefused(f"the engine's logits differ from the cell's eager forward (relative {err:.3g} > {tol}): not this model")
    if loss_ref is not None:
        mine = _log_loss(got, _raw(label)) + (loss_extra() if loss_extra is not None else 0.0)
        if not abs(mine - loss_ref) <= tol * max(abs(loss_ref), 1e-6):
            raise LoweringRefused(f"the cell's loss ({loss_ref:.6g}) is not the mean sigmoid cross-entropy of its logits "
                                  f"({mine:.6g}): not this model")
    return err


# ---- Deep&Cross ------------------------------------------------------------------------------------------------------------------
def _lower_deep_cross(cell, plan):
    from .deep_cross import DeepCrossConfig, DeepCrossEngine
    opts = [(c, _opt_kind(c)) for c in cell.cells() if _opt_kind(c)]
    if len(opts) != 1 or opts[0][1] != "Adam":
        return None
    adam = opts[0][0]
    owner = None
    for c in _cells(cell):
        if (sum(1 for x in c.cells() if _lookup_table(x)) == 1 and sum(_is_cross_layer(x) for x in c.cells()) >= 1
                and sum(_is_dense_layer(x) for x in c.cells()) == 3):
            owner = c
            break
    if owner is None:
        return None
    if plan.shard:
        raise LoweringRefused("distributed train cell (Deep&Cross under DistributedGradReducer, models/deep_and_cross/train.py:57-62): "
                              "DeepCrossEngine is a one-rank engine -- the step runs primitive by primitive with its all-reduce")
    look = next(x for x in owner.cells() if _lookup_table(x))
    table, kind = _lookup_table(look)
    if kind != "dense":
        raise LoweringRefused("Deep&Cross is lowered with a dense Gather table only")
    V, D = int(table.shape[0]), int(table.shape[1])
    B, F = int(getattr(owner, "batch_size", 0)), int(getattr(owner, "field_size", 0))
    if B <= 0 or F <= 0:
        raise LoweringRefused("the model does not state batch_size / field_size")
    X = F * D
    crosses = [x for x in owner.cells() if _is_cross_layer(x)]         # in attribute (= application) order
    dense = [x for x in owner.cells() if _is_dense_layer(x) and not _is_cross_layer(x)]
    hidden = _chain([x for x in dense if x.weight.shape[1] != 1], X)
    last = [x for x in dense if x.weight.shape[1] == 1]
    if hidden is None or len(hidden) != 2 or len(last) != 1 or last[0].weight.shape[0] != X + hidden[1].weight.shape[1]:
        raise LoweringRefused("not the Deep&Cross shape: two hidden DenseLayers, six-ish cross layers, concat -> 1")
    if any(c.__dict__["_params"] and tuple(next(iter(c.__dict__["_params"].values())).shape) != (X, 1) for c in crosses):
        raise LoweringRefused("cross layers must be [F * D, 1]")
    if getattr(adam, "weight_decay", 0.0) or abs(float(adam.beta1) - 0.9) > 1e-6 or abs(float(adam.beta2) - 0.999) > 1e-6:
        raise LoweringRefused("non-default Adam settings are not lowered")
    cfg = DeepCrossConfig(vocab_size=V, emb_dim=D, field_size=F, batch_size=B, deep_layer_dim=[int(h.weight.shape[1]) for h in hidden],
                          cross_layer_num=len(crosses), learning_rate=adam.get_lr(), eps=float(adam.eps), loss_scale=float(adam.loss_scale))
    if table.device.type != "cuda":
        raise LoweringRefused("parameters are not on an MI355X")
    if int(adam.global_step):
        raise LoweringRefused("Deep&Cross is lowered from a fresh optimizer only")
    eng = DeepCrossEngine(cfg, table.device)
    if not eng._native:
        raise LoweringRefused("these Deep&Cross shapes have no hand-written path")
    W1, b1, W2, b2, W3, b3, cw, cb = eng.dense
    cls = type(table).__mro__[1] if type(table).__name__ == "Parameter" else type(table)
    low = LoweredStep(eng, cls, 1, optimizers=(adam,))
    bind = low._binds.append
    with torch.no_grad():
        eng.table.copy_(_raw(table))
        bind((table, eng.table))
        for t, src in ((W1, hidden[0].weight), (b1, hidden[0].bias), (W2, hidden[1].weight), (b2, hidden[1].bias), (W3, last[0].weight), (b3, last[0].bias)):
            t.copy_(_raw(src).reshape(t.shape))
            bind((src, t.detach().view(src.shape)))
        for l, c in enumerate(crosses):
            ps = list(c.__dict__["_params"].items())
            wname = next((n for n, _ in ps if "weight" in n), ps[0][0])
            bname = next(n for n, _ in ps if n != wname)
            cw[l].copy_(_raw(c.__dict__["_params"][wname]).reshape(-1))
            cb[l].copy_(_raw(c.__dict__["_params"][bname]).reshape(-1))
            bind((c.__dict__["_params"][wname], cw[l].detach().view(-1, 1)))
            bind((c.__dict__["_params"][bname], cb[l].detach().view(-1, 1)))
    eng.beta1_power, eng.beta2_power, eng.step_count = np.float32(adam.beta1_power), np.float32(adam.beta2_power), int(adam.global_step)
    loss_cell = next((c for c in _cells(cell) if any(x is owner for x in c.cells())), None)
    low.kind = "deep_cross"
    low.verify = lambda ids, wts, label=None: _verify(low, eng, owner, loss_cell, ids, wts, label, False)
    return low


# ---- DeepFM ----------------------------------------------------------------------------------------------------------------------
def _lower_deepfm(cell, plan):
    """models/deepfm/src/deepfm.py: a model cell that holds its two tables as plain Parameters ([V, D] and [V, 1], looked up with
    Gather: dense gradients), a chain of DenseLayers F * D -> ... -> 1, one nn.Adam over everything, the L2 term over both whole
    tables in the loss cell."""
    from .deepfm import DeepFMConfig, DeepFMEngine
    opts = [(c, _opt_kind(c)) for c in cell.cells() if _opt_kind(c)]
    if len(opts) != 1 or opts[0][1] != "Adam":
        return None
    adam = opts[0][0]
    owner = None
    for c in _cells(cell):
        tabs = [q for q in c.__dict__.get("_params", {}).values() if isinstance(q, torch.Tensor) and q.dim() == 2]
        if (len(tabs) == 2 and tabs[0].shape[0] == tabs[1].shape[0] and sorted(int(t.shape[1]) for t in tabs)[0] == 1
                and sum(_is_dense_layer(x) for x in c.cells()) >= 2 and not any(_is_cross_layer(x) and not _is_dense_layer(x) for x in c.cells())):
            owner = c
            break
    if owner is None:
        return None
    if plan.shard:
        raise LoweringRefused("distributed train cell (DeepFM under DistributedGradReducer): DeepFMEngine is a one-rank engine -- the "
                              "step runs primitive by primitive with its all-reduce")
    tabs = [q for q in owner.__dict__["_params"].values() if isinstance(q, torch.Tensor) and q.dim() == 2]
    lin_t, emb_t = sorted(tabs, key=lambda t: int(t.shape[1]))
    V, D = int(emb_t.shape[0]), int(emb_t.shape[1])
    if D == 1:
        raise LoweringRefused("the embedding table has dimension 1: cannot tell it from the linear table")
    B = int(getattr(owner, "batch_size", 0))
    F = int(getattr(owner, "field_size", 0) or getattr(owner, "data_field_size", 0))
    if B <= 0 or F <= 0:
        raise LoweringRefused("the model does not state batch_size / field_size")
    layers = _chain([x for x in owner.cells() if _is_dense_layer(x)], F * D)
    if layers is None or layers[-1].weight.shape[1] != 1 or len(layers) < 2:
        raise LoweringRefused("the DenseLayer cells do not form a chain F * D -> ... -> 1")
    for l in layers:
        d = getattr(l, "dropout", None)
        if d is not None and float(getattr(d, "p", 1.0 - float(getattr(d, "keep_prob", 1.0)))) != 0.0:
            raise LoweringRefused("DeepFM is lowered without Dropout (the reference's DenseLayer holds Dropout(p=0.0), deepfm.py:114)")
        if str(type(getattr(l, "act_func", None)).__name__) not in ("ReLU", "NoneType") and getattr(l, "use_act", True):
            raise LoweringRefused("DeepFM is lowered with ReLU DenseLayers only")
    if getattr(layers[-1], "use_act", False):
        raise LoweringRefused("the output DenseLayer must not have an activation")
    half = bool(getattr(layers[0], "convert_dtype", False))
    if any(bool(getattr(l, "convert_dtype", False)) != half for l in layers):
        raise LoweringRefused("mixed convert_dtype settings")
    if getattr(adam, "weight_decay", 0.0) or getattr(adam, "use_nesterov", False) or abs(float(adam.beta1) - 0.9) > 1e-6 or abs(float(adam.beta2) - 0.999) > 1e-6:
        raise LoweringRefused("non-default Adam settings are not lowered")
    want = [lin_t, emb_t] + [p for l in layers for p in (l.weight, l.bias)]
    if len(list(adam.parameters)) != len(want) or not all(any(p is q for q in adam.parameters) for p in want):
        raise LoweringRefused("the optimizer does not own exactly the two tables and the DenseLayers")
    loss_cell = next((c for c in _cells(cell) if any(x is owner for x in c.cells()) and hasattr(c, "l2_coef")), None)
    if loss_cell is None:
        raise LoweringRefused("no loss cell with an l2_coef around the model")
    if float(getattr(cell, "sens", adam.loss_scale)) != float(adam.loss_scale):
        raise LoweringRefused("sens and the optimizer's loss_scale differ")
    if emb_t.device.type != "cuda":
        raise LoweringRefused("parameters are not on an MI355X")
    if int(adam.global_step):
        raise LoweringRefused("DeepFM is lowered from a fresh optimizer only")
    cfg = DeepFMConfig(data_vocab_size=V, data_emb_dim=D, data_field_size=F, batch_size=B, deep_layer_dims=[int(l.weight.shape[1]) for l in layers[:-1]],
                       l2_coef=float(loss_cell.l2_coef), learning_rate=adam.get_lr(), epsilon=float(adam.eps), loss_scale=float(adam.loss_scale),
                       mlp_dtype="fp16" if half else "fp32")
    eng = DeepFMEngine(cfg, emb_t.device)
    if not (eng._mfma or getattr(eng, "_f32net", False)):
        raise LoweringRefused("these DeepFM shapes have no hand-written path")
    cls = type(emb_t).__mro__[1] if type(emb_t).__name__ == "Parameter" else type(emb_t)
    low = LoweredStep(eng, cls, 1, optimizers=(adam,))
    bind = low._binds.append
    with torch.no_grad():
        eng.V_l2.copy_(_raw(emb_t))
        eng.W_l2.copy_(_raw(lin_t))
        eng.load_dense_parameters([_raw(l.weight) for l in layers], [_raw(l.bias) for l in layers])
        for (prefix, pid), st in adam.__dict__.get("_state", {}).items():
            which = 0 if prefix == "moment1" else 1
            tgt = None
            if pid == id(emb_t):
                tgt = eng.state["V"][which]
            elif pid == id(lin_t):
                tgt = eng.state["W"][which]
            else:
                for i, l in enumerate(layers):
                    if pid == id(l.weight):
                        tgt = (eng.dense_m if which == 0 else eng.dense_v)[_off(eng, 2 * i)].view(l.weight.shape)
                    elif pid == id(l.bias):
                        tgt = (eng.dense_m if which == 0 else eng.dense_v)[_off(eng, 2 * i + 1)].view(l.bias.shape)
            if tgt is not None:
                tgt.copy_(_raw(st))
                bind((st, tgt))
    bind((emb_t, eng.V_l2))
    bind((lin_t, eng.W_l2))
    for i, l in enumerate(layers):
        bind((l.weight, eng.dense[2 * i]))
        bind((l.bias, eng.dense[2 * i + 1]))
    l2c = float(loss_cell.l2_coef)

    def l2_term():        # the loss cell's l2_coef / 2 * (sum V^2 + sum w^2) over the whole tables (deepfm.py:252-259), for the check
        return l2c * 0.5 * float((eng.V_l2.double() ** 2).sum() + (eng.W_l2.double() ** 2).sum())
    low.kind = "deepfm"
    low.verify = lambda ids, wts, label=None: _verify(low, eng, owner, loss_cell, ids, wts, label, half, loss_extra=l2_term)
    return low
