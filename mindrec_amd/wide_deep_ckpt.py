"""Checkpoint / resume and shard merging for WideDeepEngine."""
import numpy as np
import torch

from . import ops

# ---- checkpoint / resume ---------------------------------------------------
# The reference saves through ModelCheckpoint (train_and_eval.py:96-97) and, for sliced tables, merges
# the per-rank slices at load time (eval.py:86-107: build_searched_strategy + merge_sliced_parameter).
# Here every rank saves its own row shard; merge_shards() interleaves them back (owner = id mod n).
def _engine_state(eng):
    return {
        "meta": {"rank": eng.rank, "world": eng.world, "vocab_size": eng.cfg.vocab_size, "emb_dim": eng.cfg.emb_dim,
                 "field_size": eng.cfg.field_size, "step_count": eng.step_count,
                 "beta1_power": float(eng.beta1_power), "beta2_power": float(eng.beta2_power),
                 # which optimizer owns wide_b: under "ftrl" its m / v words of the dense buffer hold FTRL's accum / linear, under
                 # "adam" Adam's moments -- the same bytes mean different things
                 "wide_b_optimizer": getattr(eng.cfg, "wide_b_optimizer", "adam")},
        "tables": {"deep": eng.deep, "deep_m": eng.deep_m, "deep_v": eng.deep_v, "wide": eng.wide,
                   "wide_accum": eng.wide_accum, "wide_linear": eng.wide_linear},
        "dense": {"dense": eng.dense_flat.detach(), "dense_m": eng.dense_m, "dense_v": eng.dense_v, "wide_b": eng.wide_b},
    }


def save_checkpoint(eng, path):
    """Writes this rank's shard (tables + optimizer state + dense parameters) to `path` (torch.save)."""
    if getattr(eng, "_sharded", False):
        eng.check_shard_overflow()          # a checkpoint of steps that dropped positions must not be written (every rank raises alike)
    eng.check_cache()
    st = _engine_state(eng)
    if eng.hb is not None:
        # host-backed tables: write the device cache back, then save the pinned host store itself (every row of the shard;
        # hash tables: the rows of the live keys, in key-export order, like the resident hash tables below)
        eng.hb.flush()
        torch.cuda.synchronize(eng.device)
        dense = {k: v.detach().cpu().contiguous() for k, v in st["dense"].items()}
        names = ("deep", "deep_m", "deep_v", "wide", "wide_accum", "wide_linear")
        if eng.hb.hashed:
            keys, rows = eng.hb.export_hashed()
            tables = {k: rows[:, a:b].contiguous() for k, (a, b) in eng.hb.col_ranges.items() if k in names}
            torch.save({"meta": dict(st["meta"], dynamic_embedding=True, host_cache=True), "keys": keys, "tables": tables, "dense": dense}, path)
        else:
            full = eng.hb.full_table()          # never-touched rows are generated on demand (their default values)
            tables = {k: full[:, a:b].contiguous() for k, (a, b) in eng.hb.col_ranges.items() if k in names}
            torch.save({"meta": dict(st["meta"], host_cache=True), "tables": tables, "dense": dense}, path)
        return
    if eng.index is not None:
        # hash tables: the live keys and, per table, the rows of those keys in key-export order (the analogue of
        # MapParameter.export_data: keys + values; row numbers are not part of the state)
        keys, rows = eng.index.export()
        r = rows.long()
        tables = {k: v.detach()[r].cpu().contiguous() for k, v in st["tables"].items()}
        out = {"meta": dict(st["meta"], dynamic_embedding=True), "keys": keys.cpu(), "tables": tables,
               "dense": {k: v.detach().cpu().contiguous() for k, v in st["dense"].items()}}
        torch.save(out, path)
        return
    out = {"meta": st["meta"], "tables": {k: v.detach().cpu().contiguous() for k, v in st["tables"].items()},
           "dense": {k: v.detach().cpu().contiguous() for k, v in st["dense"].items()}}
    torch.save(out, path)


def load_checkpoint(eng, path):
    """Restores a shard saved by save_checkpoint into an engine of the same geometry."""
    ck = torch.load(path, map_location="cpu")
    m = ck["meta"]
    for k in ("rank", "world", "vocab_size", "emb_dim", "field_size"):
        have = {"rank": eng.rank, "world": eng.world}.get(k, getattr(eng.cfg, k, None))
        if m[k] != have:
            raise ValueError(f"checkpoint {path}: {k} = {m[k]} but the engine has {have}")
    st = _engine_state(eng)
    # checkpoints written before the marker existed (rounds 1-3) kept wide_b under Adam
    saved_wb, have_wb = m.get("wide_b_optimizer", "adam"), getattr(eng.cfg, "wide_b_optimizer", "adam")
    if eng.hb is not None:
        raise NotImplementedError("load_checkpoint restores into resident tables; a checkpoint written by a host-cached engine loads "
                                  "into a resident engine of the same geometry")
    if bool(m.get("dynamic_embedding", False)) != (eng.index is not None):
        raise ValueError(f"checkpoint {path}: dynamic_embedding does not match the engine")
    with torch.no_grad():
        if eng.index is not None:
            # re-insert the keys (this engine numbers the rows its own way), then put each key's rows in place
            if len(eng.index):
                raise ValueError("load_checkpoint needs a fresh dynamic_embedding engine (its key index is not empty)")
            keys = ck["keys"].to(eng.device).contiguous()
            rows, _ = eng.index.find_or_insert(keys, insert=True)
            for k, dst in st["tables"].items():
                ops.scatter_rows_(dst, rows, ck["tables"][k].to(eng.device).contiguous())
            for k, dst in st["dense"].items():
                dst.copy_(ck["dense"][k].to(dst.device))
        else:
            for grp in ("tables", "dense"):
                for k, dst in st[grp].items():
                    dst.copy_(ck[grp][k].to(dst.device))
        if saved_wb != have_wb and hasattr(eng, "_wb_off"):
            # the optimizer that owns wide_b changed between save and load: its two state words cannot be reinterpreted -- the new
            # owner starts from its initial state (FTRL: accum = initial_accum, linear = 0; Adam: zero moments); the weight is kept
            import warnings
            warnings.warn(f"checkpoint {path}: wide_b was trained under {saved_wb!r}, this engine updates it with {have_wb!r}; "
                          f"its optimizer state is reset")
            eng.dense_m[eng._wb_off] = eng.cfg.ftrl_initial_accum if have_wb == "ftrl" else 0.0
            eng.dense_v[eng._wb_off] = 0.0
        if eng.dense16 is not None:
            eng.dense16_flat.copy_(eng.dense_flat.detach())
            eng._refresh_tail()
    eng.step_count = m["step_count"]
    eng.beta1_power, eng.beta2_power = np.float32(m["beta1_power"]), np.float32(m["beta2_power"])


def merge_shards(paths):
    """Interleaves the row shards of all ranks back into whole tables (row id = local * world + rank):
    the analogue of merge_sliced_parameter (eval.py:101-105).  Returns {name: [V, D] tensor} + the dense part
    of rank 0 (replicated by data parallelism)."""
    cks = sorted((torch.load(p, map_location="cpu") for p in paths), key=lambda c: c["meta"]["rank"])
    world = cks[0]["meta"]["world"]
    if [c["meta"]["rank"] for c in cks] != list(range(world)):
        raise ValueError("merge_shards needs exactly one checkpoint per rank")
    V = cks[0]["meta"]["vocab_size"]
    out = {}
    if any(c["meta"].get("dynamic_embedding") for c in cks):
        # hash tables: owner = hash(key) mod world and every rank exports its own keys with their rows, in its own order and
        # number -- the whole table is the concatenation of the ranks' (keys, rows) pairs (MapParameter.export_data's form)
        if not all(c["meta"].get("dynamic_embedding") for c in cks):
            raise ValueError("merge_shards: hash-table and dense-table checkpoints cannot be mixed")
        out["keys"] = torch.cat([c["keys"] for c in cks])
        if out["keys"].unique().numel() != out["keys"].numel():
            raise ValueError("merge_shards: a key lives on more than one rank")
        for name in cks[0]["tables"]:
            out[name] = torch.cat([c["tables"][name] for c in cks])
        out.update(cks[0]["dense"])
        return out
    for name, t0 in cks[0]["tables"].items():
        full = torch.empty((V, t0.shape[1]), dtype=t0.dtype)
        for c in cks:
            full[c["meta"]["rank"]::world] = c["tables"][name]
        out[name] = full
    out.update(cks[0]["dense"])
    return out


# ---- synthetic Criteo-shaped batches ------------------------------------------
