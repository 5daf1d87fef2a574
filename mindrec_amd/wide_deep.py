"""Wide&Deep training step on the MI355X embedding path.

Mirrors, op for op, what one call of TrainStepWrap.construct does in the reference
(models/wide_deep/src/wide_and_deep.py:472-492) in its sparse configuration -- the one BASELINE
config 2/4 names (SURVEY.md 8(a) "Which mode is the W&D hot path"):

  WideDeepModel.construct      :293-316  two lookups sharing one id tensor, mask multiply, wide
                                         reduce-sum + bias, 5-layer MLP, wide + deep
  NetWithLossClass.construct   :349-362  sigmoid cross-entropy, mean; no L2 term when sparse
  TrainStepWrap.__init__       :415-433  LazyAdam(lr 3.5e-4, eps 1e-8) on non-"wide" params,
                                         FTRL(lr 5e-2, l1 = l2 = 1e-8, initial_accum 1.0) on "wide" params,
                                         both with loss_scale = sens = 1024
  TrainStepWrap.construct      :472-492  backward seeded with sens, optional grad reducer, two applies

The reference writes the forward three times and lets MindSpore's graph compiler merge them; here
it simply runs once.  Embedding lookups, the id dedup, both sparse applies AND the mixed-precision
MLP (hand-written MFMA GEMMs, csrc/mrec_dense.hip) are libmrec_hip.so kernels; fp32 master weights are
kept in one flat buffer so the dense LazyAdam (= Adam on dense gradients) is a single kernel launch.

Multi-GPU ("hybrid parallel", README.md:140-144): both tables are row-sharded, owner = id mod n,
local row = id div n.  Per step: bucket ids by owner -> RCCL all-to-all ids -> local gather ->
all-to-all rows back; backward: all-to-all row-gradients to the owners -> local dedup + sparse
apply (optimizer state is shard-local, the table gradient is never all-reduced).  The MLP stays
data-parallel with one flat all-reduce(mean) (gradients_mean=True,
train_and_eval_distribute.py:135-138).
"""
import contextlib
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .wide_deep_ckpt import load_checkpoint, merge_shards, save_checkpoint  # noqa: F401  (re-exported)
from .wide_deep_mlp import DenseNetMixin, _WideProd
from .wide_deep_shard import ShardCapacityError, ShardStepMixin, grow_shard_capacity  # noqa: F401  (re-exported)


@dataclass
class WideDeepConfig:
    """Field names follow models/wide_deep/default_config.yaml:14-44."""
    vocab_size: int = 200_000_000
    emb_dim: int = 80
    field_size: int = 39
    batch_size: int = 16384
    deep_layer_dim: List[int] = field(default_factory=lambda: [1024, 512, 256, 128])
    sens: float = 1024.0                 # TrainStepWrap(sens=1024.0), wide_and_deep.py:390
    adam_lr: float = 3.5e-4              # :420
    adam_eps: float = 1e-8
    ftrl_lr: float = 5e-2                # :423-430
    ftrl_l1: float = 1e-8
    ftrl_l2: float = 1e-8
    ftrl_initial_accum: float = 1.0
    init_sigma: float = 0.01             # emb_init / weight_bias_init 'normal' [EXT: N(0, 0.01)]
    seed: int = 1000                     # set_seed(1000), train_and_eval_distribute.py:72
    mlp_dtype: str = "fp16"              # the reference's own mixed precision (use_mixed_precision: Cast to float16, wide_and_deep.py:119-128,
                                         # default_config.yaml:28); "bf16" runs the same kernels, "fp32" the library GEMMs
    id_dtype: str = "int32"              # dataset contract, process_data.py:204-206
    # HBM layout: one allocation per table with the optimizer state beside the weights --
    # deep rows are [p(D) | w accum linear pad | m(D) | v(D)] (1 KB at D = 80): the sparse apply touches one contiguous run per
    # row instead of three rows 64 GB apart, and the wide record rides the deep row's lines; the API still sees
    # p, m, v, w, ... as separate (strided) [V, D] / [V, 1] tensors.  False = separate arrays.
    fused_state: bool = True
    fold_wide: bool = True               # fused rows, 16-bit MLP: the wide lookup rides the deep gather and the wide FTRL
                                         # the deep LazyAdam apply (one row visit each); False keeps the separate wide kernels
    sparse: bool = True              # False: the reference's DEFAULT mode (default_config.yaml:36, the CPU-runnable configs[0]): the
                                     # embedding gradients are dense [V, D] tensors, nn.Adam / nn.FTRL visit every row every step
                                     # and the deep loss carries l2_coef * sum(E^2) / 2 (wide_and_deep.py:337-339,356-360,434-445)
    l2_coef: float = 8e-5            # default_config.yaml:43
    dropout_flag: bool = False       # default_config.yaml:27 (benchmarks/wide_deep/default_config.yaml:15 switches it on): Dropout on the
                                     # INPUT of every DenseLayer while training (wide_and_deep.py:117-118)
    dropout_keep_prob: float = 0.5   # DenseLayer's own default (:85) -- WideDeepModel never forwards config.keep_prob (:164-205)
    dynamic_embedding: bool = False  # both tables are hash tables keyed by the raw ids (train_and_eval.py --dynamic_embedding=True,
                                     # wide_and_deep.py:271-274): rows are created on first sight with their default values
    hash_capacity: int = 1 << 22     # rows reserved in HBM for each hash table (dynamic_embedding)
    host_cache_rows: int = 0         # > 0: both tables live in pinned host DRAM behind a device cache of this many rows (the
                                     # reference's vocab_cache_size, wide_and_deep.py:215-265)
    shard_capacity_factor: float = 1.25   # row shards: request slots a rank reserves per owner = ceil(factor * ids / ranks) -- every
                                          # message of a sharded step has a static shape (mindrec_amd/wide_deep_shard.py)
    shard_unique_factor: float = 0.0      # > 0: row shards exchange the batch's UNIQUE ids (one fp32 row / one summed gradient row per
                                          # unique id) instead of one 16-bit row per position; the value is the unique ids a batch may
                                          # hold per position (capacity = shard_capacity_factor * this * ids / ranks).  Criteo-like ids:
                                          # 0.2-0.3 unique per position (mindrec_amd/wide_deep_shard.py)
    graphs: str = "step"           # what replays as HIP graphs: "step" the whole step (sinks of steps: train_steps), "front" everything in
                                   # front of the optimizers, "mlp" the dense net only, "none" kernel by kernel
    fused_tail: bool = True        # the last two hidden layers, the output head and their input-gradient bprops as one launch
    fp32_matmul: str = "x3"        # the MatMuls of an fp32 net (amp "none"): "x3" three-part bf16 operands on the 16-bit matrix
                                   # instruction (fp32-class accuracy, csrc/mrec_gemm_x3.hip), "exact" the fp32-input matrix instruction
    wide_b_optimizer: str = "ftrl"  # which optimizer owns the wide bias.  "ftrl": what the reference's code does on MindSpore --
                                    # TrainStepWrap sorts by `"wide" in params.name` (wide_and_deep.py:407-411) and the Parameter held in
                                    # the attribute `wide_b` (:161-163) is renamed "<prefix>.wide_b" when its cell is assigned to a
                                    # parent (Cell.update_parameters_name uses the attribute path [EXT]; pinned by tests/golden/
                                    # ref_wd_*.npz, which the reference's own TrainStepWrap produced over compat/mindspore).  "adam":
                                    # the literal reading of the constructor name "Wide_b" (rounds 2-3)
    const_columns: bool = True     # fields whose id is the same in every sample of a batch (the reference's Criteo pipeline gives each
                                   # of the 13 dense features ONE id, process_data.py:138-147) are summed sample by sample by the folded
                                   # one-GPU apply instead of through the inverted index (WideDeepEngine._plan; include/mrec.h,
                                   # mrec_const_cols_detect): same products, another fixed order of additions for those rows -- False
                                   # where two engines must agree bit for bit across code paths (hash tables, row shards)


_GRAPH_LEVEL = {"none": 0, "mlp": 1, "front": 2, "step": 3}
class _DirectComm:
    """The engine's collectives: torch.distributed on the tensors as they are -- device tensors under backend "nccl"
    (= RCCL over xGMI, the product path), host tensors under gloo (the CPU logic tests).  Test harnesses that put
    several ranks on ONE GPU inject a comm that stages device tensors through the host (tests/_staged_comm.py)."""

    def __init__(self, group=None):
        self.group = group
        self._stream = None        # eager asynchronous reductions run here (see all_reduce)

    @property
    def capturable(self):
        """RCCL collectives on device tensors can be captured into a HIP graph (tools/probes/rccl_graph_probe.py); gloo cannot."""
        return dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def all_gather_rows(self, t):
        """[rows, ...] of every rank, concatenated in rank order (row counts may differ): what puts the shards of a table back
        together for a checkpoint (mindrec_amd/lowering.py).  Not on the step's path."""
        n = dist.get_world_size(self.group)
        cnt = torch.tensor([t.shape[0]], dtype=torch.int64)
        cnts = [torch.zeros_like(cnt) for _ in range(n)]
        if dist.get_backend(self.group) == "nccl":
            cnt = cnt.to(t.device)
            cnts = [c.to(t.device) for c in cnts]
        dist.all_gather(cnts, cnt, group=self.group)
        cnts = [int(c) for c in cnts]
        cap = max(cnts)
        staged = self._staged(t)
        src = t.contiguous().cpu() if staged else t.contiguous()
        pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        outs = [torch.empty_like(pad) for _ in range(n)]
        dist.all_gather(outs, pad, group=self.group)
        full = torch.cat([o[:c] for o, c in zip(outs, cnts)])
        return (full.to(t.device) if staged else full), cnts

    def _staged(self, t):
        """A gloo group handed device tensors (several ranks sharing one GPU, a box without RCCL): gloo moves host memory, so the
        message is staged through the host.  Same results; not capturable; RCCL never takes this branch."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self._staged(out):
            i8 = inp.contiguous().view(inp.shape[0], -1).view(torch.uint8).cpu()         # (gloo has no 16-bit floats: ship bytes)
            o8 = torch.empty((out.shape[0], out[0].numel() * out.element_size()), dtype=torch.uint8)
            dist.all_to_all_single(o8, i8, out_splits, in_splits, group=self.group)
            out.copy_(o8.view(out.dtype).view(out.shape))
            return
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def all_to_all_lists(self, outs, ins):
        """outs[r] <- what rank r sends to this rank, ins[r] -> rank r; entries may differ in length (zero-length: nothing moves).
        RCCL: one grouped all-to-all (capturable).  A backend without the list form (gloo: the CPU logic tests) gets the same
        exchange as point-to-point sends and receives -- which is what RCCL's all-to-all is made of."""
        if dist.get_backend(self.group) == "nccl":
            dist.all_to_all(outs, ins, group=self.group)
            return
        if any(self._staged(t) for t in list(outs) + list(ins)):
            def host(t, fill):          # the same rows as bytes in host memory
                h = torch.empty(t.shape[:-1] + (t.shape[-1] * t.element_size(),), dtype=torch.uint8)
                if fill and t.numel():
                    h.copy_(t.contiguous().view(torch.uint8))
                return h
            houts, hins = [host(t, False) for t in outs], [host(t, True) for t in ins]
            self.all_to_all_lists(houts, hins)
            for o, h in zip(outs, houts):
                if o.numel():
                    o.copy_(h.view(o.dtype).view(o.shape))
            return
        ops_ = []
        for r, (o, i) in enumerate(zip(outs, ins)):
            if i.numel():
                ops_.append(dist.P2POp(dist.isend, i, r, self.group))
            if o.numel():
                ops_.append(dist.P2POp(dist.irecv, o, r, self.group))
        if ops_:
            for w in dist.batch_isend_irecv(ops_):
                w.wait()

    def all_reduce(self, t, async_op=False):
        """Sum over ranks.  async_op=True returns a handle with wait() (device tensors only): the reduction proceeds beside the
        calling stream, which keeps issuing kernels until it waits.

        Inside a capture that is the group's own asynchronous op (its internal stream joins the capture).  OUTSIDE one it must not
        be: an eager asynchronous op leaves its end event on the group's internal stream, torch's watchdog thread queries that
        event until the work is retired (every ~100 ms), and on HIP a query fails -- and takes the process down from the
        watchdog thread -- while the event's stream is capturing, which the internal stream is as soon as a captured step issues
        its first asynchronous op (tools/probes/rccl_capture_race_probe.py reproduces it; it was the 'flaky'
        hipErrorCapturedEvent of rounds 3-4).  Eager reductions therefore run as the synchronous op on a stream of our own,
        which never captures; a synchronous op's event lives on the stream it was issued on."""
        if self._staged(t):
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
            return None
        if async_op and t.is_cuda:
            if torch.cuda.is_current_stream_capturing():
                return dist.all_reduce(t, group=self.group, async_op=True)
            if self._stream is None:
                self._stream = torch.cuda.Stream(t.device)
            self._stream.wait_stream(torch.cuda.current_stream(t.device))
            with torch.cuda.stream(self._stream):
                dist.all_reduce(t, group=self.group)
                return _StreamWork(self._stream.record_event())
        dist.all_reduce(t, group=self.group)
        return None


class _StreamWork:
    """wait(): the current stream waits for the reduction issued on the communicator's side stream (no host block)."""

    def __init__(self, event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)


class WideDeepEngine(DenseNetMixin, ShardStepMixin):
    """State + one training step.  rank/world describe the row sharding; world == 1 is one GPU."""
    _kernels = ops               # the op set (tests/ subclass the engine with a CPU stand-in to run the multi-rank host logic under gloo)
    _allow_cpu = False           # the product has no CPU path

    def __init__(self, cfg: WideDeepConfig, device, rank=0, world=1, group=None, comm=None, shard_protocol=False,
                 tables_from=None):
        """comm: collectives provider (default: torch.distributed as is, see _DirectComm).
        shard_protocol: run the row-shard protocol (routing kernels + collectives) even when world == 1 -- every
        collective then talks to itself, which executes the RCCL code path on a single GPU.
        tables_from: another engine of the same vocabulary / dim / layout whose fused-row tables this one trains on instead of
        allocating its own (a second workload over the same 205-GB table: bench.py's Criteo-like line)."""
        self.cfg, self.device, self.rank, self.world, self.group = cfg, torch.device(device), rank, world, group
        self._sharded = bool(world > 1 or shard_protocol)
        self._fold_wide = False       # set below: one GPU, fused rows, 16-bit MLP -> the wide branch rides the deep kernels
        self.k = self._kernels
        kernels = None if self.k is ops else self.k
        self.comm = comm if comm is not None else _DirectComm(group)
        self._gpu = self.device.type == "cuda"
        if not self._gpu and not self._allow_cpu:
            raise RuntimeError("WideDeepEngine runs on an MI355X (no CPU fallback)")
        if cfg.graphs not in _GRAPH_LEVEL:
            raise ValueError(f"graphs must be one of {sorted(_GRAPH_LEVEL)}")
        self._graph_level = _GRAPH_LEVEL[cfg.graphs]      # lowered when a capture is refused
        V, D = cfg.vocab_size, cfg.emb_dim
        self.local_rows = (V - rank + world - 1) // world          # rows r with r*world + rank < V
        self.index = None
        self.hb = None
        if cfg.host_cache_rows > 0:
            if kernels is not None:
                raise ValueError("host_cache_rows needs the HIP kernels")
            self.local_rows = int(cfg.host_cache_rows)
        if cfg.dynamic_embedding:
            # HashEmbeddingLookup x2 with all defaults (wide_and_deep.py:271-274; embedding.py:88-93): a device
            # key -> row index over `hash_capacity` rows; the row tables below are addressed by row number, so
            # every kernel downstream of the index probe is the one the dense-table mode uses.  Row-sharded: the raw keys
            # travel to owner = hash(key) mod n, whose own index translates them (BASELINE configs[4]).
            if kernels is not None:
                raise ValueError("dynamic_embedding needs the device key index")
            if cfg.host_cache_rows == 0:
                self.local_rows = int(cfg.hash_capacity)
                self.index = ops.KeyIndex(self.local_rows, self.device)
        dev = self.device
        self._amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[cfg.mlp_dtype]
        dims = [cfg.field_size * D] + list(cfg.deep_layer_dim) + [1]
        nl = len(dims) - 1
        self.dims = dims
        # The mixed-precision dense net runs on the hand-written MFMA kernels (csrc/mrec_dense.hip); widths that are not
        # multiples of 8 (rows not 16-byte aligned), an fp32 net and the CPU stand-in take the autograd path below.
        self._mfma = bool(self._gpu and kernels is None and self._amp is not None and D % 4 == 0 and D <= 256 and self.mfma_net_ok(dims))
        if not cfg.sparse and (self._sharded or cfg.dynamic_embedding or cfg.host_cache_rows > 0):
            raise ValueError("sparse=False (dense gradients over the whole table) runs on one GPU with resident dense tables")
        self._fold_wide = bool(cfg.sparse and cfg.fold_wide and self._mfma and not self._sharded and cfg.fused_state and cfg.host_cache_rows == 0 and D <= 252
                               and self.k.head_supported(dims[nl - 1]))
        with (torch.cuda.device(dev) if self._gpu else contextlib.nullcontext()):
            # deep table + Adam moments, wide table + FTRL accumulators: plain row-major fp32 in HBM
            R = self.local_rows
            if cfg.host_cache_rows > 0:
                # one row = [p | m | v | w, accum, linear, pad]: the deep LazyAdam row and the wide FTRL record of an id
                # travel between the host and the cache together
                # (hash tables: `hash_capacity` host rows behind a second key index; a row shard of a dense table: this
                # rank's rows, default values keyed by the global id r * world + rank)
                from .feature_cache import HostBackedTable
                host_rows = int(cfg.hash_capacity) if cfg.dynamic_embedding else (V - rank + world - 1) // world
                self.hb = HostBackedTable(host_rows, D, R, dev, hashed=cfg.dynamic_embedding, key_scale=world, key_offset=rank, columns=[
                    ("deep", D, ("normal", cfg.seed, cfg.init_sigma)), ("deep_m", D, ("fill", 0.0)), ("deep_v", D, ("fill", 0.0)),
                    ("wide", 1, ("normal", cfg.seed + 1, cfg.init_sigma)), ("wide_accum", 1, ("fill", cfg.ftrl_initial_accum)),
                    ("wide_linear", 1, ("fill", 0.0)), ("pad", 1, ("fill", 0.0))])
                for name in ("deep", "deep_m", "deep_v", "wide", "wide_accum", "wide_linear"):
                    setattr(self, name, self.hb.cols[name])
            elif cfg.fused_state and cfg.sparse:
                # ONE row per id: [p(D) | w accum linear pad | m(D) | v(D) | pad], padded to a multiple of 128 bytes (3D + 4 = 244
                # floats -> 256 floats = 1 KB at D = 80: every row is exactly 8 lines, p + w exactly 3; with 976-byte rows a row
                # straddles 8.5 lines on average and the lookup 3.5).  The wide weight sits right behind the deep weights (the
                # line the deep lookup fetches anyway) and its FTRL words inside the run the deep LazyAdam reads and writes anyway:
                # both lookups and both sparse applies visit a row once.  The API still sees p, m, v [V, D] and w, accum,
                # linear [V, 1] as (strided) tensors.
                ldrow = -(-(3 * D + 4) // 32) * 32
                if tables_from is not None:
                    if getattr(tables_from, "deep_state", None) is None or tuple(tables_from.deep_state.shape) != (R, ldrow):
                        raise ValueError("tables_from: an engine with fused rows of the same vocabulary and dim")
                    self.deep_state = tables_from.deep_state
                else:
                    self.deep_state = torch.zeros((R, ldrow), dtype=torch.float32, device=dev)
                self.deep, self.deep_m, self.deep_v = (self.deep_state[:, :D], self.deep_state[:, D + 4:2 * D + 4],
                                                       self.deep_state[:, 2 * D + 4:3 * D + 4])
                self.wide, self.wide_accum, self.wide_linear = (self.deep_state[:, D:D + 1], self.deep_state[:, D + 1:D + 2],
                                                                self.deep_state[:, D + 2:D + 3])
            else:
                self.deep = torch.empty((R, D), dtype=torch.float32, device=dev)
                self.deep_m, self.deep_v = torch.empty_like(self.deep), torch.empty_like(self.deep)
                self.wide = torch.empty((R, 1), dtype=torch.float32, device=dev)
                self.wide_accum, self.wide_linear = torch.empty_like(self.wide), torch.empty_like(self.wide)
            if tables_from is not None and getattr(self, "deep_state", None) is not tables_from.deep_state:
                raise ValueError("tables_from needs the fused-row layout (fused_state, sparse, no host cache)")
            if not cfg.dynamic_embedding and self.hb is None and tables_from is None:
                self.k.fill_normal_(self.deep, cfg.seed, cfg.init_sigma, row0=rank, row_stride=world)
                self.deep_m.zero_()
                self.deep_v.zero_()
                self.k.fill_normal_(self.wide, cfg.seed + 1, cfg.init_sigma, row0=rank, row_stride=world)
                self.wide_accum.fill_(cfg.ftrl_initial_accum)
                self.wide_linear.zero_()
            # (dynamic_embedding: rows get exactly these values when their key is first seen, _translate_keys)
            # "Wide_b" lives in the dense net's flat buffer: TrainStepWrap sorts parameters by the case-sensitive test
            # `"wide" in params.name` (wide_and_deep.py:407-411) and the bias is named "Wide_b" (:161-163), so it belongs to the
            # DEEP optimizer (Adam), not to FTRL; in that buffer it is updated by the same dense-Adam launch.
            self._init_dense_net(dims, cfg.seed + 2, cfg.init_sigma, cfg.sens, extra_seed=cfg.seed + 3,
                                 fused_tail=bool(cfg.fused_tail and cfg.field_size <= 64))      # (<= 64 wide products per sample)
            self.wide_b, self.wide_b_grad = self.extra_p, self.extra_g
            if cfg.wide_b_optimizer not in ("ftrl", "adam"):
                raise ValueError("wide_b_optimizer must be 'ftrl' or 'adam'")
            # wide_b under FTRL stays an element of the dense buffer (the head kernel writes its gradient there): the dense-Adam
            # launch treats that one element with FTRL, its m word as accum, its v word as linear (mrec_ftrl1_t)
            self._wb_ftrl = None
            if cfg.wide_b_optimizer == "ftrl":
                self._wb_ftrl = (self._wb_off, cfg.ftrl_lr, cfg.ftrl_l1, cfg.ftrl_l2, -0.5)
                self.dense_m[self._wb_off] = cfg.ftrl_initial_accum
        self._hashed = bool(cfg.dynamic_embedding)
        self._fused_rows = bool(cfg.fused_state and cfg.sparse and cfg.host_cache_rows == 0 and self._gpu)   # [p | w ... | m | v] rows
        if self._sharded:
            self._shard_init()
        self.beta1, self.beta2 = np.float32(0.9), np.float32(0.999)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)
        self.step_count = 0
        self.timers = None            # optional dict name -> list[(start_event, stop_event)]
        # the step's Unique + inverted index runs on a side stream, under the MLP
        # (default priority: a high-priority side stream was measured at 1.52 ms/step instead of 0.88)
        self._side = torch.cuda.Stream(device=self.device) if self._gpu else None
        import os
        self._fuse_finish = True      # the apply's finishing pass inside the dense Adam launch (see _choose_finish)
        self._const_cols = bool(getattr(cfg, "const_columns", True)) and os.environ.get("MREC_CONST_COLS", "1") != "0"      # (see _plan)
        self._col_bad, self._cc = None, None
        self._hot_seen = False          # the stream's first batches held hot columns (looked at in the eager steps, see _plan)
        # a field's DOMINANT id (not in every sample, but in at least max(MREC_HOT_MIN, B / 8) of them) can take the same path; 0 (the
        # default): constant columns only -- measured on the bench's Zipf ids, whose 26 categorical fields each fold their rare ids into
        # one id (~40 % of the field): k_apply_main 87 -> 90 us, finishing pass 27.9 -> 24.9 us, path 36.0 -> 35.9 %: nothing
        # (profiles/r05_const_cols_ab.txt); the pass is at its floor inside the dense Adam's launch
        self._hot_min = int(os.environ.get("MREC_HOT_MIN", "0"))
        self._plan_fork = os.environ.get("MREC_PLAN_FORK", "lookup")       # where the captured step forks its plan branch: "lookup" | "head"
        self.deep_apply_timer = None  # optional ops.KernelTimer armed right before the deep table's sparse apply
        self._dyn = False             # step scalars (Adam powers / step size) in device memory: set per step
        self._front_graph = None      # one-GPU: the whole front of the step (lookups .. MLP backward) as one captured graph
        self._step_graph = None       # ... and, with the wide branch folded, the whole step
        self._sink_graphs = {}        # (sink size, batch shape, id dtype) -> that many whole steps as one graph (train_steps)
        self._step_state = None       # ops.StepState (device-side beta powers / step size), created on first use
        self._state_step = -1         # the step count the device-side state stands at
        self._dropout = bool(cfg.dropout_flag and cfg.dropout_keep_prob < 1.0)
        self._emb_dropped = False     # the lookup has already applied the first layer's Dropout to the rows it handed out
        self._training = False        # inside train_step (`self.training and self.drop_out`, wide_and_deep.py:117)

    # ---- helpers -----------------------------------------------------------------------------
    def _tick(self, name):
        if self.timers is None:
            return None
        if not self._gpu:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        self.timers.setdefault(name, []).append(ev)
        return ev

    @staticmethod
    def _tock(ev):
        if ev is not None:
            ev[1].record()

    def _emb_out(self, n, D, dtype):
        """Static graph input to gather into (None while the step still runs eagerly)."""
        g = self._mlp_graph
        if g is not None and g["emb"].dtype == dtype and g["emb"].numel() == n * D and torch.is_grad_enabled():
            return g["emb"].view(n, D)
        return None

    # ---- forward (eval path: PredictWithSigmoid, wide_and_deep.py:495-518) ---------------------
    def lookup(self, ids, wts, defer_wide=False):
        """One GPU: returns (deep_in [B, F*D] already mask-multiplied, wide_out [B] incl. bias -- or, with the wide branch folded
        into the deep table's kernels, its per-field products for the output head to sum --, None).  defer_wide=True: wide_out is
        returned as a function to call (on whatever stream should do the work) instead of a tensor."""
        cfg = self.cfg
        B, Fd = ids.shape
        ev = self._tick("gather_deep")
        if self._fold_wide and torch.is_grad_enabled():
            # both lookups in one pass over the fused rows; the per-sample sum of the wide products is taken by the head
            d0 = self._drop(0, B)                  # Dropout on the first layer's input rides the lookup (train_step only)
            self._emb_dropped = d0 is not None
            emb, wprod = self.k.gather_rows_wide(self.deep, ids, wts, cfg.emb_dim, out=self._emb_out(B * Fd, cfg.emb_dim, self._amp), drop=d0,
                                                 out_dtype=self._amp, step_state=self._step_state if self._dyn else None)
            self._tock(ev)
            return emb.view(B, Fd * cfg.emb_dim), _WideProd(wprod), None
        if self._mfma and torch.is_grad_enabled():
            emb = self.k.gather_rows(self.deep, ids, wts, out=self._emb_out(B * Fd, cfg.emb_dim, self._amp),
                                     out_dtype=self._amp).view(B, Fd * cfg.emb_dim)
        else:
            emb = self.k.gather_rows(self.deep, ids, wts).view(B, Fd * cfg.emb_dim)
        self._tock(ev)
        if defer_wide:
            return emb, (lambda: self.k.wide_sum(self.wide, ids, wts, self.wide_b)), None
        ev = self._tick("wide_sum")
        wide = self.k.wide_sum(self.wide, ids, wts, self.wide_b)
        self._tock(ev)
        return emb, wide, None

    def predict(self, ids, wts):
        if self._sharded:
            self._guard.poll()
        with torch.no_grad():
            if self._sharded:
                emb, wprod, _ = self._shard_lookup(ids, wts, want_plan=False)       # (a collective: every rank calls predict)
                wide = wprod[..., 0].sum(dim=1) + self.wide_b
            else:
                if self.index is not None or self.hb is not None:
                    ids, _ = self._translate_keys(ids)      # MapTensorGet inserts default rows in eval too (embedding.py:193)
                emb, wide, _ = self.lookup(ids, wts)
            logit = wide.view(-1, 1) + self.mlp(emb)
        return logit, torch.sigmoid(logit)

    def _map_tables(self):
        """(tensor, sigma, fill, seed) of the six tables that share the key index's row numbering, with the values a row gets
        when its key is first seen (the same as a dense table's initial contents, keyed by the key)."""
        cfg = self.cfg
        return [(self.deep, cfg.init_sigma, None, cfg.seed), (self.deep_m, None, 0.0, 0), (self.deep_v, None, 0.0, 0),
                (self.wide, cfg.init_sigma, None, cfg.seed + 1), (self.wide_accum, None, cfg.ftrl_initial_accum, 0),
                (self.wide_linear, None, 0.0, 0)]

    def check_cache(self):
        """host_cache_rows > 0: raises if a batch did not fit the device cache since the tables were created (one host sync;
        the condition is latched on the device -- call once per sink / epoch, as check_shard_overflow)."""
        if self.hb is not None:
            self.hb.check()

    def _translate_keys(self, ids):
        """dynamic_embedding: Unique -> key-index probe / insert -> default rows for new keys (MapTensorGet with
        insert_default_value=True, embedding.py:149,192-195).  Returns (row numbers [B, F] int32, the step's
        SparsePlan with groups mapped to table rows).  The Unique is the one the optimizer side needs anyway."""
        cfg = self.cfg
        if self.hb is not None:
            # host-backed tables: make the batch resident in the device cache (no host sync; check_cache()), rows = cache rows
            plan, rows_pos = self.hb.prepare(ids)
            return rows_pos.view(ids.shape), plan
        d = self.k.unique(ids)                                   # critical path: the gather needs the row numbers
        rows_u = self.index.lookup(d.uniq_buf, insert=True, unique=True, n_dev=d.n_uniq_dev, tables=self._map_tables())
        rows_pos = ops.compose_i32(rows_u, d.inv).view(ids.shape)
        # the inverted index (two radix passes) is needed only by the sparse applies: side stream, under the MLP
        if self._side is not None:
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                plan = ops.group_by_inverse(d)
            for t in (plan.sorted_pos, plan.sorted_seg, plan.seg_offsets):
                self._rs(t, main)
        else:
            plan = ops.group_by_inverse(d)
        plan.uniq_buf = rows_u
        return rows_pos, plan

    @staticmethod
    def _rs(t, stream):
        """record_stream outside graph capture (inside a capture every tensor lives in the graph's own pool)."""
        if not torch.cuda.is_current_stream_capturing():
            t.record_stream(stream)

    def _plan(self, ids):
        """The step's inverted index (Unique + positions per unique id) and, beside it on the same stream, which fields of the batch
        are HOT COLUMNS (mrec_const_cols_detect): one id fills the field -- the reference's Criteo pipeline gives each of the 13 dense
        features one id, process_data.py:138-147 -- or most of it (the id a field's rare categories are folded into).  The folded
        one-GPU apply sums those ids' gradient rows sample by sample instead of through the index.  Whether a batch stream HAS such
        columns is looked at in the engine's first, eager steps (one host read of the mask each); a stream without them is captured
        without the detection launch and with the plain apply kernel."""
        plan = self.k.sparse_plan(ids)
        self._cc = None
        if (self._const_cols and self._fold_wide and self.cfg.sparse and not self._sharded and not self.cfg.dynamic_embedding and self.hb is None
                and ids.dim() == 2 and ids.shape[1] <= 64 and ids.dtype == torch.int32 and ids.is_contiguous() and ids.shape[0] <= 65536
                and hasattr(self.k, "const_cols_detect")):
            probing = self.step_count <= 2 and not (self._gpu and torch.cuda.is_current_stream_capturing())
            if probing or self._hot_seen:
                if self._col_bad is None:
                    self._col_bad = self.k.const_cols_state(self.device)
                B = ids.shape[0]
                st = self.k.const_cols_detect(ids, self.cfg.vocab_size, state=self._col_bad,
                                              min_count=B if self._hot_min <= 0 else max(self._hot_min, -(-B // 8)))
                if st is not None:
                    if probing and not self._hot_seen:
                        self._hot_seen = self.k.const_cols_mask(st) != 0
                    if self._hot_seen:
                        self._cc = (st, ids)
        return plan

    def _front(self, ids, wts, label, capturing=False):
        """Everything of a step in front of the sparse applies: lookups, the step's Unique + inverted index (side
        stream), the MLP forward/backward with the wide branch's work hooked in.  Returns
        (loss, g_emb, g_wide, plan_early, wide_done, route, early_gw, fused); the side stream is joined on return.
        capturing=True: called under HIP-graph capture -- the MLP is issued kernel by kernel."""
        if self._sharded:
            return self._front_sharded(ids, wts, label, capturing)
        cfg = self.cfg
        B, Fd = ids.shape
        inv_sens = 1.0 / cfg.sens
        # the wide branch on the side stream inside the whole-front graph (no graph cut to pay for there); not for the MLP-graph
        # path, where it costs an extra graph boundary
        late = bool(self._side is not None and capturing and self._mfma and not self._fold_wide)
        plan_early, fork_ev = None, None
        if self.index is not None or self.hb is not None:
            ids, plan_early = self._translate_keys(ids)        # from here on `ids` are table row numbers
        elif self._side is not None and not late:
            # the plan needs nothing but the ids -- start it on the side stream BEFORE the gathers are
            # queued, so that it runs beside them (HBM-bound) and eats less into the first GEMM
            main = torch.cuda.current_stream()
            if capturing:
                # Under capture the ORDER of issue decides nothing about concurrency but it decides which branch the graph
                # runtime keeps on the launch queue: the branch whose first node is created first.  That must be the
                # critical chain (gather -> GEMMs -> apply), or every step pays two cross-queue hops of ~10 us on it; so the
                # fork point is only marked here and the plan is issued behind the lookups.
                fork_ev = torch.cuda.Event()
                fork_ev.record(main)
            else:
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    plan_early = self._plan(ids)
                for t in (plan_early.uniq_buf, plan_early.inv, plan_early.n_uniq_dev, plan_early.sorted_pos,
                          plan_early.sorted_seg, plan_early.seg_offsets):
                    self._rs(t, main)
        emb, wide, route = self.lookup(ids, wts, defer_wide=late)
        if fork_ev is not None:
            # (forking it behind the output head instead -- beside the backward GEMMs -- was measured: 0.786 vs 0.764 ms;
            # forking it behind the gather -- which then runs alone: 0.7576 vs 0.7591 ms in round 2; round 4, five A/B pairs on one box: the
            # lookup 47 -> 41 us but the step 0.628-0.629 -> 0.632-0.638 ms -- the plan's kernels and whatever
            # they run beside stretch each other by about the same amount wherever the plan sits (behind the first GEMM: 0.774);
            # the chain cut in two -- the insert kernel here, the rest behind the first GEMM through a second event -- 0.853 ms:
            # another cross-branch dependency, and the graph runtime serialises more than the dependencies ask for)
            if self._plan_fork == "lookup":
                self._side.wait_event(fork_ev)
                with torch.cuda.stream(self._side):
                    plan_early = self._plan(ids)
        if self._side is not None and plan_early is None:
            # Side stream, in this order: (1) the wide branch, which the main stream joins only right before the
            # output head -- it runs while the hidden-layer GEMMs do; (2) the step's Unique + inverted index, which
            # needs only the ids and is joined before the sparse applies: its dozen small latency-bound kernels hide under the MLP.
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)          # the gathers are queued on main: the wide branch starts behind them
            with torch.cuda.stream(self._side):
                if late:
                    wide_t = wide()
                    ev_wide = self._side.record_event()

                    def wide():
                        torch.cuda.current_stream().wait_event(ev_wide)
                        self._rs(wide_t, torch.cuda.current_stream())
                        return wide_t
                plan_early = self._plan(ids)
            for t in (plan_early.uniq_buf, plan_early.inv, plan_early.n_uniq_dev, plan_early.sorted_pos,
                      plan_early.sorted_seg, plan_early.seg_offsets):
                self._rs(t, main)

        ev = self._tick("mlp_fwd_bwd")
        fused = self._mfma
        wide_done = False
        if fused:
            after_head = None
            if self._fold_wide:
                wide_done = True             # the wide table's FTRL rides the deep table's apply (train_step)
                if fork_ev is not None and plan_early is None:
                    # (Round 5, each an A/B pair on one box against forking behind the lookup: behind the forward GEMMs ("fwd") and behind the
                    # output head ("head") the lookup runs alone, 46.3 -> 40.6 us, and the step gets 3-6 us LONGER; the NEXT step's plan
                    # issued behind this step's sparse apply inside a sink's graph -- beside the finishing pass, the dense Adam and the
                    # staging copies -- 0.6263-0.6298 -> 0.6281-0.6313 ms with the lookup at 50.8 us: the plan's kernels cost the chain
                    # ~30 us wherever they run beside it.)
                    # MREC_PLAN_FORK=head: the plan's branch starts behind the output head, beside the BACKWARD launches (multi-round
                    # grids: a CU the plan's kernels slow down simply takes fewer workgroups) instead of beside the lookup and the
                    # one-round layer-0 forward, whose slowest CU sets its time
                    box = {}

                    def after_head(_gw):
                        self._side.wait_event(torch.cuda.current_stream().record_event())
                        with torch.cuda.stream(self._side):
                            box["plan"] = self._plan(ids)
            elif plan_early is not None and self._side is not None and cfg.sparse:
                def after_head(gw_b):
                    # wide FTRL beside the backward GEMMs: needs only the plan (already on the side stream, in order)
                    # and the head's dlogit.  The Mul bprop of wide_mul (:304) is applied as row_scale.
                    main = torch.cuda.current_stream()
                    self._side.wait_event(main.record_event())
                    self._rs(gw_b, self._side)
                    with torch.cuda.stream(self._side):
                        gw = gw_b.view(B, 1).expand(B, Fd).reshape(B * Fd, 1)
                        self.k.sparse_ftrl_(self.wide, self.wide_accum, self.wide_linear, plan_early, gw, wts, lr=cfg.ftrl_lr,
                                            l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, grad_scale=inv_sens)
                wide_done = True
            if capturing:
                bh = None
                if after_head is not None and self._plan_fork == "fwd" and self._fold_wide:
                    bh, after_head = (lambda: after_head(None)), None      # MREC_PLAN_FORK=fwd: behind the forward GEMMs, beside the tail launch
                loss, g_emb, g_wide = self._mlp_step_eager(emb, wide, label, after_head=after_head, before_head=bh)
            else:
                loss, g_emb, g_wide = self._mlp_step(emb, wide, label, after_head=after_head)
            if plan_early is None and fork_ev is not None:
                plan_early = box["plan"]
        elif self._f32net:
            loss, g_emb, g_wide = self._mlp_step_f32(emb, wide, label)   # the fp32 net by hand (ops.dense32_*)
        else:
            loss, g_emb, g_wide = self._mlp_step_generic(emb, wide, label)      # (no HIP path: the product raises UnsupportedNet)
        self._tock(ev)

        if plan_early is not None and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)          # the plan (queued long ago) is done
        return loss, g_emb, g_wide, plan_early, wide_done, route, None, fused

    # ---- the whole front of a one-GPU step as ONE HIP graph ------------------------------------------
    def _front_graph_ok(self):
        """The front of the step (and, with device-side step scalars, the whole step) can replay as one graph: one GPU, or a
        shard whose collectives can be captured (RCCL) and whose whole step has constant arguments."""
        return bool(self._graph_level >= 2 and self._gpu and self._side is not None
                    and self._mfma and self.step_count > 2 and torch.is_grad_enabled() and self.timers is None
                    and self.hb is None
                    and (not self._sharded or (self._graph_level >= 3 and self._shard_fold and getattr(self.comm, "capturable", False))))

    def _front_replay(self, ids, wts, label):
        """ids / wts / label are copied into static buffers (3.5 MB) and the captured front is replayed: the deep
        gather, wide_sum, the plan on its side branch, the MLP and the wide table's FTRL apply (constant hyper-
        parameters) on another side branch -- one launch, no graph boundaries inside (each costs ~25 us of device
        time on this stack).  The deep LazyAdam apply and the dense Adam stay outside: their step size changes
        every step (bias correction) and is a kernel argument."""
        g = self._front_graph
        if g is None or g["ids"].shape != ids.shape or g["ids"].dtype != ids.dtype:
            g = self._capture_front(ids.clone(), wts.clone(), label.clone())
            if g is None:
                return None
            self._front_graph = g
        self.k.copy3_((g["ids"], g["wts"], g["label"]), (ids, wts, label))        # one launch, not three
        g["graph"].replay()
        return g["out"]

    def _capture_front(self, ids, wts, label):
        """Captures the front on exactly these input tensors (kept alive by the returned dict)."""
        try:
            g = {"ids": ids, "wts": wts, "label": label}
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                g["out"] = self._front(ids, wts, label, capturing=True)
            g["graph"] = graph
            return g
        except RuntimeError as e:
            import warnings
            warnings.warn(f"HIP-graph capture of the step front failed, falling back to the MLP graphs: {e}")
            self._graph_level = min(self._graph_level, 1)
            self._front_graph = None
            return None

    # ---- ... and the whole step, optimizers included (wide branch folded: every kernel argument is constant) --------
    def _step_replay(self, ids, wts, label):
        g = self._step_graph
        if g is None or g["ids"].shape != ids.shape or g["ids"].dtype != ids.dtype:
            g = self._capture_step(ids.clone(), wts.clone(), label.clone())
            if g is None:
                return None
            self._step_graph = g
        self.k.copy3_((g["ids"], g["wts"], g["label"]), (ids, wts, label))
        g["graph"].replay()
        self.last_plan = g["plan"]
        return g["loss"]

    def _choose_finish(self):
        """Where the sparse apply's finishing pass runs: inside the dense net's Adam launch (default), or -- MREC_FUSE_FINISH=0 -- as a
        launch of its own in front of it.  Measured on one box (round 5): uniform ids have nothing to finish (an empty pass is free
        inside the Adam launch, ~4 us as a launch); Zipf ids x 39 fields: the pass takes 43 us inside the Adam launch (its dependent
        round trips queue behind 4.7 TB/s of streaming) and 31 us alone, but the step 0.715 ms fused against 0.732-0.737 separate --
        fused, the pass hides under the Adam pass it runs beside."""
        self._fuse_finish = os.environ.get("MREC_FUSE_FINISH", "1") != "0"

    def _capture_step(self, ids, wts, label):
        try:
            self._choose_finish()
            g = {"ids": ids, "wts": wts, "label": label}
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                front = self._front(ids, wts, label, capturing=True)
                g["loss"] = self._tail(front, ids, wts)
            g["graph"], g["plan"] = graph, self.last_plan
            return g
        except RuntimeError as e:
            import warnings
            warnings.warn(f"HIP-graph capture of the whole step failed, falling back to the front graph: {e}")
            self._graph_level = min(self._graph_level, 2)
            self._step_graph = None
            return None

    def _tail_dense(self, front, ids, wts):
        """sparse=False (the reference's default): the gradient of an embedding table is a DENSE [V, D] tensor -- the
        scatter-add of the row gradients (Gather bprop) plus, for the deep table, l2_coef * E from the L2 term of the deep
        loss (NetWithLossClass.construct, wide_and_deep.py:356-360) -- and nn.Adam / nn.FTRL (:434-445) update EVERY row
        every step: moments decay, the L2 pull acts on rows the batch never touched, and FTRL re-derives every wide weight
        from its accumulators.  O(V * D) per step: the small-vocabulary configuration."""
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.emb_dim
        inv_sens = 1.0 / cfg.sens
        loss, g_emb, g_wide, plan_early, wide_done, route, early_gw, fused = front
        ev = self._tick("plan")
        plan = plan_early if plan_early is not None else self.k.sparse_plan(ids)
        self._tock(ev)
        if getattr(self, "_gwide", None) is None:
            self._gwide = torch.empty_like(self.wide)
        akw = dict(lr=cfg.adam_lr, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.adam_eps,
                   beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=inv_sens)
        ev = self._tick("apply_deep")
        sums = self.k.segment_sum(plan, g_emb.view(B * Fd, D).float(), wts)           # UnsortedSegmentSum of the row gradients
        # nn.Adam over the WHOLE table (:434-437).  Its gradient -- the bprop of Gather -- is nonzero on the touched rows only: the
        # kernel looks every row's group sum up (no [V, D] gradient is zeroed, scattered into and read back).
        # d/dE [l2_coef * sum(E^2) / 2] = l2_coef * E, carried at the loss scale like every other gradient: added inside the Adam
        # kernel (product rounded, then added: no fused multiply-add), which also leaves sum(E^2) of the step's starting values behind
        if getattr(self, "_l2_sumsq", None) is None:
            self._l2_sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.k.dense_adam_rows_l2_(self.deep, self.deep_m, self.deep_v, plan, sums, cfg.l2_coef * cfg.sens, sumsq=self._l2_sumsq, **akw)
        self._tock(ev)
        ev = self._tick("apply_wide")
        gw = (g_wide.view(B, 1) * wts).view(B * Fd, 1)                                # Mul bprop of wide_mul (:304)
        self._gwide.zero_()
        self.k.scatter_unique_rows_(self._gwide, plan, self.k.segment_sum(plan, gw, None))
        self.k.dense_ftrl_(self.wide.view(-1), self.wide_accum.view(-1), self.wide_linear.view(-1), self._gwide.view(-1),
                           lr=cfg.ftrl_lr, l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, grad_scale=inv_sens)
        self._tock(ev)
        ev = self._tick("apply_dense")
        # d loss / d Wide_b = sum of dlogit (= the output layer's bias gradient), into Wide_b's slot of the dense gradient
        if fused:
            self._sum_dw_slabs()
            self.wide_b_grad.copy_(self.dense_grad[2 * (len(self.dims) - 2) + 1].view(1))
        else:
            self.wide_b_grad.copy_(g_wide.sum().view(1))
        self.k.dense_adam_(self.dense_flat.detach(), self.dense_m, self.dense_v, self.dense_grad_flat, ftrl1=self._wb_ftrl, **akw)
        if fused:
            self.dense16_flat.copy_(self.dense_flat.detach())
            self._refresh_tail()
        self._tock(ev)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self.last_plan = plan
        return loss.detach()

    def deep_loss(self, loss):
        """The deep optimizer's loss of NetWithLossClass.construct (:356-360) of the LAST step: log loss + l2_coef * sum(E^2) / 2
        in dense mode -- sum(E^2) at the step's starting values, a by-product of that step's Adam pass over the table --, the
        log loss itself in sparse mode."""
        if self.cfg.sparse:
            return loss
        if getattr(self, "_l2_sumsq", None) is None:
            raise RuntimeError("deep_loss: no training step has run yet")
        return loss + self.cfg.l2_coef * 0.5 * float(self._l2_sumsq[0])

    def release_graphs(self):
        """Drops every captured HIP graph (they are re-captured on demand).  A shard's graphs hold RCCL kernels: they must be gone
        before the process group is destroyed -- destroy_process_group() waits for them forever otherwise."""
        if self._gpu:
            torch.cuda.synchronize(self.device)
        self._step_graph = self._front_graph = self._mlp_graph = None
        self._sink_graphs = {}
        if self._gpu:
            torch.cuda.synchronize(self.device)

    def close(self):
        """End of training: the last steps' dropped-position check (row shards; raises ShardCapacityError on every rank alike),
        then the captured graphs are released -- call it (or leave a `with engine:` block) before destroy_process_group()."""
        try:
            if self._sharded:
                self.check_shard_overflow()
            self.check_cache()
        finally:
            self.release_graphs()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        else:
            self.release_graphs()
        return False

    def __del__(self):
        try:
            if getattr(self, "_sink_graphs", None) or getattr(self, "_step_graph", None) is not None:
                self.release_graphs()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- one training step -------------------------------------------------------------------
    def train_steps(self, batches):
        """`len(batches)` training steps per host call -- the reference's dataset_sink_mode / sink_size (Model.train(...,
        dataset_sink_mode=True), models/wide_deep/train_and_eval.py:98-101; sink_size, train_and_eval_distribute.py:115-116: the steps
        of a sink run on the device without returning to the host; RecModel.online_train restricts it to 1, rec_model.py:268-271).  Where the whole
        step replays as one HIP graph, a sink of S steps replays as ONE graph of S steps: between two graph launches the GPU idles
        for ~14 us (launch latency, with or without the staging copy in between), which a sink pays once per S steps.  Same
        kernels on the same data in the same order as S train_step calls: identical results.  Returns the S losses (copies the
        caller may keep)."""
        S = len(batches)
        g = self._step_graph
        if self._sharded:
            self._guard.poll()                        # drops of the previous call: raised here, on every rank (OverflowGuard)
        if (S > 1 and g is not None and self._graph_level >= 3 and self._front_graph_ok() and self._dyn and self._state_step == self.step_count
                and all(b[0].shape == g["ids"].shape and b[0].dtype == g["ids"].dtype for b in batches)):
            key = (S, tuple(g["ids"].shape), g["ids"].dtype)
            sg = self._sink_graphs.get(key)
            if sg is None and key not in self._sink_graphs:
                sg = self._capture_sink(key, [tuple(t.clone() for t in b) for b in batches])
            if sg is not None:
                self.k.copy_many_([t for b in sg["inputs"] for t in b], [t for b in batches for t in b])       # one launch
                for _ in range(S):
                    self.step_count += 1
                    self.beta1_power = np.float32(self.beta1_power * self.beta1)
                    self.beta2_power = np.float32(self.beta2_power * self.beta2)
                self._state_step = self.step_count
                self.deep_apply_timer = None
                sg["graph"].replay()
                self.last_plan = sg["plan"]
                if self._sharded:
                    self._guard.probe()
                return sg["out"]       # the sink's losses [S], gathered inside the graph: valid until this sink shape runs again
        return [self.train_step(*b).clone() for b in batches]       # (train_step hands out a static buffer once graphs replay)

    def _capture_sink(self, key, inputs):
        try:
            self._choose_finish()
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            losses = []
            self._training = True
            # the sink's losses: ONE [S] tensor whose elements the steps' tail launches write directly (round 5; a torch.stack of S
            # scalars inside the graph was a 4-us copy kernel per sink) -- where the fused tail launch owns the loss buffer
            ring = None
            B0 = inputs[0][0].shape[0]
            if self._mfma and self._tail_now(B0):
                ring = torch.empty(len(inputs), dtype=torch.float32, device=self.device)
                for slot in range(len(inputs)):
                    self._tail_out.setdefault((B0, self._amp, slot), {})["loss"] = ring[slot:slot + 1]
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for slot, (ids, wts, label) in enumerate(inputs):
                    self._slot = slot                 # per-step output buffers of the tail launch (the losses must not alias)
                    front = self._front(ids, wts, label, capturing=True)
                    losses.append(self._tail(front, ids, wts))
                if ring is not None and all(l.data_ptr() == ring[i:i + 1].data_ptr() for i, l in enumerate(losses)):
                    out = ring
                else:
                    out = torch.stack([l.reshape(()) for l in losses])       # inside the graph: no eager kernel between two sinks
            sg = {"graph": graph, "inputs": inputs, "losses": losses, "out": out, "plan": self.last_plan}
            self._sink_graphs[key] = sg
            return sg
        except RuntimeError as e:
            import warnings
            warnings.warn(f"HIP-graph capture of a {len(inputs)}-step sink failed, running it step by step: {e}")
            self._sink_graphs[key] = None
            return None
        finally:
            self._training = False
    
    def train_step(self, ids, wts, label):
        self._training = True
        if self._sharded:
            self._guard.poll()
        try:
            loss = self._train_step(ids, wts, label)
            if self._sharded and not (self._gpu and torch.cuda.is_current_stream_capturing()):
                self._guard.probe()
            return loss
        finally:
            self._training = False

    def _train_step(self, ids, wts, label):
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.emb_dim
        inv_sens = 1.0 / cfg.sens
        self._dyn = bool((self._fold_wide or (self._sharded and self._shard_fold)) and self._mfma and self._gpu)
        if self._dyn:
            # device-side step scalars (ops.StepState): brought in line with the host mirrors whenever somebody else moved
            # those (first step, load_checkpoint)
            if self._step_state is None:
                self._step_state = self.k.StepState(self.device)
            if self._state_step != self.step_count:
                self._step_state.reset(self.beta1_power, self.beta2_power, self.step_count)
        elif self._dropout and self._gpu and self.k is ops:
            # the Dropout kernels read the step from device memory (so that captured MLP graphs replay with a moving step):
            # where nothing else keeps a device-side step state, keep one for them
            if self._step_state is None:
                self._step_state = self.k.StepState(self.device)
            self._step_state.reset(self.beta1_power, self.beta2_power, self.step_count)
        self.step_count += 1
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._state_step = self.step_count

        if self._dyn and self._graph_level >= 3 and self._front_graph_ok():
            self.deep_apply_timer = None              # (HIP events cannot be timed from inside a graph; the kernel stamps
            loss = self._step_replay(ids, wts, label)  # its own begin / end in the step state instead)
            if loss is not None:
                return loss
        front = None
        if self._front_graph_ok() and not self._sharded:
            front = self._front_replay(ids, wts, label)
        if front is None:
            front = self._front(ids, wts, label)
        return self._tail(front, ids, wts)

    def _tail(self, front, ids, wts):
        """The optimizer half of a step: sparse applies, dense Adam / FTRL (shards: mindrec_amd/wide_deep_shard.py)."""
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.emb_dim
        inv_sens = 1.0 / cfg.sens
        loss, g_emb, g_wide, plan_early, wide_done, route, early_gw, fused = front
        if route is not None:
            return self._tail_sharded(front, ids, wts)
        if not cfg.sparse:
            return self._tail_dense(front, ids, wts)
        state = None
        if self._dyn:
            state = self._step_state
            # (one thread: powers *= betas, lr_t.  Folded into the fused tail's finishing launch -- the last launch in front of it that
            # reads nothing of the state -- it saved 1.4 us of a 0.626 ms step, three A/B pairs: the gap in front of this launch is the
            # join's with the plan's branch and stays; not kept -- DESIGN.md section 9)
            state.advance(cfg.adam_lr, float(self.beta1), float(self.beta2))

        if fused:
            # d loss / d Wide_b = sum of dlogit = the output layer's bias gradient, which the head kernel has
            # already reduced into the flat gradient buffer
            gb = self.dense_grad[2 * (len(self.dims) - 2) + 1].view(1)
        else:
            gb = g_wide.sum().view(1)
        ev = self._tick("plan")
        plan = plan_early if plan_early is not None else self._plan(ids)
        self._tock(ev)
        # (The dense Adam on the side branch beside the sparse applies was measured: the step gains ~1 % on one box and nothing
        # on the next, and the apply kernel -- two HBM-bound kernels sharing the memory system -- stretches from 0.185 to 0.20 ms.)
        # (Running the wide FTRL apply on the side stream beside the deep apply was tried and rejected:
        # sharing CUs drops the deep kernel from 5.0 to 4.0 TB/s and the step gets 0.11 ms longer.)
        ev = self._tick("apply_deep")
        if self.deep_apply_timer is not None:
            self.deep_apply_timer.arm()
            self.deep_apply_timer = None
        finish = None
        if self._fold_wide and fused:
            # LazyAdam on the deep columns + FTRL on the wide record of the same rows: one visit per touched row.  The pass that
            # finishes the runs crossing windows of the sorted index is handed back (`finish`) and rides the dense Adam launch
            # below: a latency-bound launch less on the critical path
            finish = self.k.sparse_lazy_adam_wide_(self.deep, self.deep_m, self.deep_v, plan, g_emb.view(B * Fd, D), wts, g_wide, Fd, D,
                                                   lr=cfg.adam_lr, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.adam_eps,
                                                   beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                                                   grad_scale=inv_sens, ftrl_lr=cfg.ftrl_lr, l1=cfg.ftrl_l1, l2=cfg.ftrl_l2,
                                                   step_state=state, defer=self._fuse_finish, const_cols=getattr(self, "_cc", None))
        else:
            self.k.sparse_lazy_adam_(self.deep, self.deep_m, self.deep_v, plan, g_emb.view(B * Fd, D), wts, lr=cfg.adam_lr,
                                     beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.adam_eps,
                                     beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                                     grad_scale=inv_sens)
        self._tock(ev)
        if not wide_done:
            ev = self._tick("apply_wide")
            gw = g_wide.view(B, 1).expand(B, Fd).reshape(B * Fd, 1)      # Mul bprop of wide_mul (:304): the mask is
            self.k.sparse_ftrl_(self.wide, self.wide_accum, self.wide_linear, plan, gw, wts, lr=cfg.ftrl_lr,   # applied as row_scale
                                l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, grad_scale=inv_sens)
            self._tock(ev)
        if self._side is not None and not self._dyn:          # (folded one-GPU step: nothing was queued on the side stream since
            torch.cuda.current_stream().wait_stream(self._side)   # the front joined it, and a join costs ~6 us inside a graph)
        ev = self._tick("apply_dense")
        akw = dict(lr=cfg.adam_lr, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.adam_eps,
                   beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=inv_sens)
        flat = self.dense_flat.detach()
        if not (self._fold_wide and fused):
            self.wide_b_grad.copy_(gb)                     # Wide_b's slot of the dense gradient (updated by the Adam below;
                                                           # the folded path's head kernel has already written it)
        if fused:
            # the weight gradients stay fp32 batch slabs and are added up inside the Adam kernel (nobody else needs the sums);
            # the kernel also refreshes the 16-bit operand shadow
            self.k.dense_adam_slabs_(flat, self.dense_m, self.dense_v, self.dense_grad_flat, self._slab_segments(),
                                     shadow16=self.dense16_flat, step_state=state, ftrl1=self._wb_ftrl, finish=finish, **akw)
            self._refresh_tail()
        else:
            self.k.dense_adam_(flat, self.dense_m, self.dense_v, self.dense_grad_flat, ftrl1=self._wb_ftrl, **akw)
        self._tock(ev)
        if self._side is not None and not self._dyn:
            torch.cuda.current_stream().wait_stream(self._side)
        self.last_plan = plan
        return loss.detach()


def synthetic_batch(cfg: WideDeepConfig, device, dist_kind="uniform", seed=1000, rank=0, signal=False):
    """ids int32 [B,F], wts f32 [B,F], label f32 [B,1] -- the dataset contract of
    models/wide_deep/src/datasets.py:212-216.  F = 39 puts the 13 dense fields on the constant ids
    0..12 with weights in [0,1) (process_data.py:138-147); categorical weights are 1.0 (:149-162).
      uniform : ids uniform over [0, V)            (worst case for HBM: ~no duplicates)
      zipf    : Zipf(1.05) per slot over V/n_cat-sized sub-ranges  (realistic duplicate rate)
    signal=True plants a hidden linear model over the ids (label ~ Bernoulli(sigmoid(sum_f a[id]))) so
    that AUC is meaningful; otherwise labels are Bernoulli(0.25) noise.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed * 1000003 + rank)
    B, Fd, V = cfg.batch_size, cfg.field_size, cfg.vocab_size
    n_dense = 13 if Fd == 39 else 0
    n_cat = Fd - n_dense
    idt = torch.int32 if cfg.id_dtype == "int32" else torch.int64
    if dist_kind == "uniform":
        cat = torch.randint(n_dense, V, (B, n_cat), generator=g, dtype=torch.int64)
    elif dist_kind == "zipf":
        rng = np.random.default_rng(seed * 1000003 + rank)
        slot = max((V - n_dense) // n_cat, 1)
        z = np.minimum(rng.zipf(1.05, size=(B, n_cat)) - 1, slot - 1)
        cat = torch.from_numpy(z + n_dense + slot * np.arange(n_cat)[None, :])
    else:
        raise ValueError(dist_kind)
    wts = torch.ones((B, Fd), dtype=torch.float32)
    if n_dense:
        ids = torch.cat([torch.arange(n_dense, dtype=torch.int64).expand(B, n_dense), cat], dim=1)
        wts[:, :n_dense] = torch.rand((B, n_dense), generator=g)
    else:
        ids = cat
    if signal:
        h = (ids * 2654435761 + 12345) % 1000003                       # fixed pseudo-random weight per id
        a = (h.double() / 1000003.0 - 0.5) * 1.6
        logit = (a * wts.double()).sum(dim=1, keepdim=True) - 0.8
        label = (torch.rand((B, 1), generator=g).double() < torch.sigmoid(logit)).float()
    else:
        label = (torch.rand((B, 1), generator=g) < 0.25).float()
    return ids.to(idt).to(device), wts.to(device), label.to(device)


def embedding_bytes(N, U, D, s=4, act_bytes=4):
    """Algorithmic bytes of the embedding path per step.
    act_bytes: bytes per element of the looked-up rows written by the gather and of the row gradients
    read by the apply -- 4 for fp32 (the formulas of SURVEY 8(d)), 2 when the bf16 MLP path has the
    gather emit bf16 and the apply read bf16 gradients."""
    return {
        "lookup": N * s + U * D * 4 + N * D * act_bytes,
        "apply_deep": N * s + N * D * act_bytes + U * 6 * D * 4,
        "wide_lookup": N * (s + 8),
        "apply_wide": N * s + N * 4 + U * 24,
    }
