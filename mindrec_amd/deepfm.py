"""DeepFM training step on the MI355X kernels.

Mirrors models/deepfm/src/deepfm.py of the reference:
  DeepFMModel.construct      :206-237  linear term  sum_f W_l2[id] * wt,  FM term over vx = V_l2[id] * wt,
                                       5-layer MLP on vx,  out = linear + fm + deep
  NetWithLossClass.construct :252-259  mean sigmoid-CE + l2_coef * 0.5 * (sum V_l2^2 + sum W_l2^2)  (whole tables)
  TrainStepWrap              :263-295  dense Adam(lr 5e-4, eps 5e-8, loss_scale 1024) over ALL parameters

Both tables carry an L2 term over every row, so their gradients are dense: sens * l2_coef * table, plus
the segment-sum of the row gradients scattered onto the touched rows; Adam then sweeps the whole tables
(vocab 184 965: 59 MB, trivial next to the batch).  Lookups, the FM term (one pass, mrec_fm.hip), the
segment-sums and the Adam sweeps are libmrec_hip.so kernels, and so is the dense net in the reference's
precision (convert_dtype: float16): the hand-written MFMA DenseLayer kernels of the Wide&Deep step
(mindrec_amd/wide_deep_mlp.py: forward with bias + ReLU epilogues, fused backward, the tail of the net as one
launch, dense Adam over fp32 batch slabs), the output head taking linear + fm as its per-sample addend.
mlp_dtype="fp32" keeps a torch restatement (library GEMMs) -- the form the oracle-side engine runs.
"""
import contextlib
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import ops
from .wide_deep_mlp import DenseNetMixin


@dataclass
class DeepFMConfig:
    """models/deepfm/default_config.yaml:14-33."""
    data_vocab_size: int = 184965
    data_emb_dim: int = 80
    data_field_size: int = 39
    batch_size: int = 16000
    deep_layer_dims: List[int] = field(default_factory=lambda: [1024, 512, 256, 128])
    l2_coef: float = 8e-5
    learning_rate: float = 5e-4
    epsilon: float = 5e-8
    loss_scale: float = 1024.0
    seed: int = 1000
    init_sigma: float = 0.01
    mlp_dtype: str = "fp16"       # DenseLayer casts input, weight and bias to float16 (convert_dtype: True, default_config.yaml:27;
                                  # deepfm.py:135-145); "bf16" runs the same kernels, "fp32" the fp32 kernels (ops.x3_* / ops.dense32_*)
    graphs: str = "mlp"           # "mlp": the dense net's step replays as HIP graphs (16-bit net), "none": kernel by kernel
    fp32_matmul: str = "x3"       # fp32 net: "x3" three-part bf16 operands on the 16-bit matrix instruction, "exact" the fp32-input one


class _DeepFMNet(DenseNetMixin):
    """The dense side both DeepFM engines share: DenseNetMixin's state and 16-bit MFMA step when the shapes allow it
    (self._mfma), the torch fp32 restatement otherwise."""

    def _init_net(self, cfg, dims):
        self._amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[cfg.mlp_dtype]
        self._mfma = bool(self._gpu and self.k is ops and self._amp is not None and self.mfma_net_ok(dims) and cfg.data_emb_dim % 4 == 0)
        self._graph_level = 1 if (cfg.graphs == "mlp" and self._mfma) else 0
        self.rank, self.step_count, self._step_state = getattr(self, "rank", 0), 0, None
        self._dropout = self._training = self._emb_dropped = False
        self._init_dense_net(dims, cfg.seed + 2, cfg.init_sigma, cfg.loss_scale)
        self.beta1, self.beta2 = np.float32(0.9), np.float32(0.999)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)

    def _adam_kw(self, grad_scale):
        cfg = self.cfg
        return dict(lr=cfg.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.epsilon,
                    beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=grad_scale)

    def _mlp(self, x):
        """DenseLayer x5 (deepfm.py:98-150): the 16-bit kernels' inference path, or the torch restatement."""
        return self.mlp(x)

    def _net_step(self, vx, vx16, linear, label):
        """Forward + backward of everything behind the lookups: FM term, the dense net, the loss.  Returns (loss, g_vx fp32
        [B, F, D], g_linear [B]); dense gradients are left in the slabs / the flat gradient buffer for _dense_adam."""
        cfg = self.cfg
        B, Fd, D = vx.shape
        if self._mfma:
            if vx16 is None:
                # the net's 16-bit input = the very values the FM term reads, rounded once (deepfm.py:135-137 casts that tensor): written
                # by the FM kernel's pass over them -- straight into the MLP graph's static input where there is one -- not looked up
                # a second time
                g = self._mlp_graph
                if g is not None and g["emb"].dtype == self._amp and g["emb"].numel() == vx.numel():
                    vx16 = g["emb"].view(B, Fd, D)
                else:
                    bufs = self.__dict__.setdefault("_vx16", {})
                    vx16 = bufs.get((B, Fd, D))
                    if vx16 is None:
                        vx16 = bufs[(B, Fd, D)] = torch.empty((B, Fd, D), dtype=self._amp, device=self.device)
                lin_fm, cs = self.k.fm_forward(vx, add=linear, out16=vx16)     # linear + fm: the output head's per-sample addend
            else:
                lin_fm, cs = self.k.fm_forward(vx, add=linear)
            loss, g16, dlogit = self._mlp_step(vx16.view(B, Fd * D), lin_fm, label)
            return loss, self.k.fm_backward_mix(g16.view(B, Fd, D), vx, cs, dlogit), dlogit
        if self._f32net:
            # the fp32 net by hand (ops.dense32_*, the fp32 output head); the FM term's bprop is added to the MLP's input gradient
            lin_fm, cs = self.k.fm_forward(vx, add=linear)
            loss, g, dlogit = self._mlp_step_f32(vx.view(B, Fd * D), lin_fm, label)
            g = g.view(B, Fd, D)
            self.k.fm_backward_(g, vx, cs, dlogit)
            return loss, g, dlogit
        return self._net_step_generic(vx, linear, label)

    def _net_step_generic(self, vx, linear, label):
        """No HIP path for this net: the product refuses (tests/_torch_net.py implements it for the oracle side)."""
        raise self._unsupported("training step")

    def _dense_adam(self, grad_scale, summed=False):
        """nn.Adam over the dense net.  summed: the weight-gradient slabs have been added into the flat gradient already (a shard
        all-reduces the sum)."""
        kw = self._adam_kw(grad_scale)
        if self._mfma:
            self.k.dense_adam_slabs_(self.dense_flat.detach(), self.dense_m, self.dense_v, self.dense_grad_flat,
                                     [] if summed else self._slab_segments(), shadow16=self.dense16_flat, **kw)
            self._refresh_tail()
        else:
            self.k.dense_adam_(self.dense_flat.detach(), self.dense_m, self.dense_v, self.dense_grad_flat, **kw)


class DeepFMEngine(_DeepFMNet):
    _kernels = ops               # the op set (tests/ subclass the engine with the oracle's restatements to check it step for step)
    _allow_cpu = False           # the product has no CPU path

    def __init__(self, cfg: DeepFMConfig, device):
        self.cfg, self.device = cfg, torch.device(device)
        self.k = self._kernels
        self._gpu = self.device.type == "cuda"
        if not self._gpu and not self._allow_cpu:
            raise RuntimeError("DeepFMEngine runs on an MI355X (no CPU fallback)")
        V, D, dev = cfg.data_vocab_size, cfg.data_emb_dim, self.device
        with (torch.cuda.device(dev) if self._gpu else contextlib.nullcontext()):
            self.V_l2 = torch.empty((V, D), dtype=torch.float32, device=dev)
            self.k.fill_normal_(self.V_l2, cfg.seed, cfg.init_sigma)
            self.W_l2 = torch.empty((V, 1), dtype=torch.float32, device=dev)
            self.k.fill_normal_(self.W_l2, cfg.seed + 1, cfg.init_sigma)
            self.state = {n: (torch.zeros_like(t), torch.zeros_like(t)) for n, t in (("V", self.V_l2), ("W", self.W_l2))}
            self._init_net(cfg, [cfg.data_field_size * D] + list(cfg.deep_layer_dims) + [1])

    def _forward(self, ids, wts):
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.data_emb_dim
        vx = self.k.gather_rows(self.V_l2, ids, wts)                     # [B, F, D], mask fused
        linear = self.k.wide_sum(self.W_l2, ids, wts)                    # [B]
        return vx, linear

    def predict(self, ids, wts):
        with torch.no_grad():
            vx, linear = self._forward(ids, wts)
            fm, _ = self.k.fm_forward(vx)
            logit = (linear + fm).view(-1, 1) + self._mlp(vx.view(vx.shape[0], -1))
        return logit, torch.sigmoid(logit)

    def train_step(self, ids, wts, label):
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.data_emb_dim
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self.step_count += 1
        # the step's Unique + inverted index (the bprop of Gather needs them, they need only the ids): on a side stream under the net
        plan, main = None, None
        if self._gpu:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(self.device)
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                plan = self.k.sparse_plan(ids)
            for t in (plan.uniq_buf, plan.inv, plan.n_uniq_dev, plan.sorted_pos, plan.sorted_seg, plan.seg_offsets):
                t.record_stream(main)
        vx, linear = self._forward(ids, wts)
        log_loss, g_vx, g_lin = self._net_step(vx, None, linear, label)
        # Dense table gradients = the segment sums scattered to the touched rows + sens * l2_coef * table everywhere (the L2 term of
        # the loss, deepfm.py:252-259).  The second half -- and the term's own value, l2_coef / 2 * (sum V^2 + sum W^2) at the
        # step's starting values -- come out of the Adam kernel's one pass over each table (ops.dense_adam_l2_).
        if plan is None:
            plan = self.k.sparse_plan(ids)
        else:
            main.wait_stream(self._side)
        sens = cfg.loss_scale
        kw = dict(lr=cfg.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.epsilon,
                  beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=1.0 / sens)
        if getattr(self, "_sumsq", None) is None:
            self._sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        for i, (name, table, g, scale) in enumerate((("V", self.V_l2, g_vx.view(B * Fd, D), wts),
                                                      ("W", self.W_l2, (g_lin.view(B, 1) * wts).view(B * Fd, 1), None))):
            # nn.Adam over the WHOLE table, the gradient nonzero on the touched rows only: the group sums are looked up per row inside
            # the Adam kernel's one pass (no [V, D] gradient is zeroed, scattered into and read back)
            m, v = self.state[name]
            self.k.dense_adam_rows_l2_(table, m, v, plan, self.k.segment_sum(plan, g, scale), cfg.l2_coef * sens, sumsq=self._sumsq,
                                       accumulate=i > 0, **kw)
        loss = log_loss.detach() + (self._sumsq * (cfg.l2_coef * 0.5)).to(torch.float32).view(())
        self._dense_adam(1.0 / sens)
        return loss


class DeepFMHashEngine(_DeepFMNet):
    """BASELINE configs[4]: the DeepFM model over MapParameter hash embeddings (int64 keys, dim 128,
    admission / eviction filters on).  The reference contains the two halves but not this composition
    (only models/wide_deep builds a HashEmbeddingLookup, wide_and_deep.py:271-274), so it is assembled
    the way the reference assembles Wide&Deep over hash tables:

      * V (dim D) and W (dim 1) are MapParameters; a step's keys are looked up with
        MapTensorGet(insert_default_value=True) semantics: Unique -> probe/insert -> default rows
        (embedding.py:184-206);
      * model math is DeepFMModel.construct (deepfm.py:206-237): linear + FM + MLP;
      * dynamic tables have no whole-table L2 term and no dense sweep: both tables take the sparse
        LazyAdam the reference pairs with hash tables (wide_and_deep.py:415-422), applied to admitted
        rows only (permit_filter_value); evict() drops keys not seen for evict_filter_value steps.
    One Unique serves both tables (same keys, as wide_and_deep.py:300-302)."""

    def __init__(self, cfg: DeepFMConfig, device, key_dtype=torch.int64, capacity=1 << 22, permit_filter_value=1,
                 evict_filter_value=None, rank=0, world=1, comm=None, shard_capacity_factor=1.25):
        """rank / world > 1: both hash tables are sharded by key -- owner = hash(key) mod world, the raw keys travel, every owner
        keeps its own key index, admission counters and optimizer state -- over the fixed-capacity exchange of
        mindrec_amd/wide_deep_shard.py (static message shapes, no host round trip); the dense net is data parallel with one
        all-reduce(mean).  comm: collectives provider (default: torch.distributed, i.e. RCCL on device tensors)."""
        from .experimental import MAX_SIZE, MapParameter
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeepFMHashEngine runs on an MI355X (no CPU fallback)")
        self.rank, self.world, self.cap_factor = int(rank), int(world), float(shard_capacity_factor)
        if self.world > 1:
            from .wide_deep import _DirectComm
            self.comm = comm if comm is not None else _DirectComm()
            self._overflow = torch.zeros(1, dtype=torch.int64, device=self.device)
        D, dev = cfg.data_emb_dim, self.device
        ev = MAX_SIZE if evict_filter_value is None else evict_filter_value
        mk = dict(key_dtype=key_dtype, default_value="normal", permit_filter_value=permit_filter_value,
                  evict_filter_value=ev, capacity=capacity, device=dev)
        self.V = MapParameter(value_shape=(D,), name="V_l2", seed=cfg.seed, **mk)
        self.W = MapParameter(value_shape=(1,), name="W_l2", seed=cfg.seed + 1, **mk)
        for t in (self.V, self.W):
            t.add_slot("moment1", 0.0)
            t.add_slot("moment2", 0.0)
        self.k, self._gpu = ops, True
        with torch.cuda.device(dev):
            self._init_net(cfg, [cfg.data_field_size * D] + list(cfg.deep_layer_dims) + [1])
        if self.world > 1:
            from .wide_deep_shard import OverflowGuard
            n = self.dense_flat.numel()
            self._guard = OverflowGuard(self.dense_grad_full[n:n + 1], self.cap_factor)

    def _lookup(self, keys, insert):
        flat = self.V._keys(keys)
        d = ops.unique(flat)
        _, rows_v, pos_v = self.V.lookup_rows(flat, insert=insert, dedup=d)
        _, rows_w, pos_w = self.W.lookup_rows(flat, insert=insert, dedup=d)
        return d, rows_v, pos_v, rows_w, pos_w

    # ---- key-sharded tables: the fixed-capacity exchange (mindrec_amd/wide_deep_shard.py) over two MapParameters ------------
    def shard_overflow(self):
        """Positions THIS rank dropped because an owner's bucket of the request message was full (host sync); 0 on one GPU."""
        return int(self._overflow.item()) if self.world > 1 else 0

    def check_shard_overflow(self):
        """Raises ShardCapacityError on every rank alike if any rank dropped positions since the last check (waits for the last
        step; train_step / predict poll by themselves -- OverflowGuard, mindrec_amd/wide_deep_shard.py)."""
        if self.world > 1:
            self._guard.probe()
            self._guard.poll(block=True)

    def _shard_lookup(self, keys, wts, train):
        """Raw keys to their owners, looked-up rows back.  Returns (vx [B, F, D] fp32 masked, linear [B], route state)."""
        B, Fd = keys.shape
        D, n = self.cfg.data_emb_dim, keys.numel()
        flat = self.V._keys(keys)
        cap = ops.shard_capacity(n, self.world, self.cap_factor)
        ns = self.world * cap
        req, slot_of_pos, pos_of_slot = ops.shard_route_slots(flat, wts, self.world, cap, hashed=True, overflow=self._overflow)
        recv_req = torch.empty_like(req)
        self.comm.all_to_all(recv_req, req)
        recv_keys, recv_wts = ops.shard_unpack_req(recv_req)              # padding slots carry key -1 (reserved: embedding.py:53)
        # the owner's MapTensorGet: every received position probes the index (duplicates and other ranks' copies of a key
        # welcome: one hit per key and step), new keys take the next rows in order of arrival with their default values;
        # V and W see the same keys in the same order, so they number their rows alike
        _, _, rows = self.V.lookup_rows(recv_keys, insert=True, train=train, skip_pad=True)
        _, _, rows_w = self.W.lookup_rows(recv_keys, insert=True, train=train, skip_pad=True)
        Dw, W = ops.shard_msg_words(D, torch.float32)
        ans = torch.empty((ns, W), dtype=torch.float32, device=self.device)
        ans[:, :D] = ops.gather_rows(self.V.values, rows, recv_wts)       # rows of padding slots (-1) read as zeros
        ans[:, D] = ops.gather_rows(self.W.values, rows_w, recv_wts).view(ns)
        back = torch.empty_like(ans)
        self.comm.all_to_all(back, ans)
        vx, wprod = ops.shard_unroute_slots(back, slot_of_pos, D, torch.float32)
        linear = wprod.view(B, Fd, 2)[..., 0].sum(dim=1)
        return vx.view(B, Fd, D), linear, {"pos_of_slot": pos_of_slot, "rows": rows, "recv_wts": recv_wts, "ns": ns}

    def _train_step_sharded(self, keys, wts, label):
        cfg = self.cfg
        B, Fd = keys.shape
        D = cfg.data_emb_dim
        vx, linear, route = self._shard_lookup(keys, wts, train=None)
        plan = ops.sparse_plan(route["rows"], skip_negative=True)         # Unique + inverted index of the received rows
        loss, g_vx, g_lin = self._net_step(vx, None, linear.contiguous(), label)   # (the rows arrive in fp32: the FM term wants them so)
        gmsg = ops.shard_route_grads(g_vx.view(B * Fd, D), g_lin.view(B).contiguous(), Fd, route["pos_of_slot"])
        recv_g = torch.empty_like(gmsg)
        self.comm.all_to_all(recv_g, gmsg)
        if self._mfma:
            self._sum_dw_slabs()
        self._guard.stage(self._overflow)                     # the dropped-position count rides the dense all-reduce
        self.comm.all_reduce(self.dense_grad_full)
        # gradients_mean: the owner sums the row gradients of all ranks, the mean divides by their number
        scale = 1.0 / (cfg.loss_scale * self.world)
        kw = self._adam_kw(scale)
        rows_u = plan.uniq_buf
        for t, g in ((self.V, recv_g[:, :D]), (self.W, recv_g[:, D:D + 1])):
            plan.uniq_buf = t.admitted_rows(rows_u)               # groups -> table rows, un-admitted keys -> -1 (skipped)
            ops.sparse_lazy_adam_(t.values, t.slots["moment1"]["table"], t.slots["moment2"]["table"], plan, g, route["recv_wts"], **kw)
        self._dense_adam(scale, summed=True)
        return loss.detach()

    def predict(self, keys, wts):
        B, Fd = keys.shape
        if self.world > 1:
            self._guard.poll()
        with torch.no_grad():
            if self.world > 1:
                vx, linear, _ = self._shard_lookup(keys, wts, train=False)      # (a collective: every rank calls predict)
            else:
                _, _, pos_v, _, pos_w = self._lookup(keys, insert=True)
                vx = ops.gather_rows(self.V.values, pos_v.view(B, Fd), wts)
                linear = ops.wide_sum(self.W.values, pos_w.view(B, Fd), wts)
            fm, _ = ops.fm_forward(vx)
            logit = (linear + fm).view(-1, 1) + self._mlp(vx.view(B, -1))
        return logit, torch.sigmoid(logit)

    def train_step(self, keys, wts, label):
        cfg = self.cfg
        B, Fd = keys.shape
        D = cfg.data_emb_dim
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self.step_count += 1
        if self.world > 1:
            self._guard.poll()                                # drops of the previous step: raised here, on every rank
            loss = self._train_step_sharded(keys, wts, label)
            self._guard.probe()
            return loss
        d, rows_v, pos_v, rows_w, pos_w = self._lookup(keys, insert=True)
        vx = ops.gather_rows(self.V.values, pos_v.view(B, Fd), wts)             # [B, F, D], mask fused
        linear = ops.wide_sum(self.W.values, pos_w.view(B, Fd), wts)             # [B]
        loss, g_vx, g_lin = self._net_step(vx, None, linear, label)
        kw = self._adam_kw(1.0 / cfg.loss_scale)
        plan = ops.group_by_inverse(d)
        for t, rows_u, g, scale in ((self.V, rows_v, g_vx.view(B * Fd, D), wts),
                                    (self.W, rows_w, (g_lin.view(B, 1) * wts).view(B * Fd, 1), None)):
            plan.uniq_buf = t.admitted_rows(rows_u)               # groups -> table rows, un-admitted keys -> -1 (skipped)
            ops.sparse_lazy_adam_(t.values, t.slots["moment1"]["table"], t.slots["moment2"]["table"], plan, g, scale, **kw)
        self._dense_adam(1.0 / cfg.loss_scale)
        return loss.detach()

    def evict(self):
        return self.V.evict(), self.W.evict()
