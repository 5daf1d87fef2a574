"""Deep&Cross (DCN-v1) training step on the MI355X kernels.

Mirrors models/deep_and_cross/src/deep_and_cross.py of the reference:
  EmbeddingLookup.construct   :188-203  dense Gather on a [vocab, emb_dim] table, N(0, 1/sqrt(dim)) init (:46-57)
  DeepCrossModel.construct    :293-309  gather -> mask multiply -> {DenseLayer x2 (ReLU), CrossLayer x6} ->
                                        concat -> DenseLayer -> logit; everything fp32 (convert_dtype=False)
  CrossLayer.construct        :139-149  y = x0 * (x_l . w) + b + x_l
  NetWithLossClass.construct  :326-331  sigmoid cross-entropy, mean
  TrainStepWrap               :331-354  one dense Adam(lr 1e-4, eps 1e-8, loss_scale 1000) over ALL parameters,
                                        the embedding table included (dense UnsortedSegmentSum gradient)

The six cross layers run as ONE HBM pass (mrec_cross_layers_f32 / _bwd_f32); the table's dense
gradient is the fused segment-sum of the row gradients scattered into a zero [V, D] buffer.  The two
hidden DenseLayers run in exact fp32 on the fp32-input matrix instruction (csrc/mrec_gemm_f32.hip:
forward with bias + ReLU epilogue, input gradient with the ReLU mask and the bias gradient's column sums in
its epilogue, weight gradients as fp32 batch slabs added up inside the dense Adam); the output layer over
[deep | cross] -- never concatenated --, the loss and every bprop that hangs off the logit are one pass
(csrc/mrec_dcn.hip).  No library GEMM and no autograd on the product path; with `graphs="step"` the whole
step -- Adam's bias-correction powers live in device memory -- replays as one HIP graph.  The oracle-side
engine (tests/) runs a torch restatement of the same math.
"""
import contextlib
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import ops


@dataclass
class DeepCrossConfig:
    """models/deep_and_cross/src/config.py:21-123 (argparse defaults)."""
    vocab_size: int = 200000
    emb_dim: int = 30
    field_size: int = 39
    batch_size: int = 16384
    deep_layer_dim: List[int] = field(default_factory=lambda: [1024, 1024])
    cross_layer_num: int = 6
    learning_rate: float = 1e-4
    eps: float = 1e-8
    loss_scale: float = 1000.0
    seed: int = 1000
    init_sigma: float = 0.01
    graphs: str = "step"          # "step": the whole training step replays as one HIP graph; "none": kernel by kernel
    fp32_matmul: str = "x3"       # the DenseLayers' fp32 MatMuls: "x3" = three-part bf16 operands on the 16-bit matrix instruction
                                  # (fp32-class accuracy at 6/16 of the fp32-MFMA time, csrc/mrec_gemm_x3.hip); "exact" = the
                                  # fp32-input matrix instruction (a k-ordered chain of fmaf per output, csrc/mrec_gemm_f32.hip)


def _flat_views(shapes, device):
    n = sum(int(np.prod(s)) for s in shapes)
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    views, off = [], 0
    for s in shapes:
        k = int(np.prod(s))
        views.append(flat[off:off + k].view(s))
        off += k
    return flat, views


class DeepCrossEngine:
    _kernels = ops               # the op set (tests/ subclass the engine with the oracle's restatements to check it step for step)
    _allow_cpu = False           # the product has no CPU path

    def __init__(self, cfg: DeepCrossConfig, device):
        self.cfg, self.device = cfg, torch.device(device)
        self.k = self._kernels
        self._gpu = self.device.type == "cuda"
        if not self._gpu and not self._allow_cpu:
            raise RuntimeError("DeepCrossEngine runs on an MI355X (no CPU fallback)")
        V, D, dev = cfg.vocab_size, cfg.emb_dim, self.device
        X = cfg.field_size * D
        with (torch.cuda.device(dev) if self._gpu else contextlib.nullcontext()):
            self.table = torch.empty((V, D), dtype=torch.float32, device=dev)
            self.k.fill_normal_(self.table, cfg.seed, 1.0 / float(np.sqrt(D)))      # normal_weight(shape, emb_dim)
            self.table_m = torch.zeros_like(self.table)
            self.table_v = torch.zeros_like(self.table)
            h1, h2 = cfg.deep_layer_dim
            # one flat buffer; every parameter starts on a 16-byte boundary (padding elements are parameters nobody reads:
            # gradient always 0), the whole a multiple of 4 floats: the dense Adam is one float4 launch
            shapes = [(X, h1), (h1,), (h1, h2), (h2,), (X + h2, 1), (1,), (cfg.cross_layer_num, X), (cfg.cross_layer_num, X)]
            padded, keep = [], []
            for sh in shapes:
                keep.append(len(padded))
                padded.append(sh)
                n_ = int(np.prod(sh))
                if n_ % 4:
                    padded.append((4 - n_ % 4,))
            self.dense_flat, views = _flat_views(padded, dev)
            self.dense = [views[i] for i in keep]
            self.dense_grad_flat, gviews = _flat_views(padded, dev)
            self.dense_grad = [gviews[i] for i in keep]
            self.dense_m = torch.zeros_like(self.dense_flat)
            self.dense_v = torch.zeros_like(self.dense_flat)
            self.k.fill_normal_(self.dense_flat.view(-1, 1), cfg.seed + 2, cfg.init_sigma)
            for p, g in zip(self.dense, self.dense_grad):
                p.requires_grad_(True)
                p.grad = g
        self.beta1, self.beta2 = np.float32(0.9), np.float32(0.999)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)
        self.step_count = 0
        h1, h2 = cfg.deep_layer_dim
        # the hand-written fp32 path: the product on the GPU whenever the output end fits its kernel
        self._native = bool(self._gpu and self.k is ops and ops.dcn_head_supported(h2, X) and (V * D) % 4 == 0)
        self._graph = None            # {"ids", "wts", "label", "graph", "loss"}: the captured step and its static inputs
        self._state = None            # ops.StepState: Adam's powers / step size in device memory (constant kernel arguments)
        self._state_step = -1
        self._bufs = {}               # batch size -> the step's persistent intermediates (graph replays write the same buffers)
        self._side = torch.cuda.Stream(self.device) if self._native else None      # the step's Unique + inverted index, under the GEMMs

    def forward(self, emb):
        """DeepCrossModel.construct behind the lookup (deep_and_cross.py:299-309), inference: logit [B, 1]."""
        if not self._native:
            return self._forward_generic(emb)
        W1, b1, W2, b2, W3, b3, cw, cb = [p.detach() for p in self.dense]
        d1 = self.k.dense32_fwd(emb, W1, b1, relu=True)
        d2 = self.k.dense32_fwd(d1, W2, b2, relu=True)
        c = self.k.cross_layers(emb, cw, cb)
        h2 = d2.shape[1]
        # concat([deep, cross]) . W3 + b3 (:306-308) on the two halves of W3, without materialising the [B, h2 + X] concat
        return self.k.dense32_fwd(d2, W3[:h2], None, relu=False) + self.k.dense32_fwd(c, W3[h2:], b3, relu=False)

    # ---- hooks: shapes without a HIP path.  The product refuses them; tests/_torch_net.py implements them for the oracle side ----
    def _unsupported(self, what):
        from .wide_deep_mlp import UnsupportedNet
        cfg = self.cfg
        return UnsupportedNet(f"MREC_EUNSUPPORTED: {what}: no hand-written HIP path for Deep&Cross with deep_layer_dim {cfg.deep_layer_dim}, "
                              f"input width {cfg.field_size * cfg.emb_dim}, table {cfg.vocab_size} x {cfg.emb_dim} on {self.device} (second hidden "
                              f"width a multiple of 4 and <= 1024, input width even and <= 1280, vocab_size * emb_dim a multiple of 4)")

    def _forward_generic(self, emb):
        raise self._unsupported("inference forward")

    def _train_step_generic(self, ids, wts, label):
        raise self._unsupported("training step")

    def predict(self, ids, wts):
        B, Fd = ids.shape
        emb = self.k.gather_rows(self.table, ids, wts).view(B, Fd * self.cfg.emb_dim)
        with torch.no_grad():
            logit = self.forward(emb)
        return logit, torch.sigmoid(logit)

    # ---- the hand-written step ------------------------------------------------------------------------------------------
    def _x3(self, B):
        """The DenseLayers' MatMuls on three-part bf16 operands (ops.x3_*) at this batch size?"""
        cfg, k = self.cfg, self.k
        X, (h1, h2) = cfg.field_size * cfg.emb_dim, cfg.deep_layer_dim
        return cfg.fp32_matmul == "x3" and k.x3_supported(B, X, h1) and k.x3_supported(B, h1, h2)

    def _buffers(self, B):
        """Intermediates of a step at batch B (allocated once: a captured step writes the same addresses every replay)."""
        b = self._bufs.get(B)
        if b is None:
            cfg, k, dev = self.cfg, self.k, self.device
            X = cfg.field_size * cfg.emb_dim
            h1, h2 = cfg.deep_layer_dim
            f32 = dict(dtype=torch.float32, device=dev)
            T = k.dense32_colsum_tiles(B)
            b = {"d1": torch.empty((B, h1), **f32), "d2": torch.empty((B, h2), **f32), "dd1": torch.empty((B, h1), **f32),
                 "g2": torch.empty((1, B, X), **f32),             # the MLP's input gradient; the cross stack's is added onto it
                 "head": {},
                 "dW1": torch.empty(((k.x3_wgrad_slabs if self._x3(B) else k.dense32_bwd_weight_slabs)(B, X, h1), X, h1), **f32),
                 "dW2": torch.empty(((k.x3_wgrad_slabs if self._x3(B) else k.dense32_bwd_weight_slabs)(B, h1, h2), h1, h2), **f32),
                 "db1": torch.empty((T, h1), **f32)}
            self._bufs[B] = b
        return b

    def _step_native(self, ids, wts, label):
        """One training step, kernel by kernel (every argument constant from step to step: capturable)."""
        cfg, k = self.cfg, self.k
        B, Fd = ids.shape
        D = cfg.emb_dim
        X = Fd * D
        W1, b1, W2, b2, W3, b3, cw, cb = [p.detach() for p in self.dense]
        gW1, gb1, gW2, gb2, gW3, gb3, gcw, gcb = self.dense_grad
        bf = self._buffers(B)
        self._state.advance(cfg.learning_rate, float(self.beta1), float(self.beta2))
        # The bprop of Gather needs the step's Unique + inverted index, which depend on nothing but the ids: a dozen small latency-bound
        # kernels (~140 us in a row) on a side branch under the GEMMs instead of behind them.  The fork is MARKED here and issued
        # behind the lookup: under capture the branch whose first node is created first stays on the launch queue, and that must be
        # the critical chain (mindrec_amd/wide_deep.py, _front)
        main = torch.cuda.current_stream()
        fork = main.record_event() if self._side is not None else None
        emb = k.gather_rows(self.table, ids, wts).view(B, X)
        plan = None
        if self._side is not None:
            self._side.wait_event(fork)
            with torch.cuda.stream(self._side):
                plan = k.sparse_plan(ids)
            if not torch.cuda.is_current_stream_capturing():          # (inside a capture every tensor lives in the graph's own pool)
                for t in (plan.uniq_buf, plan.inv, plan.n_uniq_dev, plan.sorted_pos, plan.sorted_seg, plan.seg_offsets):
                    t.record_stream(main)
        h1, h2 = cfg.deep_layer_dim
        x3 = self._x3(B)
        if x3:
            P = bf.get("x3")
            if P is None:
                P = bf["x3"] = {"emb": k.x3_parts(B, X, self.device), "W1": k.x3_parts(X, h1, self.device), "W2": k.x3_parts(h1, h2, self.device),
                                "d1": k.x3_parts(B, h1, self.device), "dd2": k.x3_parts(B, h2, self.device), "dd1": k.x3_parts(B, h1, self.device)}
            k.x3_split(emb, out=P["emb"])
            k.x3_split(W1, out=P["W1"])
            k.x3_split(W2, out=P["W2"])
            d1 = k.x3_fwd(P["emb"], P["W1"], B, X, h1, bf["d1"], bias=b1, relu=True, parts_out=P["d1"])
            d2 = k.x3_fwd(P["d1"], P["W2"], B, h1, h2, bf["d2"], bias=b2, relu=True)
        else:
            d1 = k.dense32_fwd(emb, W1, b1, relu=True, out=bf["d1"])
            d2 = k.dense32_fwd(d1, W2, b2, relu=True, out=bf["d2"])
        c = k.cross_layers(emb, cw, cb)
        loss, _, dd2, dc = k.dcn_head_fwd_bwd(d2, c, W3.view(-1), b3, label.view(-1), cfg.loss_scale / B, gW3.view(-1), gb2, gb3,
                                              out=bf["head"])
        if x3:
            k.x3_split(dd2, out=P["dd2"])
            k.x3_gemm(2, P["d1"], P["dd2"], B, h1, h2, bf["dW2"], S=bf["dW2"].shape[0])
            k.x3_dgrad(P["dd2"], P["W2"], B, h1, h2, bf["dd1"], h=d1, colsum=bf["db1"], parts_out=P["dd1"])
            k.x3_gemm(2, P["emb"], P["dd1"], B, X, h1, bf["dW1"], S=bf["dW1"].shape[0])
            k.x3_dgrad(P["dd1"], P["W1"], B, X, h1, bf["g2"][0])
        else:
            k.dense32_bwd_weight(d1, dd2, bf["dW2"])
            dd1 = k.dense32_bwd_input(dd2, W2, h=d1, out=bf["dd1"], colsum=bf["db1"])
            k.dense32_bwd_weight(emb, dd1, bf["dW1"])
            k.dense32_bwd_input(dd1, W1, out=bf["g2"][0])
        # the embeddings feed the deep net and the cross stack (:300-306): the cross stack's input gradient is added onto the deep net's.
        # (Measured and dropped: the cross stack -- HBM-bound -- on a second side branch beside the matrix-bound GEMMs, forward beside the
        # two forward GEMMs, backward beside the two weight-gradient GEMMs with the deep net's input gradient computed first: 1.816 ->
        # 1.957 ms on the three-part GEMMs, 2.324 -> 2.529 on the exact ones; every extra branch costs this graph runtime more than its
        # overlap returns, as in the Wide&Deep step.)
        g = bf["g2"][0]
        k.cross_layers_bwd(emb, cw, cb, dc, dx0_out=g, dw_out=gcw, db_out=gcb, accumulate=True)
        # dense table gradient = UnsortedSegmentSum of the masked row gradients (bprop of Gather)
        if plan is None:
            plan = k.sparse_plan(ids)
        else:
            main.wait_stream(self._side)
        sums = k.segment_sum(plan, g.view(B * Fd, D), wts)
        kw = dict(lr=cfg.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.eps, beta1_power=0.0, beta2_power=0.0,
                  grad_scale=1.0 / cfg.loss_scale, step_state=self._state)
        # nn.Adam over the whole table; its gradient (the bprop of Gather) is nonzero on the touched rows only: looked up per row
        k.dense_adam_rows_l2_(self.table, self.table_m, self.table_v, plan, sums, **kw)
        slabs = [(gW1.storage_offset(), bf["dW1"]), (gW2.storage_offset(), bf["dW2"]), (gb1.storage_offset(), bf["db1"])]
        k.dense_adam_slabs_(self.dense_flat.detach(), self.dense_m, self.dense_v, self.dense_grad_flat, slabs, **kw)
        return loss.view(())

    def close(self):
        """Drops the captured step (nothing of it outlives the engine; symmetrical with WideDeepEngine.close)."""
        self._graph = None

    def _train_step_native(self, ids, wts, label):
        if self._state is None:
            self._state = self.k.StepState(self.device)
        if self._state_step != self.step_count:
            self._state.reset(self.beta1_power, self.beta2_power, self.step_count)
        self.step_count += 1
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._state_step = self.step_count
        if self.cfg.graphs != "step" or self.step_count <= 2:        # (the first steps run eagerly: workspaces come into being)
            return self._step_native(ids, wts, label)
        g = self._graph
        if g is None or g["ids"].shape != ids.shape or g["ids"].dtype != ids.dtype:
            g = {"ids": ids.clone(), "wts": wts.clone(), "label": label.clone()}
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                g["loss"] = self._step_native(g["ids"], g["wts"], g["label"])
            g["graph"] = graph
            self._graph = g
        self.k.copy3_((g["ids"], g["wts"], g["label"]), (ids, wts, label))
        g["graph"].replay()
        return g["loss"]

    def train_step(self, ids, wts, label):
        if self._native:
            return self._train_step_native(ids, wts, label)
        return self._train_step_generic(ids, wts, label)
