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
gradient is the fused segment-sum of the row gradients scattered into a zero [V, D] buffer; the two
hidden DenseLayers and the output layer are GEMMs (hipBLASLt through torch).
"""
import contextlib
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

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


def _flat_views(shapes, device):
    n = sum(int(np.prod(s)) for s in shapes)
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    views, off = [], 0
    for s in shapes:
        k = int(np.prod(s))
        views.append(flat[off:off + k].view(s))
        off += k
    return flat, views


class _CrossStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, w, b, k):
        ctx.k = k
        ctx.save_for_backward(x0, w, b)
        return k.cross_layers(x0, w, b)

    @staticmethod
    def backward(ctx, dy):
        x0, w, b = ctx.saved_tensors
        dx0, dw, db = ctx.k.cross_layers_bwd(x0, w, b, dy.contiguous())
        return dx0, dw, db, None


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
            shapes = [(X, h1), (h1,), (h1, h2), (h2,), (X + h2, 1), (1,), (cfg.cross_layer_num, X), (cfg.cross_layer_num, X)]
            self.dense_flat, self.dense = _flat_views(shapes, dev)
            self.dense_grad_flat, self.dense_grad = _flat_views(shapes, dev)
            self.dense_m = torch.zeros_like(self.dense_flat)
            self.dense_v = torch.zeros_like(self.dense_flat)
            self.k.fill_normal_(self.dense_flat.view(-1, 1), cfg.seed + 2, cfg.init_sigma)
            for p, g in zip(self.dense, self.dense_grad):
                p.requires_grad_(True)
                p.grad = g
        self.beta1, self.beta2 = np.float32(0.9), np.float32(0.999)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)
        if self._gpu:
            from .wide_deep import enable_tuned_gemms
            enable_tuned_gemms()                     # shipped GEMM selections (tools/tune_gemms.py), tuning off

    def forward(self, emb):
        W1, b1, W2, b2, W3, b3, cw, cb = self.dense
        d1 = torch.relu(torch.addmm(b1, emb, W1))
        d2 = torch.relu(torch.addmm(b2, d1, W2))
        c = _CrossStack.apply(emb, cw, cb, self.k)
        # concat([deep, cross]) . W3 (deep_and_cross.py:306-308) on the halves of W3, without materialising the
        # [B, 2194] concat.  An N = 1 product is a GEMV: through the GEMM library it ran at 50-130 us per call
        # (forward and both backward products); as a broadcast multiply + row sum it is a bandwidth-bound pass.
        h2 = d2.shape[1]
        return ((d2 * W3[:h2, 0]).sum(dim=1) + (c * W3[h2:, 0]).sum(dim=1)).view(-1, 1) + b3

    def predict(self, ids, wts):
        B, Fd = ids.shape
        emb = self.k.gather_rows(self.table, ids, wts).view(B, Fd * self.cfg.emb_dim)
        with torch.no_grad():
            logit = self.forward(emb)
        return logit, torch.sigmoid(logit)

    def train_step(self, ids, wts, label):
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.emb_dim
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        emb = self.k.gather_rows(self.table, ids, wts).view(B, Fd * D)
        emb.requires_grad_(True)
        self.dense_grad_flat.zero_()
        logit = self.forward(emb)
        loss = F.binary_cross_entropy_with_logits(logit, label)
        (loss * cfg.loss_scale).backward()
        # dense table gradient = UnsortedSegmentSum of the masked row gradients (bprop of Gather)
        plan = self.k.sparse_plan(ids)
        sums = self.k.segment_sum(plan, emb.grad.view(B * Fd, D), wts)
        gtab = torch.zeros_like(self.table)
        self.k.scatter_unique_rows_(gtab, plan, sums)
        kw = dict(lr=cfg.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.eps,
                  beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                  grad_scale=1.0 / cfg.loss_scale)
        self.k.dense_adam_(self.table, self.table_m, self.table_v, gtab, **kw)
        self.k.dense_adam_(self.dense_flat, self.dense_m, self.dense_v, self.dense_grad_flat, **kw)
        return loss.detach()
