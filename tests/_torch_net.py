"""TEST INFRASTRUCTURE ONLY: torch restatements of the dense nets -- autograd over torch's own GEMMs -- as mixins for the
oracle-side engines (tests/_oracle_engine.py).  They implement the hooks the product engines leave raising UnsupportedNet
(`_mlp_generic`, `_head_generic`, `_mlp_step_generic`, `_forward_generic`, `_train_step_generic`, `_net_step_generic`): the
product has no library-GEMM and no autograd path; the CPU-side checker does, and so do GPU tests that need a net the kernels
do not cover (e.g. Dropout on the fp32 net)."""
import numpy as np
import torch
import torch.nn.functional as F


class TorchDenseNetMixin:
    """DenseLayer x n (models/wide_deep/src/wide_and_deep.py:113-133) + NetWithLossClass (:349-362) through autograd."""

    def _mlp_autograd(self, x):
        n = len(self.dims) - 1
        amp = self._amp
        h = x.to(amp) if amp is not None else x
        for i in range(n):
            W, b = self.dense[2 * i], self.dense[2 * i + 1]
            d = self._drop(i, h.shape[0])
            if d is not None:              # `x = self.dropout(x)`, :117-118 (autograd multiplies the gradient by the same mask)
                h = h * self.k.dropout_mask(h.shape[0], h.shape[1], d, self.device).to(h.dtype)
            if amp is not None and i < n - 1:
                h = torch.addmm(b.to(amp), h, W.to(amp))
            else:
                h = torch.addmm(b, h.float(), W)
            if i < n - 1:
                h = torch.relu(h)
        return h.float()

    def _mlp_generic(self, x):
        with torch.no_grad():
            return self._mlp_autograd(x)

    def _mlp_step_generic(self, emb, wide, label):
        with torch.enable_grad():
            emb = emb.detach().requires_grad_(True)
            wide = wide.detach().requires_grad_(True)
            self.dense_grad_flat.zero_()
            logit = wide.view(-1, 1) + self._mlp_autograd(emb)
            loss = F.binary_cross_entropy_with_logits(logit, label)      # SigmoidCrossEntropyWithLogits + ReduceMean
            (loss * self._sens).backward()                               # sens_param seeding, :479-486
        return loss.detach(), emb.grad, wide.grad

    def _head_generic(self, hs, wide, label, dhs):
        amp, n = self._amp, len(self.dims) - 1
        B = hs[0].shape[0]
        W5, b5 = self.dense[2 * (n - 1)].detach(), self.dense[2 * (n - 1) + 1].detach()
        h4 = hs[-1].float()
        logit = torch.addmm(b5, h4, W5) + wide.view(-1, 1)
        loss = F.binary_cross_entropy_with_logits(logit, label)
        dlogit = (torch.sigmoid(logit) - label) * (self._sens / B)          # d(sens * mean BCE)/d logit
        torch.mm(h4.t(), dlogit, out=self.dense_grad[2 * (n - 1)])
        torch.sum(dlogit, dim=0, out=self.dense_grad[2 * (n - 1) + 1])
        dh = torch.ops.aten.threshold_backward((torch.mm(dlogit, W5.t()) * dhs).to(amp), hs[-1], 0)
        torch.sum(dh, dim=0, dtype=torch.float32, out=self.dense_grad[2 * (n - 2) + 1])
        return {"hs": hs, "loss": loss, "g_wide": dlogit.view(-1), "dh": dh}


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


class TorchDeepCrossMixin:
    """DeepCrossModel.construct + TrainStepWrap (models/deep_and_cross/src/deep_and_cross.py:293-354) through autograd; the
    cross stack and the embedding ops still go through the engine's op set."""

    def _forward_autograd(self, emb):
        W1, b1, W2, b2, W3, b3, cw, cb = self.dense
        d1 = torch.relu(torch.addmm(b1, emb, W1))
        d2 = torch.relu(torch.addmm(b2, d1, W2))
        c = _CrossStack.apply(emb, cw, cb, self.k)
        h2 = d2.shape[1]
        return ((d2 * W3[:h2, 0]).sum(dim=1) + (c * W3[h2:, 0]).sum(dim=1)).view(-1, 1) + b3      # concat([deep, cross]) . W3 + b3

    def _forward_generic(self, emb):
        with torch.no_grad():
            return self._forward_autograd(emb)

    def _train_step_generic(self, ids, wts, label):
        cfg = self.cfg
        self.step_count += 1
        B, Fd = ids.shape
        D = cfg.emb_dim
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        emb = self.k.gather_rows(self.table, ids, wts).view(B, Fd * D)
        emb.requires_grad_(True)
        self.dense_grad_flat.zero_()
        logit = self._forward_autograd(emb)
        loss = F.binary_cross_entropy_with_logits(logit, label)
        (loss * cfg.loss_scale).backward()
        # dense table gradient = UnsortedSegmentSum of the masked row gradients (bprop of Gather)
        plan = self.k.sparse_plan(ids)
        sums = self.k.segment_sum(plan, emb.grad.view(B * Fd, D), wts)
        gtab = torch.zeros_like(self.table)
        self.k.scatter_unique_rows_(gtab, plan, sums)
        kw = dict(lr=cfg.learning_rate, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.eps,
                  beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=1.0 / cfg.loss_scale)
        self.k.dense_adam_(self.table, self.table_m, self.table_v, gtab, **kw)
        self.k.dense_adam_(self.dense_flat, self.dense_m, self.dense_v, self.dense_grad_flat, **kw)
        return loss.detach()


class _FMTerm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vx, k):
        ctx.k = k
        fm, cs = k.fm_forward(vx)
        ctx.save_for_backward(vx, cs)
        return fm

    @staticmethod
    def backward(ctx, dout):
        vx, cs = ctx.saved_tensors
        g = torch.zeros_like(vx)
        ctx.k.fm_backward_(g, vx, cs, dout.contiguous())
        return g, None


class TorchDeepFMMixin(TorchDenseNetMixin):
    """DeepFMModel.construct behind the lookups (models/deepfm/src/deepfm.py:215-237) through autograd."""

    def _net_step_generic(self, vx, linear, label):
        cfg = self.cfg
        B, Fd, D = vx.shape
        with torch.enable_grad():
            vx = vx.detach().requires_grad_(True)
            linear = linear.detach().requires_grad_(True)
            self.dense_grad_flat.zero_()
            fm = _FMTerm.apply(vx, self.k)
            logit = (linear + fm).view(-1, 1) + self._mlp_autograd(vx.view(B, Fd * D))
            loss = F.binary_cross_entropy_with_logits(logit, label)
            (loss * cfg.loss_scale).backward()
        return loss.detach(), vx.grad, linear.grad
