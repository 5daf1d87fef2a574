"""TEST INFRASTRUCTURE ONLY: the Wide&Deep training step in MIXED PRECISION restated on the CPU (numpy + the C oracle),
with a 16-bit rounding applied exactly where the MI355X path rounds -- so that the step bench.py measures (16-bit
looked-up rows, 16-bit MLP on the matrix cores, 16-bit row gradients, fp32 tables and optimizers) is checked against an
oracle and not only against itself.

Reference being restated: TrainStepWrap.construct, models/wide_deep/src/wide_and_deep.py:472-492, in its sparse
configuration with use_mixed_precision (DenseLayer casts input and weight to 16 bits, :119-128); hyper-parameters
:415-433.  Rounding points (dt = "bf16" | "f16"; PARITY UNPINNED at the MindSpore boundary like the rest of the oracle):
  1. looked-up rows       emb = round16(table[id] * wt)                       (fp32 product, one rounding)
  2. hidden layer i       h = round16(relu(h . round16(W_i) + b_i))           (exact products, b fp32, one rounding)
  3. output head          fp32 on 16-bit h4: logit, loss, dlogit; dh4 = round16(dlogit * w5) where h4 > 0
  4. input gradients      dh = round16(dh . round16(W_i)^T) masked by h > 0   (bias gradient = sum of the rounded dh)
  5. weight gradients     dW_i = h^T . dh in full precision (never rounded)
  6. row gradients        g_emb is 16-bit; the sparse apply widens it exactly and works in fp32 (oracle C code)
  7. operand shadow       the next step's W16 = round16(updated fp32 W)
  8. Dropout (cfg.dropout_flag; :117-118, on every DenseLayer's input while training): the stored 16-bit input is replaced by
     round16(x * mask / keep); the input gradients carry the same factor before their rounding (oracle.dropout_mask)
Sums the GPU takes in fp32 are taken in float64 here: the oracle is at least as exact as the device.

fast=True (the long statistical runs: AUC over hundreds of steps at the bench shape, where a float64 numpy step takes ~20 s):
the same restatement with the three GEMMs of every layer taken by torch's CPU BLAS in fp32 -- the precision of the device's
accumulators, in the host library's summation order -- and the 16-bit roundings by torch's casts (round-to-nearest-even, as
oracle.round16).  tests/test_bench_shape_gpu.py checks one fast step against the float64 one."""
import numpy as np
import torch

from oracle import oracle as O

_T16 = {"bf16": torch.bfloat16, "f16": torch.float16}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


class _Fast:
    """oracle.dense_layer / dense_bwd_input / dense_bwd_weight / round16 on torch-CPU fp32 (see the module docstring)."""

    @staticmethod
    def round16(x, dt):
        return _t(x).to(_T16[dt]).float().numpy()

    @staticmethod
    def dense_layer(x16, w16, bias, relu, dt):
        acc = torch.addmm(_t(bias), _t(x16), _t(w16)) if bias is not None else _t(x16) @ _t(w16)
        if relu:
            acc.relu_()
        return acc.to(_T16[dt]).float().numpy()

    @staticmethod
    def dense_bwd_input(dy16, w16, h16, dt, scale=1.0, mask=None):
        g = _t(dy16) @ _t(w16).t()
        if scale != 1.0:
            g *= float(np.float32(scale))
        g = g.to(_T16[dt]).float()
        if h16 is not None:
            g = torch.where(_t(h16) > 0, g, torch.zeros((), dtype=torch.float32))
        elif mask is not None:
            g = torch.where(_t(mask) > 0, g, torch.zeros((), dtype=torch.float32))
        return g.numpy(), g.sum(dim=0, dtype=torch.float64).numpy()

    @staticmethod
    def dense_bwd_weight(x16, dy16):
        return (_t(x16).t() @ _t(dy16)).numpy()


class OracleMixedEngine:
    def __init__(self, cfg, dt, fast=False):
        self.cfg, self.dt = cfg, dt
        self._g = _Fast if fast else O              # who takes the GEMMs and the roundings
        V, D = cfg.vocab_size, cfg.emb_dim
        self.deep = O.fill_normal(cfg.seed, V, D, cfg.init_sigma)
        self.deep_m = np.zeros_like(self.deep); self.deep_v = np.zeros_like(self.deep)
        self.wide = O.fill_normal(cfg.seed + 1, V, 1, cfg.init_sigma)
        self.wide_accum = np.full_like(self.wide, cfg.ftrl_initial_accum); self.wide_linear = np.zeros_like(self.wide)
        dims = [cfg.field_size * D] + list(cfg.deep_layer_dim) + [1]
        self.dims = dims
        nl = len(dims) - 1
        shapes_h = [(dims[i], dims[i + 1]) for i in range(nl - 1)]
        shapes_s = [(dims[i + 1],) for i in range(nl - 1)] + [(dims[nl - 1], dims[nl]), (dims[nl],)]
        n_real = sum(int(np.prod(s)) for s in shapes_h + shapes_s)
        n = n_real + ((-n_real - 1) % 4) + 1          # + "Wide_b" right behind the net's parameters, then padding to 4
        self._wb_off = n_real
        self.flat = np.zeros(n, np.float32)
        self.flat[:] = O.fill_normal(cfg.seed + 2, n, 1, cfg.init_sigma).ravel()          # same keyed init as the engine
        self.m = np.zeros(n, np.float32); self.v = np.zeros(n, np.float32)
        offs, o = [], 0
        for s in shapes_h + shapes_s:
            offs.append((o, s)); o += int(np.prod(s))
        self.W = [self.flat[offs[i][0]: offs[i][0] + int(np.prod(offs[i][1]))].reshape(offs[i][1]) for i in range(nl - 1)]
        self.b = [self.flat[offs[nl - 1 + i][0]: offs[nl - 1 + i][0] + dims[i + 1]] for i in range(nl - 1)]
        o5 = offs[2 * (nl - 1)]
        self.w5 = self.flat[o5[0]: o5[0] + dims[nl - 1]]
        self.b5 = self.flat[offs[2 * (nl - 1) + 1][0]: offs[2 * (nl - 1) + 1][0] + 1]
        self._offs = offs
        # the wide bias belongs to the FTRL optimizer: by the time TrainStepWrap tests `"wide" in params.name` (wide_and_deep.py:407-411)
        # MindSpore has renamed the Parameter held in the attribute `wide_b` to "<prefix>.wide_b" [EXT]; pinned by the fixtures the
        # reference's own TrainStepWrap produced (tests/golden/ref_wd_*.npz).  It keeps its slot in the flat buffer; its m / v words
        # are FTRL's accum / linear.
        self.wide_b = self.flat[n_real: n_real + 1]
        self.wide_b[:] = O.fill_normal(cfg.seed + 3, 1, 1, cfg.init_sigma).ravel()
        self._wb_ftrl = getattr(cfg, "wide_b_optimizer", "ftrl") == "ftrl"
        if self._wb_ftrl:
            self.m[n_real] = cfg.ftrl_initial_accum
        self.b1p = np.float32(1.0); self.b2p = np.float32(1.0)
        self.t = 0                                    # 0-based index of the training step (keys the Dropout masks)

    def _mask(self, layer, B, row0=0):
        cfg = self.cfg
        if not (getattr(cfg, "dropout_flag", False) and cfg.dropout_keep_prob < 1.0):
            return None
        return O.dropout_mask(B, self.dims[layer], cfg.seed + 4, self.t, layer, cfg.dropout_keep_prob, row0)

    def forward_backward(self, ids, wts, label):
        """Everything of a step in front of the optimizers; returns a dict of every intermediate."""
        cfg, dt = self.cfg, self.dt
        B, Fd = ids.shape
        nl = len(self.dims) - 1
        r = {}
        G = self._g
        emb = G.round16(O.gather_rows(self.deep, ids, wts, threads=8).reshape(B, -1), dt)   # 1.
        wide = O.wide_sum(self.wide, ids, wts, float(self.wide_b[0]))
        W16 = [G.round16(w, dt) for w in self.W]
        masks = [self._mask(i, B) for i in range(nl)]                                      # 8.
        drop = masks[0] is not None
        scale = float(np.float32(1.0) / np.float32(cfg.dropout_keep_prob)) if drop else 1.0
        if drop:
            emb = O.dropout(emb, masks[0], dt)
        hs = [emb]
        for i in range(nl - 1):                                                            # 2.
            h = G.dense_layer(hs[i], W16[i], self.b[i], True, dt)
            hs.append(O.dropout(h, masks[i + 1], dt) if drop else h)
        head = O.head_fwd_bwd(hs[-1], self.w5, float(self.b5[0]), wide, label, cfg.sens / B, dh_scale=scale)   # 3. (float64 inside)
        dh = G.round16(head["dh4"].astype(np.float32), dt)
        r.update(emb=emb, wide=wide, hs=hs, loss=float(head["loss"]), dlogit=head["dlogit"].astype(np.float32), dh_top=dh)
        gW, gb = [None] * (nl - 1), [None] * (nl - 1)
        gb[nl - 2] = head["db4"]                                                           # sum of the UNrounded dh4 (head kernel)
        for i in range(nl - 2, 0, -1):
            gW[i] = G.dense_bwd_weight(hs[i], dh)                                          # 5.
            dh, gb[i - 1] = G.dense_bwd_input(dh, W16[i], hs[i], dt, scale=scale)          # 4.
        gW[0] = G.dense_bwd_weight(hs[0], dh)
        g_emb, _ = G.dense_bwd_input(dh, W16[0], None, dt, scale=scale, mask=masks[0])
        r.update(gW=gW, gb=gb, gw5=head["dw5"], gb5=head["db5"], g_emb=g_emb)
        return r

    def apply(self, ids, wts, r, g_emb=None):
        """The optimizers.  g_emb: row gradients to apply (default: the oracle's own; the kernel-level check feeds the GPU's)."""
        cfg = self.cfg
        B, Fd = ids.shape
        D = cfg.emb_dim
        nl = len(self.dims) - 1
        self.b1p = np.float32(self.b1p * np.float32(0.9)); self.b2p = np.float32(self.b2p * np.float32(0.999))
        self.t += 1
        inv = 1.0 / cfg.sens
        g = (r["g_emb"] if g_emb is None else g_emb).reshape(B * Fd, D)
        O.sparse_lazy_adam(self.deep, self.deep_m, self.deep_v, ids, g, wts, lr=cfg.adam_lr, eps=cfg.adam_eps,
                           b1_pow=float(self.b1p), b2_pow=float(self.b2p), grad_scale=inv, threads=8)
        gw = np.repeat(r["dlogit"].reshape(B, 1), Fd, axis=1).reshape(B * Fd, 1)
        O.sparse_ftrl(self.wide, self.wide_accum, self.wide_linear, ids, gw, wts, lr=cfg.ftrl_lr, l1=cfg.ftrl_l1, l2=cfg.ftrl_l2,
                      grad_scale=inv, threads=8)
        grad = np.zeros_like(self.flat)
        for i in range(nl - 1):
            o, s = self._offs[i]
            grad[o: o + int(np.prod(s))] = r["gW"][i].astype(np.float32).ravel()
            o, s = self._offs[nl - 1 + i]
            grad[o: o + s[0]] = np.asarray(r["gb"][i], np.float64).astype(np.float32)
        o, s = self._offs[2 * (nl - 1)]
        grad[o: o + s[0]] = r["gw5"].astype(np.float32)
        o, s = self._offs[2 * (nl - 1) + 1]
        grad[o] = np.float32(r["gb5"])
        grad[self._wb_off] = np.float32(r["gb5"])           # d loss / d Wide_b = sum of dlogit
        self.last_dense_grad = grad
        i = self._wb_off
        keep = [a[i:i + 1].copy() for a in (self.flat, self.m, self.v)]
        O.dense_adam(self.flat, self.m, self.v, grad, lr=cfg.adam_lr, eps=cfg.adam_eps, b1_pow=float(self.b1p), b2_pow=float(self.b2p),
                     grad_scale=inv)
        if self._wb_ftrl:
            O.dense_ftrl(keep[0], keep[1], keep[2], grad[i:i + 1].copy(), lr=cfg.ftrl_lr, l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, grad_scale=inv)
            self.flat[i], self.m[i], self.v[i] = keep[0][0], keep[1][0], keep[2][0]


    def predict(self, ids, wts):
        """PredictWithSigmoid (wide_and_deep.py:495-518) on the engine's inference path: 16-bit hidden layers, fp32 output layer."""
        G, dt = self._g, self.dt
        B = ids.shape[0]
        nl = len(self.dims) - 1
        h = G.round16(O.gather_rows(self.deep, ids, wts, threads=8).reshape(B, -1), dt)
        for i in range(nl - 1):
            h = G.dense_layer(h, G.round16(self.W[i], dt), self.b[i], True, dt)
        logit = h.astype(np.float64) @ self.w5.astype(np.float64) + float(self.b5[0]) + O.wide_sum(self.wide, ids, wts, float(self.wide_b[0]))
        return logit, 1.0 / (1.0 + np.exp(-logit))

    def train_step(self, ids, wts, label):
        r = self.forward_backward(ids, wts, label)
        self.apply(ids, wts, r)
        return r["loss"]
