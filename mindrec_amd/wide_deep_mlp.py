"""The dense net of the Wide&Deep engine: DenseLayer x5 (models/wide_deep/src/wide_and_deep.py:113-133, :164-205) forward and
backward on the hand-written MFMA kernels (csrc/mrec_dense.hip, csrc/mrec_tail.hip), Dropout on the layer inputs, the HIP graphs
of the MLP step.  A mixin of WideDeepEngine (mindrec_amd/wide_deep.py), which owns the state these methods work on."""
import numpy as np
import torch

from . import ops


class UnsupportedNet(NotImplementedError):
    """MREC_EUNSUPPORTED at the engine level: the requested dense net has no hand-written HIP path.  The product never falls back
    to a library GEMM or to autograd; the torch restatements of these nets live under tests/ (tests/_torch_net.py), where the
    oracle-side engines use them."""


def _flat_views(shapes, device, dtype=torch.float32):
    n = sum(int(np.prod(s)) for s in shapes)
    flat = torch.zeros(n, dtype=dtype, device=device)
    views, off = [], 0
    for s in shapes:
        k = int(np.prod(s))
        views.append(flat[off:off + k].view(s))
        off += k
    return flat, views


class _WideProd:
    """The wide branch as its per-field products [B, F] (written by the fused lookup, summed inside the output head)."""

    def __init__(self, prod):
        self.prod = prod


class DenseNetMixin:
    """Needs from its host class: self.k (op set), self.device, self._gpu, self._amp (torch dtype of the 16-bit net, None: fp32),
    self._mfma, self._graph_level, self.rank, self.step_count, self._step_state, self._dropout / self._training / self._emb_dropped,
    self.cfg.dropout_keep_prob / .seed (Dropout only)."""

    def _init_dense_net(self, dims, seed, init_sigma, sens, extra_seed=None, fused_tail=True):
        """The dense net's state: fp32 master weights in one flat buffer, same for gradients / Adam moments.  Flat order: the
        hidden layers' weight matrices first (the 16-bit GEMM operands), then the biases and the fp32 output layer; `self.dense`
        lists them in layer order W0, b0, W1, b1, ...  One EXTRA scalar parameter rides behind them (self.extra_p / extra_g: the
        Wide&Deep model's "Wide_b"), then padding to a multiple of 4 floats: the dense Adam is one float4 launch (a 1-element
        tail launch cost 5 us + a gap every step; the pad elements are parameters nobody reads, gradient always 0).
        sens: the loss scale the backward is seeded with (TrainStepWrap(sens=...), wide_and_deep.py:390,479-486)."""
        dev = self.device
        self.dims, self._sens = list(dims), float(sens)
        nl = len(dims) - 1
        shapes_h = [(dims[i], dims[i + 1]) for i in range(nl - 1)]
        shapes_s = [(dims[i + 1],) for i in range(nl - 1)] + [(dims[nl - 1], dims[nl]), (dims[nl],)]
        self.n_h = sum(int(np.prod(x)) for x in shapes_h)

        def interleave(vh, vs):
            out = []
            for i in range(nl - 1):
                out += [vh[i], vs[i]]
            return out + [vs[nl - 1], vs[nl]]

        n_real = sum(int(np.prod(x)) for x in shapes_h + shapes_s)
        pad = [(((-n_real - 1) % 4) + 1,)]
        self._wb_off = n_real
        self.dense_flat, views = _flat_views(shapes_h + shapes_s + pad, dev)
        views = views[:len(shapes_h + shapes_s)]
        self.dense = interleave(views[:nl - 1], views[nl - 1:])
        # the gradient buffer carries 4 more floats than the parameters: [n] is the row shards' dropped-position count, summed over
        # the ranks by the dense all-reduce it rides on (OverflowGuard, wide_deep_shard.py); the optimizers see [:n]
        self.dense_grad_full, gviews = _flat_views(shapes_h + shapes_s + pad + [(4,)], dev)
        self.dense_grad_flat = self.dense_grad_full[:self.dense_flat.numel()]
        gviews = gviews[:len(shapes_h + shapes_s)]
        self.dense_grad = interleave(gviews[:nl - 1], gviews[nl - 1:])
        self.dense_m = torch.zeros_like(self.dense_flat)
        self.dense_v = torch.zeros_like(self.dense_flat)
        # identical on every rank: global row 0.. of a [n, 1] "table" keyed by a private seed
        self.k.fill_normal_(self.dense_flat.view(-1, 1), seed, init_sigma)
        for p, g in zip(self.dense, self.dense_grad):
            p.requires_grad_(True)
            p.grad = g
        self.extra_p = self.dense_flat.detach()[self._wb_off:self._wb_off + 1]
        self.extra_g = self.dense_grad_flat[self._wb_off:self._wb_off + 1]
        if extra_seed is not None:
            self.k.fill_normal_(self.extra_p.view(1, 1), extra_seed, init_sigma)
        else:
            self.extra_p.zero_()
        self.dense16 = None
        if self._mfma:
            # 16-bit shadow of every dense parameter, kept current by the dense-Adam kernel: the hidden layers'
            # [in, out] weight matrices in it are the GEMM operands of the forward AND of the input-gradient kernel
            flat16, v16 = _flat_views(shapes_h + shapes_s + pad, dev, self._amp)
            v16 = v16[:len(shapes_h + shapes_s)]
            flat16.copy_(self.dense_flat.detach())
            self.dense16_flat, self.dense16 = flat16, interleave(v16[:nl - 1], v16[nl - 1:])
        # The last two hidden layers + the output head + their input-gradient bprops as ONE launch (ops.tail_fwd_bwd) where the
        # net ends ... -> 512 -> 256 -> 128 -> 1 (the reference's) and the batch is a multiple of 64; any other net: layer by layer.
        # The fp32 net (mlp_dtype "fp32": DenseLayer without casts) by hand as well: hidden layers on the fp32 matrix instruction
        # (ops.dense32_*), the output end as one pass (ops.head_fwd_bwd on fp32 activations); Dropout keeps the torch restatement.
        self._f32net = bool(self._gpu and self.k is ops and not self._mfma and getattr(self, "_amp", None) is None and nl >= 2
                            and (self.k.head_supported(dims[nl - 1]) or self.k.dcn_head_supported(dims[nl - 1], 2)))
        self._tail_packed, self._dense16_t = None, None
        self._tail_ok = bool(self._mfma and fused_tail and nl >= 4 and self.k.tail_supported(64, *self.dims[nl - 3:nl]))
        self._mlp_graph = None        # dict: captured MLP step + its static input / output tensors
        self._dw = {}                 # hidden layer -> fp32 batch slabs [S, in, out] of its weight gradient (persistent:
                                      # graph replays and eager steps write the same buffers, the dense Adam reads them)
        self._db = {}                 # hidden layer -> fp32 partial sums of its bias gradient (same idea)
        self._dw_batch = None
        self._tail_out = {}           # (batch, dtype, slot) -> the tail launch's output tensors (persistent: graph replays write them)
        self._slot = 0
        self._refresh_tail()

    def load_dense_parameters(self, weights, biases, extra=None):
        """Overwrites the dense net's fp32 master parameters -- weights[i] [in_i, out_i] and biases[i] [out_i] in layer order, the
        layout of DenseLayer.weight / .bias (wide_and_deep.py:92-93) -- and, optionally, the extra scalar (`wide_b`), then refreshes
        every copy derived from them (16-bit operand shadow, transposes, the tail launch's fragment order).  Optimizer state is left
        alone.  How a checkpoint, a fixture or a `mindspore` Cell's parameters get into the engine."""
        n = len(self.dims) - 1
        if len(weights) != n or len(biases) != n:
            raise ValueError(f"load_dense_parameters: {n} layers expected, got {len(weights)} weights and {len(biases)} biases")
        with torch.no_grad():
            for i in range(n):
                W, b = self.dense[2 * i], self.dense[2 * i + 1]
                w_src = torch.as_tensor(weights[i], dtype=torch.float32).reshape(W.shape)
                b_src = torch.as_tensor(biases[i], dtype=torch.float32).reshape(b.shape)
                W.copy_(w_src)
                b.copy_(b_src)
            if extra is not None:
                self.extra_p.copy_(torch.as_tensor(extra, dtype=torch.float32).reshape(1))
            if self.dense16 is not None:
                self.dense16_flat.copy_(self.dense_flat.detach())
        self._refresh_tail()

    @staticmethod
    def mfma_net_ok(dims):
        """Shapes the hand-written 16-bit net covers: at least one hidden layer, every width a multiple of 8 (16-byte rows)."""
        return len(dims) - 1 >= 2 and all(d % 8 == 0 for d in dims[:-1])

    def _slab_segments(self):
        """[(offset in the flat gradient, slab tensor)] of the weight- and bias-gradient slabs of the last backward."""
        return ([(self.dense_grad[2 * i].storage_offset(), t) for i, t in sorted(self._dw.items())] +
                [(self.dense_grad[2 * i + 1].storage_offset(), t) for i, t in sorted(self._db.items())])

    def _sum_dw_slabs(self):
        """Weight- and bias-gradient slabs -> the flat gradient buffer, one launch (needed only where somebody other than the
        dense Adam reads the summed gradient: the data-parallel all-reduce, the dense-gradient mode)."""
        self.k.sum_slab_segments_(self.dense_grad_flat, self._slab_segments())

    def _drop(self, layer, B):
        """Dropout descriptor of DenseLayer `layer`'s input for the training step in flight (None: no dropout).  The mask is a
        function of (seed, step, layer, global sample row, column): on the GPU the step is read from the device-side step
        state (a captured step replays with a moving step), row0 makes N data-parallel ranks draw the mask of one big batch."""
        if not (self._dropout and self._training):
            return None
        return self.k.Dropout(self.cfg.dropout_keep_prob, self.cfg.seed + 4, layer, step=self.step_count - 1, row0=self.rank * B,
                              step_state=self._step_state if self._gpu and self.k is ops else None)

    def _refresh_tail(self):
        """Derived copies of the 16-bit weights, rewritten (one launch) whenever the shadow changes: the transposes [out, in] the
        forward GEMMs read (both operands K-contiguous: 15 % faster than W as stored through transposing LDS reads) and the tail
        kernel's fragment-ordered weights."""
        if not self._mfma:
            return
        n = len(self.dims) - 1
        hidden = range(n - 3 if self._tail_ok else n - 1)          # the layers that run as GEMM launches of their own
        if self._dense16_t is None:
            self._dense16_t = {i: torch.empty((self.dims[i + 1], self.dims[i]), dtype=self._amp, device=self.device) for i in hidden}
            if self._tail_ok:
                self._tail_packed = torch.empty(2 * (self.dims[n - 3] * self.dims[n - 2] + self.dims[n - 2] * self.dims[n - 1]),
                                                dtype=self._amp, device=self.device)
        tr = [(self.dense16[2 * i], self._dense16_t[i]) for i in hidden]
        tail = (self.dense16[2 * (n - 3)], self.dense16[2 * (n - 2)], self._tail_packed) if self._tail_ok else None
        for k in range(0, max(len(tr), 1), 4):                      # (at most 4 transposes per launch; the reference's net: 2)
            self.k.operand_copies(tr[k:k + 4], tail if k == 0 else None)

    # (Round 5, built and measured: the tail launch's small second kernel -- the partial sums: loss, dw5, bias gradients, which feed only
    # the dense optimizer -- on the step's side stream behind an event instead of on the chain in front of the backward launches:
    # the captured step 0.629 -> 0.742 ms.  The graph runtime puts the chain's next kernels behind what the branch holds (the plan's
    # launches): the same finding as the weight gradients on a branch, _mlp_bwd.  It stays on the chain: 5.6 us.)
    def _tail_now(self, B):
        return bool(self._tail_ok and self.k.tail_supported(B, *self.dims[len(self.dims) - 4:len(self.dims) - 1]))

    @torch.no_grad()
    def mlp(self, x):
        """DenseLayer x5 (wide_and_deep.py:113-133), inference: act(x W + b), ReLU on all but the last; returns the fp32 logit.
        16-bit nets: the MFMA kernels of csrc/mrec_dense.hip, the output layer in fp32; fp32 nets: the exact-fp32 MFMA kernels."""
        n = len(self.dims) - 1
        amp = self._amp
        if self._mfma:
            h = x if x.dtype == amp else x.to(amp)
            for i in range(n - 1):
                h = self.k.dense_fwd(h, self.dense16[2 * i], self.dense[2 * i + 1].detach(), relu=True)
            # the output layer (K5 -> 1) in fp32, on the exact-fp32 matrix kernel
            return self.k.dense32_fwd(h.float(), self.dense[2 * (n - 1)].detach(), self.dense[2 * (n - 1) + 1].detach(), relu=False)
        if self._f32net:
            h = x.contiguous()
            for i in range(n - 1):
                h = self.k.dense32_fwd(h, self.dense[2 * i].detach(), self.dense[2 * i + 1].detach(), relu=True)
            return self.k.dense32_fwd(h, self.dense[2 * (n - 1)].detach(), self.dense[2 * (n - 1) + 1].detach(), relu=False)
        return self._mlp_generic(x)

    # ---- hooks: nets without a HIP path.  The product refuses them; tests/_torch_net.py implements them for the oracle side ----
    def _unsupported(self, what):
        dt = {None: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}[getattr(self, "_amp", None)]
        return UnsupportedNet(f"MREC_EUNSUPPORTED: {what}: no hand-written HIP path for the {dt} dense net {self.dims} on {self.device} "
                              f"(16-bit nets: every width a multiple of 8; the layer in front of the output a power of two times 8, <= 512; "
                              f"head widths as above)")

    def _mlp_generic(self, x):
        raise self._unsupported("inference forward")

    def _head_generic(self, hs, wide, label, dhs):
        raise self._unsupported("output head")

    def _mlp_step_generic(self, emb, wide, label):
        raise self._unsupported("training step")

    # ---- the mixed-precision dense net, forward + backward by hand on the MFMA kernels -----------------
    def _db_slabs(self, i, B):
        """fp32 per-tile-row partial sums [ceil(B/256), out_i] of hidden layer i's bias gradient (written by the input-gradient
        kernel of layer i + 1, added up inside the dense Adam like the weight-gradient slabs)."""
        if self._dw_batch != B:
            self._dw, self._db, self._dw_batch = {}, {}, B
        t = self._db.get(i)
        if t is None:
            # written by the backward launch of layer i + 1, whose reduction width is dims[i + 2]
            rows = (self.k.dense32_colsum_tiles(B) if self._f32net else
                    self.k.dense_bwd_bias_slabs(B, self.dims[i + 1], self.dims[i + 2]))
            t = torch.empty((rows, self.dims[i + 1]), dtype=torch.float32, device=self.device)
            self._db[i] = t
        return t

    def _x3_now(self, B):
        """fp32 net: its MatMuls on three-part bf16 operands (csrc/mrec_gemm_x3.hip; fp32-class accuracy, measured against the exact
        kernels in tests/test_x3_gemm_gpu.py) at this batch size?  Shapes too small for it stay on the fp32-input matrix instruction."""
        n = len(self.dims) - 1
        return (getattr(self.cfg, "fp32_matmul", "x3") == "x3" and n >= 2
                and all(self.k.x3_supported(B, self.dims[i], self.dims[i + 1]) for i in range(n - 1)))

    def _dw_slabs(self, i, B):
        """fp32 batch slabs [S, in, out] the weight-gradient kernel of hidden layer i writes (allocated once per batch size)."""
        if self._dw_batch != B:
            self._dw, self._db, self._dw_batch = {}, {}, B
        t = self._dw.get(i)
        if t is None:
            K, N = self.dims[i], self.dims[i + 1]
            S = ((self.k.x3_wgrad_slabs if self._x3_now(B) else self.k.dense32_bwd_weight_slabs)(B, K, N) if self._f32net else
                 self.k.dense_bwd_weight_slabs(B, K, N))
            t = torch.empty((S, K, N), dtype=torch.float32, device=self.device)
            self._dw[i] = t
        return t

    @torch.no_grad()
    def _mlp_fwd(self, emb):
        """Hidden layers forward: the activations hs[0..n-1] (hs[0] = the MLP input), bias + ReLU in the GEMM epilogue."""
        n = len(self.dims) - 1
        hs = [emb if emb.dtype == self._amp else emb.to(self._amp)]
        B = hs[0].shape[0]
        d0 = self._drop(0, B)
        if d0 is not None and not self._emb_dropped:
            self.k.dropout_(hs[0], d0)     # the looked-up rows are consumed by the first layer only: in place
        self._emb_dropped = False
        for i in range(n - 3 if self._tail_now(B) else n - 1):         # (fused tail: its two layers run in _mlp_head's launch)
            # Dropout on the input of layer i + 1 (:117-118) = on this layer's output, in the GEMM epilogue
            hs.append(self.k.dense_fwd(hs[i], self.dense16[2 * i], self.dense[2 * i + 1].detach(), relu=True, drop_next=self._drop(i + 1, B),
                                       wt=self._dense16_t.get(i)))
        return hs

    @torch.no_grad()
    def _mlp_head(self, hs, wide, label):
        """Output layer + wide/deep add + sigmoid cross-entropy, forward AND backward.  Returns the context the
        backward needs: hs, loss, dlogit (= the wide branch's gradient), dh."""
        amp, n = self._amp, len(self.dims) - 1
        B = hs[0].shape[0]
        W5, b5 = self.dense[2 * (n - 1)], self.dense[2 * (n - 1) + 1]
        K5 = self.dims[n - 1]
        dl = self._drop(n - 1, B)
        dhs = dl.scale if dl is not None else 1.0
        if len(hs) == n - 2:
            # fused tail: layers n - 3 and n - 2 forward, the head, and the input gradients back to the output of layer n - 4
            prod = isinstance(wide, _WideProd)
            loss, dlogit, y2, dz4, dz3, dz2 = self.k.tail_fwd_bwd(
                hs[-1], self._tail_packed, self.dense[2 * (n - 3) + 1].detach(), self.dense[2 * (n - 2) + 1].detach(), W5.detach().view(-1),
                b5.detach(), wide.prod if prod else wide, self.wide_b if prod else None, label.view(-1), self._sens / B,
                self.dense_grad[2 * (n - 1)].view(-1), self.dense_grad[2 * (n - 2) + 1], self.dense_grad[2 * (n - 1) + 1],
                self.dense_grad[2 * (n - 3) + 1], self.dense_grad[2 * (n - 4) + 1], dwide_bias_out=self.wide_b_grad if prod else None,
                drop_in=self._drop(n - 3, B), out=self._tail_out.setdefault((B, hs[-1].dtype, self._slot), {}))
            return {"hs": hs + [y2], "loss": loss.view(()), "g_wide": dlogit.view(-1), "dh": dz2, "tail": (dz4, dz3)}
        if isinstance(wide, _WideProd):
            loss, _, dlogit, dh = self.k.head_fwd_bwd_wide(hs[-1], W5.detach().view(-1), b5.detach(), wide.prod, self.wide_b,
                                                            label.view(-1), self._sens / B, self.dense_grad[2 * (n - 1)].view(-1),
                                                            self.dense_grad[2 * (n - 2) + 1], self.dense_grad[2 * (n - 1) + 1],
                                                            dwide_bias_out=self.wide_b_grad, dh_scale=dhs)
            loss = loss.view(())
        elif self.k.head_supported(K5):
            # output layer + wide/deep add + sigmoid cross-entropy, forward AND backward, one pass over h4
            loss, _, dlogit, dh = self.k.head_fwd_bwd(hs[-1], W5.detach().view(-1), b5.detach(), wide, label.view(-1),
                                                       self._sens / B, self.dense_grad[2 * (n - 1)].view(-1),
                                                       self.dense_grad[2 * (n - 2) + 1], self.dense_grad[2 * (n - 1) + 1], dh_scale=dhs)
            loss = loss.view(())
        else:
            return self._head_generic(hs, wide, label, dhs)
        return {"hs": hs, "loss": loss, "g_wide": dlogit.view(-1), "dh": dh}

    @torch.no_grad()
    def _mlp_bwd(self, ctx):
        """Backward through the hidden layers; returns g_emb [B, F*D] (16-bit).  One launch per layer, from the top:
        the input gradient (MatMul bprop fused with the ReLU and BiasAdd bprops of the layer below: that layer's bias
        gradient is left as per-tile-row partial sums) and the weight gradient (fp32 batch slabs) are workgroups of the
        same kernel -- both read dh, and for the narrow layers neither fills the chip alone.  Slabs and partial sums are
        added up inside the dense Adam.  (Weight gradients on a parallel stream / graph branch instead were measured:
        the graph runtime queues chain kernels behind side-branch work, 0.87 -> 0.95 ms/step.  Round 5, layer 0 only: its input
        gradient alone on the chain and its weight gradient -- which feeds nothing but the dense Adam -- on a branch beside the
        SPARSE APPLY, an HBM-bound launch that leaves the matrix cores idle: the apply stretched by the whole GEMM, 177 -> 241 us
        on uniform ids and 96 -> 203 us on Zipf ids x 39 fields, the step 0.632 -> 0.655 ms; forked in front of the input gradient
        0.749 ms.  A GEMM workgroup holds its CU's registers and LDS for its whole K loop: the apply's waves do not fit beside it.)"""
        n = len(self.dims) - 1
        hs, dh = ctx["hs"], ctx["dh"]
        B = hs[0].shape[0]
        top, extra = n - 2, None
        if "tail" in ctx:
            # the tail launch has gone back through layers n - 2 and n - 3 already; their weight gradients (batch reductions) remain:
            # they ride the backward launch of layer n - 4
            dz4, dz3 = ctx["tail"]
            extra = [(hs[n - 3], dz3, self._dw_slabs(n - 3, B)), (hs[n - 2], dz4, self._dw_slabs(n - 2, B))]
            top = n - 4
        for i in range(top, -1, -1):
            dh = self.k.dense_bwd(dh, self.dense16[2 * i], hs[i], self._dw_slabs(i, B), mask=i > 0,
                                  db_slabs=self._db_slabs(i - 1, B) if i > 0 else None, drop_in=self._drop(i, B), extra=extra)
            extra = None
        return dh

    @torch.no_grad()
    def _mlp_step_f32(self, emb, wide, label):
        """Forward + backward of the fp32 net by hand (self._f32net): DenseLayer = MatMul + BiasAdd + ReLU in fp32
        (wide_and_deep.py:113-133 without use_mixed_precision; deepfm.py:98-150 with convert_dtype False) on ops.dense32_*, the
        output layer + wide/deep add + sigmoid cross-entropy and their bprops in one pass (ops.head_fwd_bwd), every gradient of
        the dense parameters summed into the flat gradient buffer.  Returns (loss, g_emb fp32 [B, F * D], g_wide [B])."""
        k, n = self.k, len(self.dims) - 1
        B = emb.shape[0]
        # Dropout on every DenseLayer's input while training (wide_and_deep.py:117-118; on in benchmarks/wide_deep/default_config.yaml:15,
        # whose net is fp32): x_i = h_(i-1) * mask_i / keep, applied in place to the stored activation -- its zeros then carry the ReLU's
        # AND the mask's pattern for the input-gradient kernel -- and the 1 / keep of the bprop is applied to dz_i before it goes back
        # through W_i (after the weight gradient, which wants dz_i itself)
        drops = [self._drop(i, B) for i in range(n)]
        if drops[0] is not None:
            k.dropout_(emb, drops[0])
        x3 = self._x3_now(B)
        hs = [emb]
        if x3:
            P = self.__dict__.setdefault("_x3_parts", {})
            if P.get("B") != B:
                P.clear()
                P.update(B=B, h=[k.x3_parts(B, self.dims[i], self.device) for i in range(n - 1)],
                         w=[k.x3_parts(self.dims[i], self.dims[i + 1], self.device) for i in range(n - 1)],
                         dz=[k.x3_parts(B, self.dims[i + 1], self.device) for i in range(n - 1)],
                         act=[torch.empty((B, self.dims[i + 1]), dtype=torch.float32, device=self.device) for i in range(n - 1)],
                         dx=[torch.empty((B, self.dims[i]), dtype=torch.float32, device=self.device) for i in range(n - 1)])
            k.x3_split(emb, out=P["h"][0])
            for i in range(n - 1):
                k.x3_split(self.dense[2 * i].detach(), out=P["w"][i])
                nxt = P["h"][i + 1] if i + 1 < n - 1 else None
                # bias + ReLU, the Dropout on the next layer's input and that input's parts: all in the GEMM's epilogue
                h = k.x3_fwd(P["h"][i], P["w"][i], B, self.dims[i], self.dims[i + 1], P["act"][i], bias=self.dense[2 * i + 1].detach(),
                             relu=True, parts_out=nxt, drop_next=drops[i + 1])
                hs.append(h)
        for i in range(0 if x3 else n - 1):
            h = k.dense32_fwd(hs[i], self.dense[2 * i].detach(), self.dense[2 * i + 1].detach(), relu=True)
            if drops[i + 1] is not None:
                k.dropout_(h, drops[i + 1])
            hs.append(h)
        K5 = self.dims[n - 1]
        if k.head_supported(K5):
            loss, _, dlogit, dh = k.head_fwd_bwd(hs[-1], self.dense[2 * (n - 1)].detach().view(-1), self.dense[2 * (n - 1) + 1].detach(),
                                                 wide, label.view(-1), self._sens / B, self.dense_grad[2 * (n - 1)].view(-1),
                                                 self.dense_grad[2 * (n - 2) + 1], self.dense_grad[2 * (n - 1) + 1],
                                                 dh_scale=drops[n - 1].scale if drops[n - 1] is not None else 1.0)
        else:
            # a last hidden layer wider than the head kernel's 512 columns (the reference's benchmark net ends 1024 -> 1): the
            # output end of Deep&Cross does the same arithmetic over [h | c] . w3 -- with c = [wide, 0] and w3 = [W5 | 1, 0] that is
            # h . W5 + wide + b5, its loss and every bprop, in one pass (csrc/mrec_dcn.hip)
            st = self.__dict__.setdefault("_wide_head", {})
            if st.get("B") != B:
                st.update(B=B, c=torch.zeros((B, 2), dtype=torch.float32, device=self.device),
                          w3=torch.zeros(K5 + 2, dtype=torch.float32, device=self.device),
                          dw3=torch.empty(K5 + 2, dtype=torch.float32, device=self.device), out={})
                st["w3"][K5] = 1.0
            st["c"][:, 0].copy_(wide.view(-1))
            st["w3"][:K5].copy_(self.dense[2 * (n - 1)].detach().view(-1))
            loss, _, dh, dc = k.dcn_head_fwd_bwd(hs[-1], st["c"], st["w3"], self.dense[2 * (n - 1) + 1].detach(), label.view(-1), self._sens / B,
                                                 st["dw3"], self.dense_grad[2 * (n - 2) + 1], self.dense_grad[2 * (n - 1) + 1], out=st["out"])
            self.dense_grad[2 * (n - 1)].view(-1).copy_(st["dw3"][:K5])
            dlogit = dc[:, 0].contiguous()
            if drops[n - 1] is not None:
                dh.mul_(drops[n - 1].scale)
                self.dense_grad[2 * (n - 2) + 1].mul_(drops[n - 1].scale)
        if x3:
            k.x3_split(dh, out=P["dz"][n - 2])
        for i in range(n - 2 if x3 else -1, -1, -1):
            Ki, Ni = self.dims[i], self.dims[i + 1]
            dw = self._dw_slabs(i, B)
            k.x3_gemm(2, P["h"][i], P["dz"][i], B, Ki, Ni, dw, S=dw.shape[0])
            if i > 0:      # ReLU (and Dropout mask) of layer i - 1, the 1 / keep of the Dropout in front of layer i, the bias gradient
                dh = k.x3_dgrad(P["dz"][i], P["w"][i], B, Ki, Ni, P["dx"][i], h=hs[i], colsum=self._db_slabs(i - 1, B),
                                parts_out=P["dz"][i - 1], scale=drops[i].scale if drops[i] is not None else 1.0)
            else:
                dh = k.x3_dgrad(P["dz"][0], P["w"][0], B, Ki, Ni, P["dx"][0])
                if drops[0] is not None:
                    k.dropout_(dh, drops[0])
        for i in range(-1 if x3 else n - 2, -1, -1):
            k.dense32_bwd_weight(hs[i], dh, self._dw_slabs(i, B))
            if drops[i] is not None and i > 0:
                dh.mul_(drops[i].scale)
            if i > 0:      # through the ReLU (and the Dropout mask) of layer i - 1; the column sums are that layer's bias gradient
                dh = k.dense32_bwd_input(dh, self.dense[2 * i].detach(), h=hs[i], colsum=self._db_slabs(i - 1, B))
            else:
                dh = k.dense32_bwd_input(dh, self.dense[0].detach())
                if drops[0] is not None:
                    k.dropout_(dh, drops[0])            # mask and 1 / keep of the looked-up rows (no ReLU in front of them to carry the mask)
        self._sum_dw_slabs()
        return loss.view(()), dh, dlogit.view(-1)

    def _mlp_step_eager(self, emb, wide, label, after_head=None, before_head=None):
        """Forward + backward of the mixed-precision MLP written out by hand (no autograd graph).  `emb` arrives
        in 16 bits straight from the gather kernel, and the gradient of the MLP input is returned in 16 bits
        for the sparse apply to widen on load.  Weights are read from their 16-bit shadows (no per-step cast kernels).
        Returns (loss, g_emb [B, F*D] 16-bit, g_wide [B] fp32).  after_head(g_wide) is called as soon as the wide
        branch's gradient exists (the caller may start the wide table's update beside the backward GEMMs)."""
        hs = self._mlp_fwd(emb)
        if before_head is not None:
            before_head()
        if callable(wide):
            wide = wide()                  # joins whatever stream computed the wide branch; returns the tensor
        ctx = self._mlp_head(hs, wide, label)
        if after_head is not None:
            after_head(ctx["g_wide"])
        g_emb = self._mlp_bwd(ctx)
        return ctx["loss"], g_emb, ctx["g_wide"]

    def _mlp_step(self, emb, wide, label, after_head=None):
        """The MLP step, replayed from HIP graphs once the engine has run two eager steps (workspaces exist by then).
        The graphs hold exactly the kernels of the eager path, in the same order, on the same buffers (weights /
        gradients are updated in place, so their addresses are stable); inputs are staged in three static tensors --
        the gather writes the embeddings there directly.  Graphs: hidden-layer forward [| head] | backward.  `wide` may
        be a function: it is called between the first two (the wide branch is computed on the side stream meanwhile);
        after_head runs before the backward graph (the wide branch's gradient exists from there on)."""
        if not (self._graph_level >= 1 and self._gpu and self.step_count > 2):
            return self._mlp_step_eager(emb, wide, label, after_head=after_head)
        g = self._mlp_graph
        if (g is None or g["emb"].shape != emb.shape or g["emb"].dtype != emb.dtype or (g["graph_head"] is None) == callable(wide)
                or isinstance(g["wide"], _WideProd) != isinstance(wide, _WideProd)):
            try:
                g = self._capture_mlp(emb, wide, label)
            except RuntimeError as e:          # capture refused: stay eager
                import warnings
                warnings.warn(f"HIP-graph capture of the MLP step failed, running it eagerly: {e}")
                self._graph_level = 0
                self._mlp_graph = None
                return self._mlp_step_eager(emb, wide, label, after_head=after_head)
        if emb.data_ptr() != g["emb"].data_ptr():
            g["emb"].copy_(emb)
        g["label"].copy_(label)

        def stage_wide():
            w_ = wide() if callable(wide) else wide
            if isinstance(w_, _WideProd):
                g["wide"].prod.copy_(w_.prod)
            else:
                g["wide"].copy_(w_)

        if g["graph_head"] is None:
            stage_wide()
            g["graph_fwd"].replay()                # hidden layers + head in one graph
        else:
            g["graph_fwd"].replay()
            stage_wide()
            g["graph_head"].replay()
        if after_head is not None:
            after_head(g["ctx"]["g_wide"])
        g["graph_bwd"].replay()
        return g["ctx"]["loss"], g["g_emb"], g["ctx"]["g_wide"]

    def _capture_mlp(self, emb, wide, label):
        late = callable(wide)
        if late:
            wide = wide()
        g = {"emb": torch.empty_like(emb), "label": torch.empty_like(label)}
        g["wide"] = _WideProd(wide.prod.clone()) if isinstance(wide, _WideProd) else wide.clone()
        g["emb"].copy_(emb)
        g["label"].copy_(label)
        torch.cuda.synchronize(self.device)
        # thread_local: RCCL's watchdog thread may query events while this thread captures
        g1 = torch.cuda.CUDAGraph()
        if late:
            # cut between the hidden layers and the head: the wide branch arrives in between
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                g["hs"] = self._mlp_fwd(g["emb"])
            gh = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gh, capture_error_mode="thread_local"):
                g["ctx"] = self._mlp_head(g["hs"], g["wide"], g["label"])
            g["graph_head"] = gh
        else:
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                g["hs"] = self._mlp_fwd(g["emb"])
                g["ctx"] = self._mlp_head(g["hs"], g["wide"], g["label"])
            g["graph_head"] = None
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            g["g_emb"] = self._mlp_bwd(g["ctx"])
        g["graph_fwd"], g["graph_bwd"] = g1, g2
        self._mlp_graph = g
        return g
