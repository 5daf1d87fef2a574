"""TEST INFRASTRUCTURE: a Wide&Deep train step and a Deep&Cross train step written against the `mindspore` API surface of
compat/ (this repo's own code, not the reference's: the reference's sources cannot travel to the GPU box).  They state the same
model as models/wide_deep/src/wide_and_deep.py:136-492 and models/deep_and_cross/src/deep_and_cross.py:117-354 in a
different shape (layer lists, one helper per branch) so that the GPU tests can drive compat/mindspore + the HIP kernel set
with a realistic script and compare it with the fixtures the reference's own code produced."""
import mindspore.common.dtype as mstype
from mindspore import Parameter, ParameterTuple, nn, ops
from mindspore.common.initializer import initializer
from mindspore.nn.optim import FTRL, Adam, LazyAdam
from mindspore_rec import HashEmbeddingLookup


class Layer(nn.Cell):
    def __init__(self, n_in, n_out, relu=True, half=False):
        super().__init__()
        self.weight = Parameter(initializer("normal", [n_in, n_out], mstype.float32), name="weight")
        self.bias = Parameter(initializer("normal", [n_out], mstype.float32), name="bias")
        self.mm, self.add, self.act, self.cast = ops.MatMul(), ops.BiasAdd(), ops.ReLU() if relu else None, ops.Cast()
        self.half = half

    def construct(self, x):
        w, b = self.weight, self.bias
        if self.half:
            x, w, b = self.cast(x, mstype.float16), self.cast(w, mstype.float16), self.cast(b, mstype.float16)
        y = self.add(self.mm(x, w), b)
        if self.act is not None:
            y = self.act(y)
        return self.cast(y, mstype.float32) if self.half else y


class WideDeep(nn.Cell):
    def __init__(self, vocab, dim, fields, batch, hidden, sparse, dynamic, half=False, capacity=None):
        super().__init__()
        self.B, self.F, self.D = batch, fields, dim
        if dynamic:
            kw = {} if capacity is None else {"capacity": capacity}
            self.deep_table = HashEmbeddingLookup(embedding_size=dim, **kw)
            self.wide_table = HashEmbeddingLookup(embedding_size=1, **kw)
        else:
            self.deep_table = nn.EmbeddingLookup(vocab, dim, target="DEVICE", sparse=sparse)
            self.wide_table = nn.EmbeddingLookup(vocab, 1, target="DEVICE", sparse=sparse)
        self.wide_bias = Parameter(initializer("normal", [1], mstype.float32), name="wide_bias")
        dims = [fields * dim] + list(hidden) + [1]
        self.n_layers = len(dims) - 1
        for i in range(self.n_layers):
            setattr(self, f"layer{i}", Layer(dims[i], dims[i + 1], relu=i < self.n_layers - 1, half=half))
        self.table = self.deep_table.embedding_table
        self.reshape, self.sum, self.mul = ops.Reshape(), ops.ReduceSum(keep_dims=False), ops.Mul()

    def construct(self, ids, wts):
        mask = self.reshape(wts, (self.B, self.F, 1))
        wide = self.reshape(self.sum(self.mul(self.wide_table(ids), mask), 1) + self.wide_bias, (-1, 1))
        h = self.reshape(self.mul(self.deep_table(ids), mask), (-1, self.F * self.D))
        for i in range(self.n_layers):
            h = getattr(self, f"layer{i}")(h)
        return wide + h, self.table


class WideDeepLoss(nn.Cell):
    def __init__(self, net, l2_coef, with_l2):
        super().__init__(auto_prefix=False)
        self.net, self.l2_coef, self.with_l2 = net, l2_coef, with_l2
        self.ce, self.mean, self.sum, self.sq = ops.SigmoidCrossEntropyWithLogits(), ops.ReduceMean(), ops.ReduceSum(), ops.Square()

    def construct(self, ids, wts, label):
        logit, table = self.net(ids, wts)
        log_loss = self.mean(self.ce(logit, label))
        if not self.with_l2:
            return log_loss, log_loss
        return log_loss, log_loss + self.l2_coef * (self.sum(self.sq(table)) / 2)


class _Pick(nn.Cell):
    def __init__(self, net, i):
        super().__init__()
        self.net, self.i = net, i

    def construct(self, a, b, c):
        return self.net(a, b, c)[self.i]


class WideDeepTrainStep(nn.Cell):
    """FTRL on the wide table and the wide bias, (Lazy)Adam on the rest, both seeded with sens (wide_and_deep.py:387-492)."""

    def __init__(self, loss_net, lazy, sens=1024.0):
        super().__init__()
        self.loss_net = loss_net
        ws = loss_net.trainable_params()
        self.w_wide = ParameterTuple([p for p in ws if "wide" in p.name])
        self.w_deep = ParameterTuple([p for p in ws if "wide" not in p.name])
        self.opt_deep = (LazyAdam if lazy else Adam)(self.w_deep, learning_rate=3.5e-4, eps=1e-8, loss_scale=sens)
        self.opt_wide = FTRL(self.w_wide, learning_rate=5e-2, l1=1e-8, l2=1e-8, initial_accum=1.0, loss_scale=sens)
        self.grad = ops.GradOperation(get_by_list=True, sens_param=True)
        self.pick_w, self.pick_d = _Pick(loss_net, 0), _Pick(loss_net, 1)
        self.sens = sens

    def construct(self, ids, wts, label):
        lw, ld = self.loss_net(ids, wts, label)
        seed = ops.Fill()(ops.DType()(lw), ops.Shape()(lw), self.sens)
        gw = self.grad(self.pick_w, self.w_wide)(ids, wts, label, seed)
        gd = self.grad(self.pick_d, self.w_deep)(ids, wts, label, seed)
        self.opt_wide(gw)
        self.opt_deep(gd)
        return lw, ld


def wide_deep_from_fixture(z, cfg, comp, capacity=None):
    """The model above with the fixture's configuration and initial parameters; returns (train step, model)."""
    from mindspore import Tensor
    dyn = bool(cfg["dynamic_embedding"])
    net = WideDeep(cfg["vocab_size"], cfg["emb_dim"], cfg["field_size"], cfg["batch_size"], cfg["deep_layer_dim"], bool(cfg["sparse"]), dyn,
                   half=bool(cfg["use_mixed_precision"]), capacity=capacity)
    net.wide_bias.set_data(Tensor(z["init/wide_b"]))
    if not dyn:
        net.deep_table.embedding_table.set_data(Tensor(z["init/embedding_table"]))
        net.wide_table.embedding_table.set_data(Tensor(z["init/wide_embeddinglookup.embedding_table"]))
    for i in range(net.n_layers):
        lay = getattr(net, f"layer{i}")
        lay.weight.set_data(Tensor(z[f"init/dense_layer_{i + 1}.weight"]))
        lay.bias.set_data(Tensor(z[f"init/dense_layer_{i + 1}.bias"]))
    step = WideDeepTrainStep(WideDeepLoss(net, comp["l2_coef"], not comp["no_l2loss"]), lazy=comp["optimizer_d"] == "LazyAdam", sens=comp["sens"])
    step.set_train()
    return step, net
