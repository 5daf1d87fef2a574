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

    def __init__(self, loss_net, lazy, sens=1024.0, reduce=False):
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
        # data parallel (wide_and_deep.py:458-470,487-489): the mean of every rank's gradients before either optimizer sees them
        self.reducer_flag = bool(reduce)
        if reduce:
            from mindspore.communication.management import get_group_size
            from mindspore.nn.wrap.grad_reducer import DistributedGradReducer
            self.reduce_w = DistributedGradReducer(self.w_wide, True, get_group_size())
            self.reduce_d = DistributedGradReducer(self.w_deep, True, get_group_size())

    def construct(self, ids, wts, label):
        lw, ld = self.loss_net(ids, wts, label)
        seed = ops.Fill()(ops.DType()(lw), ops.Shape()(lw), self.sens)
        gw = self.grad(self.pick_w, self.w_wide)(ids, wts, label, seed)
        gd = self.grad(self.pick_d, self.w_deep)(ids, wts, label, seed)
        if self.reducer_flag:
            gw, gd = self.reduce_w(gw), self.reduce_d(gd)
        self.opt_wide(gw)
        self.opt_deep(gd)
        return lw, ld


def wide_deep_from_fixture(z, cfg, comp, capacity=None, reduce=False):
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
    step = WideDeepTrainStep(WideDeepLoss(net, comp["l2_coef"], not comp["no_l2loss"]), lazy=comp["optimizer_d"] == "LazyAdam", sens=comp["sens"],
                             reduce=reduce)
    step.set_train()
    return step, net


# ---- Deep&Cross (models/deep_and_cross/src/deep_and_cross.py:117-354 restated against the same API) ---------------------------------
class Cross(nn.Cell):
    """y = x0 * (x_l . w) + b + x_l  (DCN-v1 cross layer, :139-149)."""

    def __init__(self, width):
        super().__init__()
        self.cross_weight = Parameter(initializer("normal", [width, 1], mstype.float32), name="cross_weight")
        self.cross_bias = Parameter(initializer("normal", [width, 1], mstype.float32), name="cross_bias")
        self.mm, self.reshape = ops.MatMul(), ops.Reshape()

    def construct(self, x_l, x_0):
        s = self.mm(x_l, self.cross_weight)                                   # [B, 1]
        return x_0 * s + self.reshape(self.cross_bias, (1, -1)) + x_l


class DeepCross(nn.Cell):
    def __init__(self, vocab, dim, fields, batch, hidden, n_cross):
        super().__init__()
        self.batch_size, self.field_size, self.D = batch, fields, dim
        self.lookup = nn.EmbeddingLookup(vocab, dim, target="DEVICE", sparse=False)
        X = fields * dim
        self.n_cross = n_cross
        for i in range(n_cross):
            setattr(self, f"cross{i}", Cross(X))
        self.hidden0, self.hidden1 = Layer(X, hidden[0]), Layer(hidden[0], hidden[1])
        self.out = Layer(hidden[1] + X, 1, relu=False)
        self.reshape, self.mul, self.concat = ops.Reshape(), ops.Mul(), ops.Concat(axis=1)

    def construct(self, ids, wts):
        x = self.reshape(self.mul(self.lookup(ids), self.reshape(wts, (self.batch_size, self.field_size, 1))), (-1, self.field_size * self.D))
        d = self.hidden1(self.hidden0(x))
        c = x
        for i in range(self.n_cross):
            c = getattr(self, f"cross{i}")(c, x)
        return self.out(self.concat((d, c)))


class LogLoss(nn.Cell):
    def __init__(self, net):
        super().__init__(auto_prefix=False)
        self.net, self.ce, self.mean = net, ops.SigmoidCrossEntropyWithLogits(), ops.ReduceMean()

    def construct(self, ids, wts, label):
        return self.mean(self.ce(self.net(ids, wts), label))


class AdamTrainStep(nn.Cell):
    def __init__(self, loss_net, lr=1e-4, eps=1e-8, loss_scale=1000.0):
        super().__init__(auto_prefix=False)
        self.loss_net = loss_net
        self.weights = ParameterTuple(loss_net.trainable_params())
        self.optimizer = Adam(self.weights, learning_rate=lr, eps=eps, loss_scale=loss_scale)
        self.grad, self.sens = ops.GradOperation(get_by_list=True, sens_param=True), loss_scale

    def construct(self, ids, wts, label):
        loss = self.loss_net(ids, wts, label)
        seed = ops.Fill()(ops.DType()(loss), ops.Shape()(loss), self.sens)
        self.optimizer(self.grad(self.loss_net, self.weights)(ids, wts, label, seed))
        return loss


def deep_cross_from_fixture(z, cfg, comp):
    from mindspore import Tensor
    net = DeepCross(cfg["vocab_size"], cfg["emb_dim"], cfg["field_size"], cfg["batch_size"], cfg["deep_layer_dim"], cfg["cross_layer_num"])
    net.lookup.embedding_table.set_data(Tensor(z["init/deep_embeddinglookup.embedding_table"]))
    for i in range(net.n_cross):
        c = getattr(net, f"cross{i}")
        c.cross_weight.set_data(Tensor(z[f"init/cross_layer_{i + 1}.cross_weight"]))
        c.cross_bias.set_data(Tensor(z[f"init/cross_layer_{i + 1}.cross_bias"]))
    for lay, k in ((net.hidden0, 1), (net.hidden1, 2), (net.out, 3)):
        lay.weight.set_data(Tensor(z[f"init/dense_layer_{k}.weight"]))
        lay.bias.set_data(Tensor(z[f"init/dense_layer_{k}.bias"]))
    step = AdamTrainStep(LogLoss(net), lr=comp["lr"], eps=comp["eps"], loss_scale=comp["loss_scale"])
    step.set_train()
    return step, net


# ---- DeepFM (models/deepfm/src/deepfm.py:152-295 restated against the same API) -----------------------------------------------------
class DeepFM(nn.Cell):
    """logit = sum_f w[id_f] * x_f  +  0.5 * sum_d ((sum_f v_f)^2 - sum_f v_f^2)_d  +  MLP(concat_f v_f),  v_f = V[id_f] * x_f."""

    def __init__(self, vocab, dim, fields, batch, hidden, half=False):
        super().__init__()
        self.batch_size, self.field_size, self.D = batch, fields, dim
        self.linear = Parameter(initializer("normal", [vocab, 1], mstype.float32), name="linear")
        self.factors = Parameter(initializer("normal", [vocab, dim], mstype.float32), name="factors")
        dims = [fields * dim] + list(hidden) + [1]
        self.n_layers = len(dims) - 1
        for i in range(self.n_layers):
            setattr(self, f"layer{i}", Layer(dims[i], dims[i + 1], relu=i < self.n_layers - 1, half=half))
        self.gather, self.reshape, self.mul, self.sum, self.sq = ops.Gather(), ops.Reshape(), ops.Mul(), ops.ReduceSum(keep_dims=False), ops.Square()

    def construct(self, ids, wts):
        x = self.reshape(wts, (self.batch_size, self.field_size, 1))
        first = self.sum(self.mul(self.gather(self.linear, ids, 0), x), 1)                          # [B, 1]
        v = self.mul(self.gather(self.factors, ids, 0), x)                                          # [B, F, D]
        second = self.reshape(0.5 * self.sum(self.sq(self.sum(v, 1)) - self.sum(self.sq(v), 1), 1), (-1, 1))
        h = self.reshape(v, (-1, self.field_size * self.D))
        for i in range(self.n_layers):
            h = getattr(self, f"layer{i}")(h)
        return first + second + h


class DeepFMLoss(nn.Cell):
    """mean log loss + l2_coef / 2 * (sum V^2 + sum w^2) over the WHOLE tables (deepfm.py:252-259)."""

    def __init__(self, net, l2_coef):
        super().__init__(auto_prefix=False)
        self.net, self.l2_coef = net, l2_coef
        self.ce, self.mean, self.sum, self.sq = ops.SigmoidCrossEntropyWithLogits(), ops.ReduceMean(), ops.ReduceSum(keep_dims=False), ops.Square()

    def construct(self, ids, wts, label):
        loss = self.mean(self.ce(self.net(ids, wts), label))
        return loss + self.l2_coef * (self.sum(self.sq(self.net.factors)) + self.sum(self.sq(self.net.linear))) * 0.5


def deepfm_from_fixture(z, cfg, comp):
    from mindspore import Tensor
    net = DeepFM(cfg["data_vocab_size"], cfg["data_emb_dim"], cfg["data_field_size"], cfg["batch_size"], cfg["deep_layer_args"][0],
                 half=bool(comp["convert_dtype"]))
    net.linear.set_data(Tensor(z["init/fm_w"]))
    net.factors.set_data(Tensor(z["init/embedding_table"]))
    for i in range(net.n_layers):
        lay = getattr(net, f"layer{i}")
        lay.weight.set_data(Tensor(z[f"init/dense_layer_{i + 1}.weight"]))
        lay.bias.set_data(Tensor(z[f"init/dense_layer_{i + 1}.bias"]))
    step = AdamTrainStep(DeepFMLoss(net, comp["l2_coef"]), lr=comp["lr"], eps=comp["eps"], loss_scale=comp["loss_scale"])
    step.set_train()
    return step, net
