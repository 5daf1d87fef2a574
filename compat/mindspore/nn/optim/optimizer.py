"""Optimizer base of `mindspore.nn.optim` [EXT, SURVEY A.7]: gradients are multiplied by 1 / loss_scale first; a
RowTensor / MapTensorGrad is applied to the touched rows only, by the kernel set's fused dedup + segment-sum + update."""
import numpy as np
import torch

from ..._kernels import K
from ...common.parameter import Parameter, ParameterTuple
from ...common.sparse_tensor import MapTensorGrad, RowTensor
from ...common.tensor import Tensor
from ...experimental import MapParameter
from ..cell import Cell


def _raw(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t


class Optimizer(Cell):
    def __init__(self, learning_rate, parameters, weight_decay=0.0, loss_scale=1.0):
        super().__init__(auto_prefix=False)
        params = list(parameters)
        if not params:
            raise ValueError("For 'Optimizer', the argument parameters must not be empty.")
        if params and isinstance(params[0], dict):
            raise NotImplementedError("parameter groups are not used by the in-scope models (all call sites pass a ParameterTuple)")
        for p in params:
            if not isinstance(p, (Parameter, MapParameter)):
                raise TypeError(f"For 'Optimizer', the 'parameters' must be 'Parameter' or 'MapParameter', but got {type(p).__name__}.")
        if isinstance(loss_scale, int):
            loss_scale = float(loss_scale)
        if not isinstance(loss_scale, float) or loss_scale <= 0:
            raise ValueError(f"For 'Optimizer', the 'loss_scale' must be a float > 0, but got {loss_scale!r}.")
        if weight_decay < 0:
            raise ValueError(f"For 'Optimizer', the 'weight_decay' must be >= 0, but got {weight_decay}.")
        self.__dict__["parameters"] = ParameterTuple(params)
        self.__dict__["_parameters"] = self.parameters
        self.learning_rate = learning_rate
        self.weight_decay = float(weight_decay)
        self.loss_scale = float(loss_scale)
        self.reciprocal_scale = 1.0 / float(loss_scale)
        self.global_step = 0
        self._target = "Ascend"
        self.exec_weight_decay = self.weight_decay > 0

    @property
    def target(self):
        return self._target

    @target.setter
    def target(self, value):
        if value not in ("CPU", "Ascend", "GPU"):
            raise ValueError(f"For 'Optimizer', the argument 'target' must be one of ['CPU', 'Ascend', 'GPU'], but got {value}.")
        self.__dict__["_target"] = value

    @property
    def unique(self):
        return True

    def get_lr(self):
        lr = self.learning_rate
        if callable(lr) and not isinstance(lr, torch.Tensor):
            lr = lr(self.global_step)
        return float(lr)

    def _slot(self, p, prefix, init):
        """Optimizer state of a dense Parameter: a Parameter named `<prefix>.<param name>`."""
        key = (prefix, id(p))
        st = self.__dict__.setdefault("_state", {})
        if key not in st:
            st[key] = Parameter(torch.full_like(_raw(p).detach(), float(init)), name=f"{prefix}.{p.name}", requires_grad=False)
        return st[key]

    def get_parameters(self, expand=True):
        yield from self.__dict__.get("_state", {}).values()

    def _dense_grad(self, p, g):
        g = _raw(g).detach()
        if g.shape != p.shape:
            raise ValueError(f"For '{self.cls_name}', the gradient of {p.name} has shape {tuple(g.shape)}, expected {tuple(p.shape)}.")
        if self.exec_weight_decay:
            g = g + self.weight_decay * self.loss_scale * _raw(p).detach()
        return g.to(torch.float32).contiguous()

    def _check(self, gradients):
        if len(gradients) != len(self.parameters):
            raise ValueError(f"For '{self.cls_name}', the number of gradients ({len(gradients)}) must equal the number of "
                             f"parameters ({len(self.parameters)}).")


class _AdamBase(Optimizer):
    lazy = False

    def __init__(self, params, learning_rate=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, use_locking=False, use_nesterov=False,
                 weight_decay=0.0, loss_scale=1.0, use_amsgrad=False, **kw):
        super().__init__(learning_rate, params, weight_decay, loss_scale)
        if not 0.0 < beta1 < 1.0 or not 0.0 < beta2 < 1.0:
            raise ValueError(f"For '{self.cls_name}', beta1 and beta2 must be in (0, 1), but got {beta1}, {beta2}.")
        if eps <= 0:
            raise ValueError(f"For '{self.cls_name}', the 'eps' must be > 0, but got {eps}.")
        if use_amsgrad:
            raise NotImplementedError("use_amsgrad is not used by the in-scope models")
        self.beta1, self.beta2, self.eps = np.float32(beta1), np.float32(beta2), float(eps)
        self.use_nesterov, self.use_locking = bool(use_nesterov), bool(use_locking)
        self.beta1_power, self.beta2_power = np.float32(1.0), np.float32(1.0)
        for p in self.parameters:                    # state exists from construction on (a checkpoint can be loaded before step 1)
            if isinstance(p, Parameter):
                self._slot(p, "moment1", 0.0)
                self._slot(p, "moment2", 0.0)
        # the step scalars as (non-trainable) Parameters too, so that a checkpoint carries them: resuming continues bit for bit
        self.__dict__["_scalars"] = {n: Parameter(Tensor(np.array([v], np.float64), device="cpu"), name=n, requires_grad=False)
                                     for n, v in (("beta1_power", 1.0), ("beta2_power", 1.0), ("global_step", 0.0))}

    def get_parameters(self, expand=True):
        yield from super().get_parameters(expand)
        yield from self._scalars.values()

    def _sync_from_parameters(self):
        """After load_param_into_net: the host mirrors follow the loaded Parameters."""
        s = self._scalars
        self.beta1_power = np.float32(float(s["beta1_power"][0]))
        self.beta2_power = np.float32(float(s["beta2_power"][0]))
        self.global_step = int(round(float(s["global_step"][0])))

    def construct(self, gradients):
        self._check(gradients)
        self.global_step += 1
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        for n, v in (("beta1_power", self.beta1_power), ("beta2_power", self.beta2_power), ("global_step", self.global_step)):
            self._scalars[n].as_subclass(torch.Tensor)[0] = float(v)
        kw = dict(lr=self.get_lr(), beta1=float(self.beta1), beta2=float(self.beta2), eps=self.eps,
                  beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                  grad_scale=self.reciprocal_scale, use_nesterov=self.use_nesterov)
        k = K()
        with torch.no_grad():
            for p, g in zip(self.parameters, gradients):
                if isinstance(p, MapParameter):
                    if not isinstance(g, MapTensorGrad):
                        raise TypeError(f"For '{self.cls_name}', the gradient of MapParameter {p.name} must come from MapTensorGet.")
                    if g.keys.numel():
                        p._store.apply_lazy_adam(g.keys, g.values, **kw)
                    continue
                w = _raw(p).detach()
                m, v = _raw(self._slot(p, "moment1", 0.0)), _raw(self._slot(p, "moment2", 0.0))
                if isinstance(g, RowTensor):
                    if self.lazy:
                        k.sparse_lazy_adam_(w, m, v, k.sparse_plan(g.indices), g.values, None, **kw)
                    else:
                        # nn.Adam on a RowTensor off the host: every row's moments decay, the touched rows take the summed
                        # gradient -- Adam on the densified gradient [EXT]
                        k.dense_adam_(w, m, v, k.gather_bwd_dense(w.shape[0], g.indices, g.values), **kw)
                    continue
                k.dense_adam_(w, m, v, self._dense_grad(p, g), **kw)
        return True


class Adam(_AdamBase):
    """nn.Adam (wide_and_deep.py:435-437; deep_and_cross.py:342-344)."""


class LazyAdam(_AdamBase):
    """nn.LazyAdam: Adam on dense gradients; on a RowTensor only the touched rows' moments and weights move
    (wide_and_deep.py:420-422)."""
    lazy = True


class FTRL(Optimizer):
    """nn.FTRL(params, initial_accum=0.1, learning_rate=0.001, lr_power=-0.5, l1=0.0, l2=0.0, use_locking=False,
    loss_scale=1.0, weight_decay=0.0) (wide_and_deep.py:423-430,438-445)."""

    def __init__(self, params, initial_accum=0.1, learning_rate=0.001, lr_power=-0.5, l1=0.0, l2=0.0, use_locking=False,
                 loss_scale=1.0, weight_decay=0.0):
        super().__init__(learning_rate, params, weight_decay, loss_scale)
        if initial_accum < 0 or l1 < 0 or l2 < 0:
            raise ValueError(f"For 'FTRL', 'initial_accum', 'l1' and 'l2' must be >= 0, but got {initial_accum}, {l1}, {l2}.")
        if lr_power > 0:
            raise ValueError(f"For 'FTRL', the 'lr_power' must be <= 0, but got {lr_power}.")
        if not isinstance(learning_rate, (int, float)) or learning_rate <= 0:
            raise ValueError(f"For 'FTRL', the 'learning_rate' must be a float > 0 (dynamic learning rates are not supported), "
                             f"but got {learning_rate!r}.")
        self.initial_accum, self.lr_power, self.l1, self.l2 = float(initial_accum), float(lr_power), float(l1), float(l2)
        self.use_locking = bool(use_locking)
        for p in self.parameters:
            if isinstance(p, Parameter):
                self._slot(p, "accum", self.initial_accum)
                self._slot(p, "linear", 0.0)
        self.__dict__["_scalars"] = {"global_step": Parameter(Tensor(np.array([0.0], np.float64), device="cpu"), name="global_step",
                                                                requires_grad=False)}

    def get_parameters(self, expand=True):
        yield from super().get_parameters(expand)
        yield from self._scalars.values()

    def _sync_from_parameters(self):
        self.global_step = int(round(float(self._scalars["global_step"][0])))

    def construct(self, gradients):
        self._check(gradients)
        self.global_step += 1
        self._scalars["global_step"].as_subclass(torch.Tensor)[0] = float(self.global_step)
        kw = dict(lr=self.get_lr(), l1=self.l1, l2=self.l2, lr_power=self.lr_power, grad_scale=self.reciprocal_scale)
        k = K()
        with torch.no_grad():
            for p, g in zip(self.parameters, gradients):
                if isinstance(p, MapParameter):
                    if not isinstance(g, MapTensorGrad):
                        raise TypeError(f"For 'FTRL', the gradient of MapParameter {p.name} must come from MapTensorGet.")
                    if g.keys.numel():
                        p._store.apply_ftrl(g.keys, g.values, initial_accum=self.initial_accum, **kw)
                    continue
                w = _raw(p).detach()
                acc, lin = _raw(self._slot(p, "accum", self.initial_accum)), _raw(self._slot(p, "linear", 0.0))
                if isinstance(g, RowTensor):
                    k.sparse_ftrl_(w, acc, lin, k.sparse_plan(g.indices), g.values, None, **kw)
                    continue
                k.dense_ftrl_(w, acc, lin, self._dense_grad(p, g), **kw)
        return True
