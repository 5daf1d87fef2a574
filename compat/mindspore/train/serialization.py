"""`mindspore.train.serialization`: save_checkpoint / load_checkpoint / load_param_into_net
(models/wide_deep/eval.py:86-107; examples/online_learning/online_train.py:27).  MindSpore's `.ckpt` is a protobuf
stream; THIS ONE IS A NUMPY `.npz` STREAM under the same file name (a checkpoint written by MindSpore itself cannot be read, nor
the other way round): one array per Parameter under MindSpore's parameter names, optimizer slots as `moment1.<name>` /
`moment2.<name>` / `accum.<name>` / `linear.<name>`, the optimizers' step scalars under their cell path
(`optimizer_d.global_step`); a MapParameter is three arrays (`<name>::keys`, `<name>::values`, `<name>::status`) plus one per
optimizer slot (`<name>::slot::<slot>`)."""
import os

import numpy as np
import torch

from ..common.parameter import Parameter
from ..common.tensor import Tensor
from ..experimental import MapParameter


_SLOT_PREFIXES = ("moment1.", "moment2.", "accum.", "linear.")


def _items(save_obj, legacy=False):
    """(checkpoint name, parameter).  Optimizer slots go under MindSpore's own names -- `moment1.<parameter name>`, `accum.<...>`:
    a script that filters a checkpoint by `filter_prefix='moment1'` keeps meaning what it meant -- ; the step scalars every optimizer
    holds under the same name (`global_step`, `beta1_power`, ...) under the optimizer cell's path, which is what tells two
    optimizers' apart (TrainStepWrap has two, wide_and_deep.py:415-445).  legacy=True: the names of files written before round 5
    (slots under the cell path too), which load_param_into_net still accepts."""
    from ..nn.cell import Cell
    if isinstance(save_obj, Cell):
        seen = set()
        for cname, c in save_obj.cells_and_names():
            for p in c._params.values():
                if id(p) not in seen:
                    seen.add(id(p))
                    yield p.name, p
            for p in c.get_parameters(expand=False):          # state a cell keeps outside its attributes (optimizers)
                if id(p) not in seen:
                    seen.add(id(p))
                    under_path = (cname + "." if cname else "") + p.name
                    yield (under_path if legacy or not str(p.name).startswith(_SLOT_PREFIXES) else p.name), p
        return
    if isinstance(save_obj, dict):
        yield from save_obj.items()
        return
    if isinstance(save_obj, (list, tuple)):
        for e in save_obj:
            if isinstance(e, dict):
                yield e["name"], e["data"]
            else:
                yield e.name, e
        return
    raise TypeError(f"For 'save_checkpoint', the argument 'save_obj' should be nn.Cell, list or dict, but got {type(save_obj)}.")


def save_checkpoint(save_obj, ckpt_file_name, integrated_save=True, async_save=False, append_dict=None, enc_key=None,
                    enc_mode="AES-GCM", choice_func=None, **kw):
    if enc_key is not None:
        raise NotImplementedError("checkpoint encryption is not provided")
    if not isinstance(ckpt_file_name, str):
        raise TypeError(f"For 'save_checkpoint', the argument 'ckpt_file_name' must be string, but got {type(ckpt_file_name)}.")
    if not ckpt_file_name.endswith(".ckpt"):
        ckpt_file_name += ".ckpt"
    from ..nn.cell import Cell
    if isinstance(save_obj, Cell):
        for _, c in save_obj.cells_and_names():
            low = c.__dict__.get("_lowered")
            if low and getattr(low, "sharded", False) and low.dirty:
                raise RuntimeError("save_checkpoint: this train cell runs as a ROW-SHARDED engine and its full-size tables are stale "
                                   "mirrors of the ranks' shards; mindspore.Model.train refreshes them on every rank wherever a "
                                   "ModelCheckpoint is due -- outside of it call Model.sync_parameters() on every rank first")
    out = {}
    for name, p in _items(save_obj):
        if choice_func is not None and not choice_func(name):
            continue
        if isinstance(p, MapParameter):
            k, v, st = p.export_data(False)
            out[name + "::keys"], out[name + "::values"], out[name + "::status"] = k.asnumpy(), v.asnumpy(), st.asnumpy()
            for slot, arr in p._store.export_slots().items():
                out[f"{name}::slot::{slot}"] = arr
        elif isinstance(p, torch.Tensor):
            out[name] = p.detach().cpu().numpy()
        else:
            out[name] = np.asarray(p)
    for k, v in (append_dict or {}).items():
        out["__append__::" + k] = np.asarray(v)
    d = os.path.dirname(os.path.abspath(ckpt_file_name))
    os.makedirs(d, exist_ok=True)
    tmp = ckpt_file_name + ".tmp"
    with open(tmp, "wb") as f:
        np.savez(f, **out)
    os.replace(tmp, ckpt_file_name)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return ckpt_file_name


def load_checkpoint(ckpt_file_name, net=None, strict_load=False, filter_prefix=None, dec_key=None, dec_mode="AES-GCM",
                    specify_prefix=None, choice_func=None):
    """-> {name: Parameter}; a MapParameter comes back as the entries `<name>::keys|values|status|slot::*` (host arrays
    wrapped as Parameters) which `load_param_into_net` feeds to `import_data`."""
    if not isinstance(ckpt_file_name, str) or not ckpt_file_name.endswith(".ckpt"):
        raise ValueError(f"For 'load_checkpoint', the checkpoint file should end with '.ckpt', but got {ckpt_file_name!r}.")
    if not os.path.exists(ckpt_file_name):
        raise ValueError(f"For 'load_checkpoint', the checkpoint file: {ckpt_file_name} does not exist, please check "
                         "whether the 'ckpt_file_name' is correct.")
    pref = (filter_prefix,) if isinstance(filter_prefix, str) else tuple(filter_prefix or ())
    params = {}
    with np.load(ckpt_file_name, allow_pickle=False) as z:
        for name in z.files:
            if name.startswith("__append__::"):
                params[name[len("__append__::"):]] = z[name]
                continue
            if pref and name.startswith(pref):
                continue
            if choice_func is not None and not choice_func(name):
                continue
            params[name] = Parameter(Tensor(z[name], device="cpu"), name=name, requires_grad=False)
    if not params:
        raise ValueError(f"The loaded parameter dict is empty after filter or specify, please check whether "
                         f"'filter_prefix' or 'specify_prefix' are set correctly.")
    if net is not None:
        load_param_into_net(net, params, strict_load)
    return params


def load_param_into_net(net, parameter_dict, strict_load=False):
    """Copies by name; returns the list of the net's parameters that found no entry."""
    from ..nn.cell import Cell
    if not isinstance(net, Cell):
        raise TypeError(f"For 'load_param_into_net', the argument 'net' should be a Cell, but got {type(net)}.")
    if not isinstance(parameter_dict, dict):
        raise TypeError(f"For 'load_param_into_net', the argument 'parameter_dict' should be a dict, but got {type(parameter_dict)}.")
    missing = []
    old_names = {id(q): n for n, q in _items(net, legacy=True)}
    for name, p in _items(net):
        if name not in parameter_dict and old_names.get(id(p)) in parameter_dict:
            name = old_names[id(p)]                            # a file written before round 5
        if isinstance(p, MapParameter):
            if name + "::keys" not in parameter_dict:
                missing.append(name)
                continue
            g = lambda s: parameter_dict[name + s].as_subclass(torch.Tensor)      # noqa: E731
            p._store.clear()
            p.import_data((g("::keys").to(p.device), g("::values").to(p.device), None))
            slots = {k[len(name) + len("::slot::"):]: v.as_subclass(torch.Tensor) for k, v in parameter_dict.items()
                     if isinstance(k, str) and k.startswith(name + "::slot::")}
            if slots:
                p._store.import_slots(g("::keys").to(p.device), {k: v.to(p.device) for k, v in slots.items()})
            continue
        src = parameter_dict.get(name)
        if src is None:
            missing.append(name)
            continue
        src = src.as_subclass(torch.Tensor) if isinstance(src, torch.Tensor) else torch.as_tensor(src)
        if tuple(src.shape) != tuple(p.shape):
            raise RuntimeError(f"For 'load_param_into_net', {name} in the argument 'net' should have the same shape as {name} in "
                               f"the argument 'parameter_dict'. But got its shape {tuple(p.shape)} in the argument 'net' and shape "
                               f"{tuple(src.shape)} in the argument 'parameter_dict'.")
        p.set_data(src.to(p.device, p.dtype))
    for _, c in net.cells_and_names():
        if hasattr(c, "_sync_from_parameters"):
            c._sync_from_parameters()           # optimizers: step scalars held on the host follow the loaded Parameters
    if strict_load and missing:
        raise RuntimeError(f"For 'load_param_into_net', {missing} in the argument 'net' are not loaded.")
    return missing


def build_searched_strategy(strategy_filename):
    """Auto-parallel's saved slicing strategy (models/wide_deep/eval.py:88): this package's row shards are merged by
    `merge_sliced_parameter` from their shapes alone, so the strategy is an empty record."""
    if not os.path.exists(strategy_filename):
        raise ValueError(f"For 'build_searched_strategy', the strategy file {strategy_filename} does not exist.")
    return {}


def merge_sliced_parameter(sliced_parameters, strategy=None):
    """Row slices of one parameter (one per rank, in rank order) -> the whole parameter (eval.py:96-104)."""
    if not sliced_parameters:
        raise ValueError("For 'merge_sliced_parameter', the argument 'sliced_parameters' should not be empty.")
    name = sliced_parameters[0].name
    data = torch.cat([p.as_subclass(torch.Tensor) for p in sliced_parameters], dim=0)
    return Parameter(Tensor(data), name=name, requires_grad=sliced_parameters[0].trainable)
