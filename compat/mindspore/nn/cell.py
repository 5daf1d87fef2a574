"""`mindspore.nn.Cell`: parameter / sub-cell bookkeeping with MindSpore's naming rule (a child cell assigned to an
attribute of a parent whose `auto_prefix` is on gets `attr.` in front of its parameters' names -- TrainStepWrap splits its
parameters by `"wide" in params.name`, models/wide_deep/src/wide_and_deep.py:407-411), `construct` as the call."""
from collections import OrderedDict

import torch

from ..common.parameter import Parameter
from ..experimental import MapParameter

_PARAM_TYPES = (Parameter, MapParameter)
_call_depth = [0]


class Cell:
    def __init__(self, auto_prefix=True, flags=None):
        d = self.__dict__
        d["_params"], d["_cells"] = OrderedDict(), OrderedDict()
        d["_auto_prefix"], d["training"], d["requires_grad"] = bool(auto_prefix), False, False
        d["_flags"] = dict(flags or {})
        d["phase"] = "train"
        d["parameter_layout_dict"] = {}
        d["_lowered"] = None

    # ---- attribute bookkeeping ---------------------------------------------------------------------------------
    def __setattr__(self, name, value):
        d = self.__dict__
        if "_params" not in d:
            raise AttributeError("For 'Cell', can not assign attributes before Cell.__init__() is called.")
        if isinstance(value, _PARAM_TYPES):
            d["_cells"].pop(name, None)
            d.pop(name, None)
            if value.name in ("Parameter", None, ""):
                value.name = name
            d["_params"][name] = value
        elif isinstance(value, Cell):
            d["_params"].pop(name, None)
            d.pop(name, None)
            d["_cells"][name] = value
            if d["_auto_prefix"]:
                value.update_parameters_name(name + ".")
        else:
            if name in d["_params"]:
                del d["_params"][name]
            if name in d["_cells"]:
                del d["_cells"][name]
            d[name] = value

    def __getattr__(self, name):
        d = self.__dict__
        if "_params" in d:
            if name in d["_params"]:
                return d["_params"][name]
            if name in d["_cells"]:
                return d["_cells"][name]
        raise AttributeError(f"The '{type(self).__name__}' object has no attribute '{name}'.")

    def __delattr__(self, name):
        d = self.__dict__
        for store in (d["_params"], d["_cells"], d):
            if name in store:
                del store[name]
                return
        raise AttributeError(name)

    @property
    def cls_name(self):
        return type(self).__name__

    # ---- calling -----------------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        # A TOP-LEVEL call of a training cell is where MindSpore's GRAPH_MODE compiles; here it is where a recognised train step is
        # handed to the fused engine (mindspore/_lower.py).  Calls made from inside another cell's construct just run.
        if _call_depth[0] == 0 and self.__dict__["training"] and not kwargs and self.__dict__.get("_lowered") is not False:
            from .._lower import lowered
            low = lowered(self, args)
            if low is not None:
                return low(*args)
        fwd = self.__dict__.get("_lowered_forward")
        if fwd is not None and not self.__dict__["training"] and not kwargs and len(args) == 2:
            # the model cell of a ROW-SHARDED lowered train step, evaluated: its tables are shards on their owner ranks, so the
            # forward is the engine's collective one (every rank evaluates, as the reference's scripts do)
            return fwd(*args)
        _call_depth[0] += 1
        try:
            return self.construct(*args, **kwargs)
        finally:
            _call_depth[0] -= 1

    def construct(self, *args, **kwargs):
        raise NotImplementedError(f"For 'Cell', the method 'construct' of {self.cls_name} is not defined.")

    def compile(self, *args, **kwargs):
        return None

    # ---- traversal ---------------------------------------------------------------------------------------------
    def cells(self):
        return list(self._cells.values())

    def name_cells(self):
        return OrderedDict(self._cells)

    def cells_and_names(self, cells=None, name_prefix=""):
        seen = cells if cells is not None else set()
        if id(self) in seen:
            return
        seen.add(id(self))
        yield name_prefix, self
        for n, c in self._cells.items():
            yield from c.cells_and_names(seen, (name_prefix + "." if name_prefix else "") + n)

    def insert_child_to_cell(self, child_name, child_cell):
        if not isinstance(child_cell, Cell):
            raise TypeError(f"For 'insert_child_to_cell', the child must be a Cell, but got {type(child_cell).__name__}.")
        self.__setattr__(child_name, child_cell)

    def insert_param_to_cell(self, param_name, param, check_name_contain_dot=True):
        self.__setattr__(param_name, param)

    def parameters_and_names(self, name_prefix="", expand=True):
        """(structural name, parameter): the attribute path from this cell, NOT `parameter.name` -- the rule MindSpore names
        parameters by when a cell is assigned to a parent (`update_parameters_name`)."""
        seen = set()
        cells = self.cells_and_names(name_prefix=name_prefix) if expand else [(name_prefix, self)]
        for cname, c in cells:
            for attr, p in c._params.items():
                if id(p) not in seen:
                    seen.add(id(p))
                    yield (cname + "." + attr if cname else attr), p

    def get_parameters(self, expand=True):
        for _, p in self.parameters_and_names(expand=expand):
            yield p

    def parameters_dict(self, recurse=True):
        return OrderedDict((p.name, p) for p in self.get_parameters(expand=recurse))

    def trainable_params(self, recurse=True):
        return [p for p in self.get_parameters(expand=recurse) if p.trainable]

    def untrainable_params(self, recurse=True):
        return [p for p in self.get_parameters(expand=recurse) if not p.trainable]

    def update_parameters_name(self, prefix="", recurse=True):
        """parameter.name = prefix + the parameter's STRUCTURAL name inside this cell [EXT: MindSpore's rule -- it is why a
        BatchNorm's `Parameter(name="mean")` held in the attribute `moving_mean` is saved as `<cell>.moving_mean`].  Consequence
        for Wide&Deep: `self.wide_b = Parameter(name="Wide_b")` (models/wide_deep/src/wide_and_deep.py:161-163) is named
        `network.network.wide_b` by the time TrainStepWrap tests `"wide" in params.name` (:407-411) -- it belongs to FTRL."""
        for name, p in self.parameters_and_names(expand=recurse):
            p.name = prefix + name

    def init_parameters_data(self, auto_parallel_mode=False):
        return self.parameters_dict()

    # ---- modes and flags ---------------------------------------------------------------------------------------
    def set_train(self, mode=True):
        for _, c in self.cells_and_names():
            c.__dict__["training"] = bool(mode)
            c.__dict__["phase"] = "train" if mode else "predict"
        return self

    def set_grad(self, requires_grad=True):
        self.__dict__["requires_grad"] = bool(requires_grad)
        return self

    def add_flags(self, **flags):
        self._flags.update(flags)
        return self

    def add_flags_recursive(self, **flags):
        for _, c in self.cells_and_names():
            c._flags.update(flags)
        return self

    def get_flags(self):
        return self._flags

    def to_float(self, dst_type):
        self._flags["to_float"] = dst_type
        return self

    def set_broadcast_flag(self, mode=True):
        self._flags["broadcast_flag"] = bool(mode)
        return self

    def set_comm_fusion(self, fusion_type, recurse=True):
        return self

    def set_auto_parallel(self):
        return self

    def recompute(self, **kw):
        return self

    def shard(self, *a, **k):
        return self

    def __repr__(self):
        inner = ", ".join(self._cells)
        return f"{self.cls_name}<{inner}>"


class GraphCell(Cell):
    """A network loaded from a compiled graph: RecModel.online_train refuses it in sink mode
    (mindspore_rec/train/rec_model.py:153-156)."""

    def __init__(self, graph=None, params_init=None):
        super().__init__(auto_prefix=True)
        self.graph = graph
