"""`mindspore.ops.primitive`: the Primitive base and `constexpr` (mindspore_rec/ops/embedding.py:32,41-44)."""


class Primitive:
    """A named operator object; `__call__` runs it now on device tensors (there is no graph compiler here: a whole
    recognised train step is lowered to one HIP graph instead, mindrec_amd/lowering.py)."""

    def __init__(self, name=None):
        self.name = name or type(self).__name__
        self.attrs = {}

    def add_prim_attr(self, name, value):
        self.attrs[name] = value
        return self

    def del_prim_attr(self, name):
        self.attrs.pop(name, None)
        return self

    def set_prim_instance_name(self, instance_name):
        self.instance_name = instance_name
        return self

    def shard(self, in_strategy=None, out_strategy=None):
        """Auto-parallel sharding strategies (wide_and_deep.py:212,510-511) are layout hints for MindSpore's graph
        compiler; the row-sharded engine (mindrec_amd/wide_deep_shard.py) decides its own layout.  Recorded only."""
        self.attrs["in_strategy"], self.attrs["out_strategy"] = in_strategy, out_strategy
        return self

    def set_device(self, device_target):
        self.attrs["primitive_target"] = device_target
        return self

    set_stage = recompute = place = lambda self, *a, **k: self

    def __call__(self, *args):
        raise NotImplementedError(f"primitive {self.name} has no implementation in this package")

    def __repr__(self):
        return f"Prim[{self.name}]"


PrimitiveWithInfer = PrimitiveWithCheck = Primitive


def prim_attr_register(fn):
    return fn


def constexpr(fn=None, get_instance=True, name=None, reuse_result=True, check=True):
    """Compile-time constant folding in MindSpore; here the function simply runs."""
    if fn is None:
        return lambda f: f
    return fn


_primexpr = constexpr
