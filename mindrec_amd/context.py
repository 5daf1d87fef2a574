"""The slice of `mindspore.context` that the hot path consults:
`get_context("device_target")` (rec_model.py:179) and the parameter-server flags read by
HashEmbeddingLookup.__init__ (mindspore_rec/ops/embedding.py:103-116)."""
GRAPH_MODE = 0
PYNATIVE_MODE = 1

_ctx = {"mode": GRAPH_MODE, "device_target": "GPU", "device_id": 0}
_ps = {"enable_ps": False, "ms_role": "MS_WORKER"}


def set_context(**kw):
    for k, v in kw.items():
        if k == "device_target" and v not in ("GPU", "CPU", "Ascend"):
            raise ValueError(f"For 'set_context', 'device_target' must be one of ['GPU', 'CPU', 'Ascend'], but got {v}.")
        _ctx[k] = v


def get_context(key):
    return _ctx.get(key)


def set_ps_context(**kw):
    _ps.update(kw)


def get_ps_context(key):
    return _ps.get(key)


def reset_ps_context():
    _ps.update({"enable_ps": False, "ms_role": "MS_WORKER"})
