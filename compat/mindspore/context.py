"""`mindspore.context`: the knobs the in-scope scripts set and read.  Most of MindSpore's are
graph-compiler switches with nothing to switch here; they are stored and returned, not acted on."""
import os

import torch

GRAPH_MODE = 0
PYNATIVE_MODE = 1


class ParallelMode:
    STAND_ALONE = "stand_alone"
    DATA_PARALLEL = "data_parallel"
    HYBRID_PARALLEL = "hybrid_parallel"
    SEMI_AUTO_PARALLEL = "semi_auto_parallel"
    AUTO_PARALLEL = "auto_parallel"
    MODE_LIST = [STAND_ALONE, DATA_PARALLEL, HYBRID_PARALLEL, SEMI_AUTO_PARALLEL, AUTO_PARALLEL]


_ctx = {"mode": GRAPH_MODE, "device_target": "GPU", "device_id": int(os.environ.get("LOCAL_RANK", os.environ.get("DEVICE_ID", 0)) or 0)}
_auto = {}
_ps = {}


def _reset_auto():
    _auto.clear()
    _auto.update({"parallel_mode": ParallelMode.STAND_ALONE, "gradients_mean": False, "device_num": 1, "global_rank": 0,
                  "full_batch": False, "parameter_broadcast": False, "all_reduce_fusion_config": [], "enable_parallel_optimizer": False,
                  "search_mode": "dynamic_programming", "strategy_ckpt_save_file": "", "strategy_ckpt_load_file": ""})


def _reset_ps():
    _ps.clear()
    _ps.update({"enable_ps": False, "ms_role": os.environ.get("MS_ROLE", "MS_WORKER"), "enable_ssl": False})


_reset_auto()
_reset_ps()


def set_context(**kw):
    for k, v in kw.items():
        if k == "device_target" and v not in ("GPU", "CPU", "Ascend"):
            raise ValueError(f"For 'set_context', 'device_target' must be one of ['GPU', 'CPU', 'Ascend'], but got {v}.")
        if k == "mode" and v not in (GRAPH_MODE, PYNATIVE_MODE):
            raise ValueError(f"For 'set_context', 'mode' must be GRAPH_MODE (0) or PYNATIVE_MODE (1), but got {v}.")
        _ctx[k] = v


def get_context(attr_key):
    return _ctx.get(attr_key)


def set_auto_parallel_context(**kw):
    for k, v in kw.items():
        if k == "parallel_mode" and v not in ParallelMode.MODE_LIST:
            raise ValueError(f"For 'set_auto_parallel_context', unknown 'parallel_mode' {v!r}.")
        _auto[k] = v


def get_auto_parallel_context(attr_key):
    if attr_key not in _auto:
        raise ValueError(f"Get context keyword {attr_key} is not recognized!")
    return _auto[attr_key]


def reset_auto_parallel_context():
    _reset_auto()


def set_ps_context(**kw):
    _ps.update(kw)


def get_ps_context(attr_key):
    return _ps.get(attr_key)


def reset_ps_context():
    _reset_ps()


_host_tensors = False      # TEST hook (tests/golden/make_ref_fixtures.py): keep Tensors in host memory whatever device_target says,
                           # so that the reference's CI cases, which ask for "GPU", can run in the GPU-less build container


def _torch_device():
    """Where new Tensors live: the MI355X named by device_id, or host memory under device_target='CPU' (only usable with a
    test-installed kernel set: the hot-path primitives have no CPU implementation)."""
    if _ctx["device_target"] == "CPU" or _host_tensors:
        return torch.device("cpu")
    return torch.device("cuda", int(_ctx.get("device_id") or 0))
