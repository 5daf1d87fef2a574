"""`mindspore.communication.management`: init / get_rank / get_group_size over torch.distributed -- backend "nccl" IS RCCL
over xGMI on this platform (models/wide_deep/train_and_eval_distribute.py:135; models/deep_and_cross/train.py:62).
One process per GPU; rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
import datetime
import os

import torch
import torch.distributed as dist

from .. import context

GlobalComm = type("GlobalComm", (), {"WORLD_COMM_GROUP": "world", "INITED": False, "BACKEND": None})


def init(backend_name=None):
    if dist.is_initialized():
        GlobalComm.INITED = True
        return
    target = context.get_context("device_target")
    if backend_name is None:
        backend_name = "nccl" if target == "GPU" else "gloo"
    if backend_name in ("nccl", "rccl"):
        backend = "nccl"
    elif backend_name in ("gloo", "mccl"):
        backend = "gloo"
    else:
        raise RuntimeError(f"For 'init', the argument 'backend_name' must be one of 'nccl' (RCCL) or 'gloo', but got {backend_name!r}.")
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        raise RuntimeError("For 'init', launch one process per GPU with torch.distributed.run (RANK / WORLD_SIZE / MASTER_ADDR / "
                           "MASTER_PORT in the environment); MindSpore's MS_SCHED / MS_WORKER roles do not exist here.")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        context.set_context(device_id=int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("MREC_INIT_TIMEOUT_S", 300))))
    GlobalComm.INITED, GlobalComm.BACKEND = True, backend
    context.set_auto_parallel_context(device_num=dist.get_world_size(), global_rank=dist.get_rank())


def release():
    if dist.is_initialized():
        dist.destroy_process_group()
    GlobalComm.INITED = False


def _need():
    if not dist.is_initialized():
        raise RuntimeError("Distributed Communication has not been inited; call mindspore.communication.management.init() first.")


def get_rank(group=None):
    _need()
    return dist.get_rank()


def get_group_size(group=None):
    _need()
    return dist.get_world_size()


def get_local_rank(group=None):
    _need()
    return int(os.environ.get("LOCAL_RANK", 0))
