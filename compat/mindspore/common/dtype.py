"""`mindspore.common.dtype`: the dtype objects ARE torch's (tensors are torch device tensors)."""
import torch

float16 = half = torch.float16
float32 = single = torch.float32
float64 = double = torch.float64
bfloat16 = torch.bfloat16
int8, int16, int32, int64 = torch.int8, torch.int16, torch.int32, torch.int64
uint8 = torch.uint8
bool_ = torch.bool

_BY_NAME = {"float16": float16, "float32": float32, "float64": float64, "bfloat16": bfloat16, "int8": int8, "int16": int16,
            "int32": int32, "int64": int64, "uint8": uint8, "bool": bool_}


def dtype_to_nptype(t):
    import numpy as np
    return {float16: np.float16, float32: np.float32, float64: np.float64, int8: np.int8, int16: np.int16, int32: np.int32,
            int64: np.int64, uint8: np.uint8, bool_: np.bool_}[t]


def pytype_to_dtype(t):
    return {float: float32, int: int64, bool: bool_}[t]
