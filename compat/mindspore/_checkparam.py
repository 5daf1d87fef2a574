"""`mindspore._checkparam` (imported as `validator` / `Validator` by mindspore_rec/ops/embedding.py:22 and
mindspore_rec/train/rec_model.py:22).  Message texts follow MindSpore 2.x [EXT]; the substrings the reference's CI pins
(ci/st/online_learning/test_online_learning.py:72,93,114) are produced by check_positive_int / check_bool."""
from mindrec_amd._validator import (check_bool, check_non_negative_int, check_positive_float,  # noqa: F401
                                    check_positive_int, check_value_type)
import numbers


def check_is_int(value, arg_name=None, prim_name=None):
    if isinstance(value, bool) or not isinstance(value, numbers.Integral):
        raise TypeError(f"The '{arg_name}' must be int, but got '{value}' with type '{type(value).__name__}'.")
    return int(value)


def check_is_float(value, arg_name=None, prim_name=None):
    if isinstance(value, bool) or not isinstance(value, numbers.Real):
        raise TypeError(f"The '{arg_name}' must be float, but got '{value}' with type '{type(value).__name__}'.")
    return float(value)


def check_non_negative_float(value, arg_name=None, prim_name=None):
    v = check_is_float(value, arg_name, prim_name)
    if v < 0:
        raise ValueError(f"The '{arg_name}' must be float and must >= 0, but got '{value}'.")
    return v


def check_string(value, valid_values, arg_name=None, prim_name=None):
    if isinstance(value, str) and value in valid_values:
        return value
    raise ValueError(f"For '{prim_name}', the '{arg_name}' must be str and must be in '{valid_values}', but got '{value}'.")


def check_float_range(value, lo, hi, inc, arg_name=None, prim_name=None):
    v = check_is_float(value, arg_name, prim_name)
    ok = {"neither": lo < v < hi, "left": lo <= v < hi, "right": lo < v <= hi, "both": lo <= v <= hi}[inc]
    if not ok:
        raise ValueError(f"For '{prim_name}', the '{arg_name}' must be in range of ({lo}, {hi}) [{inc}], but got {value}.")
    return v


INC_NEITHER, INC_LEFT, INC_RIGHT, INC_BOTH = "neither", "left", "right", "both"
