"""Argument checks with the messages of `mindspore._checkparam` that the reference's tests pin
(ci/st/online_learning/test_online_learning.py:72,93,114) [EXT: message text recalled from MindSpore 2.x]."""
import numbers


def check_bool(value, arg_name=None, prim_name=None):
    if not isinstance(value, bool):
        pre = f"For '{prim_name}', the '{arg_name}'" if prim_name and arg_name else "The input value"
        raise TypeError(f"{pre} must be a bool, but got {type(value).__name__}." if prim_name and arg_name
                        else f"The input value must be a bool, but got {type(value).__name__}.")
    return value


def _check_int(value, arg_name, prim_name, cond, cond_txt):
    pre = f"For '{prim_name}', the '{arg_name}'" if prim_name and arg_name else (f"The '{arg_name}'" if arg_name else "The input value")
    if isinstance(value, bool) or not isinstance(value, numbers.Integral):
        raise TypeError(f"{pre} must be int and must {cond_txt}, but got '{value}' with type '{type(value).__name__}'.")
    if not cond(value):
        raise ValueError(f"{pre} must be int and must {cond_txt}, but got '{value}' with type '{type(value).__name__}'.")
    return int(value)


def check_positive_int(value, arg_name=None, prim_name=None):
    return _check_int(value, arg_name, prim_name, lambda v: v > 0, "> 0")


def check_non_negative_int(value, arg_name=None, prim_name=None):
    return _check_int(value, arg_name, prim_name, lambda v: v >= 0, ">= 0")


def check_positive_float(value, arg_name=None, prim_name=None):
    pre = f"For '{prim_name}', the '{arg_name}'" if prim_name and arg_name else "The input value"
    if isinstance(value, bool) or not isinstance(value, numbers.Real):
        raise TypeError(f"{pre} must be float and must > 0, but got '{value}' with type '{type(value).__name__}'.")
    if not value > 0:
        raise ValueError(f"{pre} must be float and must > 0, but got '{value}' with type '{type(value).__name__}'.")
    return float(value)


def check_value_type(arg_name, value, valid_types, prim_name=None):
    valid_types = tuple(valid_types) if isinstance(valid_types, (list, tuple)) else (valid_types,)
    if (isinstance(value, bool) and bool not in valid_types) or not isinstance(value, valid_types):
        names = [t.__name__ for t in valid_types]
        raise TypeError(f"For '{prim_name}', the type of '{arg_name}' should be one of {names}, "
                        f"but got '{value}' with type '{type(value).__name__}'.")
    return value
