"""`mindspore.parallel._utils` (mindspore_rec/train/rec_model.py:26)."""
from .. import context


def _device_number_check(parallel_mode, device_number):
    if parallel_mode == context.ParallelMode.STAND_ALONE and device_number != 1:
        raise ValueError(f"If parallel_mode is {parallel_mode}, device_number must be 1, but got device_number: {device_number}")


def _get_parallel_mode():
    return context.get_auto_parallel_context("parallel_mode")


def _get_device_num():
    return context.get_auto_parallel_context("device_num")


def _get_gradients_mean():
    return context.get_auto_parallel_context("gradients_mean")
