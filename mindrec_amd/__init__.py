"""mindrec_amd -- MI355X-native embedding hot path behind the mindspore_rec API.

Host Python + torch (device memory and streams only) over hand-written gfx950 HIP kernels in
csrc/, reached through the C-ABI of include/mrec.h.  See DESIGN.md.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
