"""The kernel set every hot-path primitive of this package runs on.

Default (and the only one the product ships): `mindrec_amd.ops` + `mindrec_amd.experimental` -- libmrec_hip.so on an
MI355X.  There is NO CPU path here: a primitive handed host tensors raises.  `_install()` is a TEST hook: the
build container has no GPU, so `tests/golden/make_ref_fixtures.py` swaps in the oracle's restatements
(`tests/_ms_cpu_kernels.py`) to run the reference's own Python there and record fixtures; nothing under
`compat/` or `mindrec_amd/` ever calls it.
"""
_set = None


def K():
    """The active kernel set (lazily the HIP one)."""
    global _set
    if _set is None:
        from . import _hip_kernels
        _set = _hip_kernels
    return _set


def _install(kernel_set):
    global _set
    prev, _set = _set, kernel_set
    return prev
