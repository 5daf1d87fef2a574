"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mrec.h
declares; argument validation and workspace queries work without touching the device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mrec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrec_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    from mindrec_amd import _lib
    l = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/mrec.h but not exported"
    assert sorted(names) == _lib.EXPORTED, "python binding table and header disagree"


def test_no_extra_exports():
    """Only mrec_* symbols are visible (kernels and helpers are hidden)."""
    import subprocess
    from mindrec_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if " T " in ln]
    extra = [s for s in syms if not s.startswith("mrec_") and s not in ("_init", "_fini")]
    assert not extra, extra


def test_version_and_strerror():
    from mindrec_amd import _lib
    l = _lib.lib()
    assert l.mrec_version() >= 100
    assert l.mrec_strerror(0) == b"ok"
    assert b"workspace" in l.mrec_strerror(-2)
    assert l.mrec_strerror(-99) == b"unknown error"


def test_workspace_queries_and_validation():
    from mindrec_amd import _lib
    n = 16384 * 39
    assert _lib.query_bytes("mrec_dedup_workspace_bytes", n) >= 2 * 4 * 2 * n
    assert _lib.query_bytes("mrec_group_workspace_bytes", n) >= 2 * 4 * n
    assert _lib.query_bytes("mrec_sparse_apply_workspace_bytes", n, 80) >= 2 * (n // 8) * 80 * 4      # two carry rows per 8-entry window
    assert _lib.query_bytes("mrec_map_bytes", 1000) >= 1000 * (8 + 1 + 4)
    assert _lib.query_bytes("mrec_shard_route_workspace_bytes", n, 8) > 0
    assert _lib.query_bytes("mrec_cross_layers_bwd_workspace_bytes", 6, 16384, 1170) > 0
    l = _lib.lib()
    out = C.c_size_t()
    assert l.mrec_dedup_workspace_bytes(-1, C.byref(out)) == -1            # MREC_EINVAL
    assert l.mrec_dedup_workspace_bytes(1 << 31, C.byref(out)) == -3       # MREC_EUNSUPPORTED
    assert l.mrec_sparse_apply_workspace_bytes(10, 0, C.byref(out)) == -1
    # argument errors are reported before any HIP call
    assert l.mrec_gather_rows_f32_i32(None, 10, 8, 8, None, -1, None, None, None) == -1
    assert l.mrec_gather_rows_f32_i32(None, 10, 4, 8, None, 5, None, None, None) == -1     # ld < D
    assert l.mrec_cross_layers_bwd_f32(None, None, None, 9, 4, 4, None, None, None, None, None, 0, None) in (-1, -3)
    with pytest.raises(_lib.MrecError) as e:
        _lib.call("mrec_fill_normal_f32", None, 4, 0, 4, 0, 0, 1, 0.0, None)
    assert e.value.code == -1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from mindrec_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensors_refused():
    import torch
    from mindrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ops.unique(torch.zeros(2, dtype=torch.int32))
