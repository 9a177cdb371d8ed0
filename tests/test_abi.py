"""C-ABI surface checks that need no GPU: the library loads, exports everything include/*.h declares,
sizes its workspaces sanely and reports errors through return codes (never by crashing)."""
import ctypes
import os
import re

import pytest

from adaptigraph_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "adaptigraph_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/adaptigraph_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_version_and_capacity_queries():
    L = _lib.lib()
    assert L.ag_version() >= 1
    assert L.ag_edge_capacity(256, 1001, 10, 0, 1) == 256 * 1001 * 10
    assert L.ag_edge_capacity(2, 4097, 5, 1, 1) == 2 * 4097 * 6
    assert L.ag_edge_capacity(2, 7, 10, 0, 1) == 2 * 7 * 7          # topk clipped to N (graph.py:128)
    e_cap = 256 * 1001 * 10
    fwd = L.ag_forward_workspace_bytes(256, 1001, e_cap)
    assert fwd >= e_cap * 160 * 4 + 5 * 256 * 1001 * 160 * 4      # Eterm + five node tables
    prm = _lib.RolloutParams(256, 1001, 1000, 1, 10, 0, 1, 10, 0, 0.0)
    assert L.ag_rollout_workspace_bytes(ctypes.byref(prm)) > fwd
    assert L.ag_edges_workspace_bytes(256, 1001, 10, 0, 1) >= 256 * 1001 * 11 * 4


def test_errors_are_codes_not_crashes():
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.ag_model_create(None, None, ctypes.byref(h)) != 0
    assert b"null" in L.ag_last_error()
    bad = _lib.ModelConfig(128, 4, 2, 1, 3, 3, 100.0)              # nf the kernels are not built for
    dummy = (ctypes.c_void_p * 22)(*([1] * 22))
    assert L.ag_model_create(ctypes.byref(bad), dummy, ctypes.byref(h)) == -4
    assert b"nf=128" in L.ag_last_error()
    assert L.ag_build_edges(None, None, None, None, 10, 0, 1, 1, 8, 1, None, None, None, 80, None, 0, None) != 0
    assert L.ag_model_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
