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


def test_dynamic_symbol_table_is_exactly_the_header():
    """-fvisibility=hidden + csrc/exports.map: `nm -D` shows the declared ag_* functions and nothing else (no mangled launchers,
    no template instantiations, no __hip_cuid_* words)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.check_output([nm, "-D", "--defined-only", _lib.build()], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_symbols()


def test_every_engine_option_is_documented_in_the_header_and_readable():
    """The option names ag_set_option accepts (csrc/ag_api.hip) are exactly the ones ag_get_option answers, and each is described in the header's option list
    (a caller of the C ABI has nothing else to go by)."""
    src = open(os.path.join(ROOT, "adaptigraph_amd", "csrc", "ag_api.hip")).read()
    body = lambda fn: src[src.index(f"int {fn}("):src.index("return AG_OK;", src.index(f"int {fn}("))]
    names = lambda fn: set(re.findall(r'!strcmp\(name, "([a-z0-9_]+)"\)', body(fn)))
    setters, getters = names("ag_set_option"), names("ag_get_option")
    assert setters == getters and len(setters) >= 12
    header = open(os.path.join(ROOT, "include", "adaptigraph_hip.h")).read()
    for n in sorted(setters):
        assert f'"{n}"' in header, f"option {n} is not described in include/adaptigraph_hip.h"


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
    assert L.ag_rollout_workspace_bytes(ctypes.byref(prm)) <= 2.95e9               # bounded compact tables (r05; 3.55 GB until r04); 2.11 GB for a default-mode model
    assert L.ag_rollout_workspace_bytes_for(None, ctypes.byref(prm)) == L.ag_rollout_workspace_bytes(ctypes.byref(prm))   # no model: any mode
    assert L.ag_edges_workspace_bytes(256, 1001, 10, 0, 1) >= 256 * 1001 * 11 * 4


def test_errors_are_codes_not_crashes():
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.ag_model_create(None, None, ctypes.byref(h)) != 0
    assert b"null" in L.ag_last_error()
    v = ctypes.c_int(7)
    assert L.ag_get_option(None, b"precision", ctypes.byref(v)) != 0 and b"null" in L.ag_last_error() and v.value == 7
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


def test_training_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks of the row-n4 entry points run before any HIP call: null tables, widths beyond the compiled 150, chain
    kinds / input widths that do not exist, ragged element counts — all come back as AG_ERR_ARG (-1) with a message."""
    L = _lib.lib()
    assert L.ag_train_pack(None, None, 150, 150, 150, 0, 0, 0, 5, 0, None, None) == -1
    buf = (ctypes.c_float * 8)()
    assert L.ag_train_pack(buf, None, 150, 151, 151, 0, 0, 0, 5, 0, buf, None) == -1                 # n_in > 150
    assert L.ag_train_pack(buf, buf, 150, 40, 40, 0, 0, 1, 1, 0, buf, None) == -1                    # compact image holds <= 32 columns
    arr = (ctypes.c_void_p * 4)()
    assert L.ag_train_chain(7, 0, 0, buf, buf, arr, None, arr, None, 10, 17, None) == -1               # unknown kind
    assert L.ag_train_chain(0, 0, 0, buf, buf, arr, None, arr, None, 10, 16, None) == -1               # edge chain takes 17 inputs
    assert L.ag_train_chain(0, 0, 0, buf, buf, arr, None, arr, None, 10, 17, None) == -1               # null layer table
    assert b"table 0" in L.ag_last_error()
    assert L.ag_add3_relu(buf, buf, buf, buf, 6, None) == -1                                           # n % 4 != 0
    assert L.ag_relu_mask(None, buf, buf, 8, None) == -1
    i32 = (ctypes.c_int32 * 4)(160, 0, 0, 0)
    assert L.ag_train_weight_grads(0, arr, i32, arr, i32, i32, 10, buf, buf, 1 << 20, None) == -1      # n_layers < 1
    assert L.ag_train_weight_grads_workspace_bytes(48000, 4) > L.ag_train_weight_grads_workspace_bytes(1000, 1) > 0
    assert L.ag_edge_inputs_forward(buf, 15, 2, 14, None, None, None, 0, None) == -1                   # attr + group > D
    assert L.ag_model_status(None, None, None) == -1
