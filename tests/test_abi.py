"""CPU: the C-ABI library builds, loads, and exports every symbol include/vitlens_hip.h declares
(no compute calls without a GPU); the Python binding table matches the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vitlens_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vl_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    names = declared_functions()
    assert "vl_gemm_bf16" in names and "vl_attn_fwd_bf16" in names and len(names) >= 15


def test_library_exports_every_declared_symbol():
    from vitlens_hip import _lib
    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.lib_path())
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_covers_header():
    from vitlens_hip import _lib
    names = set(declared_functions()) - {"vl_last_error"}
    assert names == set(_lib.SIGNATURES), (names ^ set(_lib.SIGNATURES))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vitlens_hip import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "_HERE", str(tmp_path))
    with pytest.raises(_lib.LibraryNotBuilt):
        _lib.load_library()


def test_ops_refuse_cpu_tensors():
    import torch
    from vitlens_hip import ops
    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.gemm(a, a)


def test_abi_version_is_one_number_in_header_library_and_binding():
    """`VL_ABI_VERSION` (header) == `vl_version()` (library) == `_lib.ABI_VERSION` (Python binding): a client built against an
    older header than the library it loads must notice before its first call (advisor finding, round 5: vl_ln_row_stats
    gained five parameters under an unchanged version 100)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "vitlens_hip.h")).read()
    want = int(re.search(r"#define\s+VL_ABI_VERSION\s+(\d+)", hdr).group(1))
    from vitlens_hip import _lib
    assert _lib.ABI_VERSION == want
    assert int(_lib.load_library().vl_version()) == want
