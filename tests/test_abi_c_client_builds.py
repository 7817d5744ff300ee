"""CPU: the C client of the ABI (tests/native/abi_c_client.c) compiles as plain C against include/vitlens_hip.h and links
against the built library - the header is valid C, every entry point it calls resolves (it runs on the GPU box:
tests/test_hip_abi_c_client.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_compiles_and_links(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not installed")
    from vitlens_hip import _lib
    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__ as g
        g.build()
    libdir = os.path.dirname(_lib.lib_path())
    r = subprocess.run([gcc, "-O2", "-Wall", "-Werror=implicit-function-declaration", os.path.join(ROOT, "tests", "native", "abi_c_client.c"),
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-L", libdir,
                        "-L", os.path.join(rocm, "lib"), "-lvitlens_hip", "-lamdhip64", "-lm", "-o", str(tmp_path / "abi_c_client")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
