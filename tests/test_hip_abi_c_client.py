"""GPU: the C ABI driven from plain C (tests/native/abi_c_client.c) - no Python, no torch in the process that computes: device
memory from the HIP runtime, `vl_gemm_bf16`, `vl_layernorm_fwd`, `vl_gemm_f32` against host loops, and the status-code error
protocol.  This is the binding INTEGRATION.md describes, exercised end to end."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_of_the_abi(tmp_path):
    gcc = shutil.which("gcc") or "gcc"          # plain C: the header is C, the HIP runtime has a C API
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    libdir = os.path.join(ROOT, "vit-lens_amd", "vitlens_hip")
    exe = str(tmp_path / "abi_c_client")
    build = subprocess.run([gcc, "-O2", os.path.join(ROOT, "tests", "native", "abi_c_client.c"), "-I", os.path.join(ROOT, "include"),
                            "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-L", libdir, "-L", os.path.join(rocm, "lib"),
                            "-lvitlens_hip", "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(run.stdout)
    assert run.returncode == 0 and "ALL OK" in run.stdout, (run.stdout[-2000:], run.stderr[-2000:])
