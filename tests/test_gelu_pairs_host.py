"""CPU: the packed-pair GELU helpers of csrc/vl_common.h (v_pk_fma_f32 forms used by the GEMM epilogues) compute, lane by lane
and bit for bit, what the scalar helpers compute - checked on a HOST build of the same header over every finite bf16 value in
both lanes (tests/native/gelu_pairs_host.cpp).  Written after a packed form shipped a wrong gelu' for the high lane
(`__builtin_bit_cast` of a vector element read element 0): the GPU parity tests catch that too, this one catches it before a
GPU is involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (ext_vector_type, __builtin_elementwise_*)")
def test_packed_gelu_forms_equal_the_scalar_forms_on_every_bf16_value(tmp_path):
    src = open(os.path.join(ROOT, "vit-lens_amd", "csrc", "vl_common.h")).read().replace("#include <hip/hip_runtime.h>", "")
    hdr = tmp_path / "vl_common_host.h"
    hdr.write_text(src)
    exe = str(tmp_path / "gelu_pairs_host")
    build = subprocess.run([CLANG, "-O2", "-std=c++17", "-ffp-contract=off", f'-DVL_COMMON_HOST_H="{hdr}"',
                            os.path.join(ROOT, "tests", "native", "gelu_pairs_host.cpp"), "-o", exe, "-lm"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "bad=0" in run.stdout, run.stdout[-500:]
    assert int(run.stdout.split("checked=")[1].split()[0]) > 65000
