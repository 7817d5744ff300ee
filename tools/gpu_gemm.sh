#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -n 4 -k "gemm" -p no:cacheprovider 2>&1 | tail -15 | cut -c1-300 > gpurun_out/pytest_gemm.log
cat gpurun_out/pytest_gemm.log
timeout 600 python tools/kernel_bench.py --quick > gpurun_out/kernel_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/kernel_bench.log
