#!/bin/bash
# Round 5, third GPU visit: the true-fp32 inference path (new kernels) and every test that now reaches it through
# precision="fp32" + eval mode; the C1 photograph test after its arg-max fix; a C3 bench line with the fp16 GEMMs timed.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_f32.py tests/test_hip_api.py tests/test_hip_zero_shot.py tests/test_hip_f16.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -E "fp32 arithmetic|C1 cosine|passed|failed|^E  |Error" | cut -c1-300 | tee gpurun_out/r05v3_tests.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r05v3_bench_c3.json | cut -c1-1800
