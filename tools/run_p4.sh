set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_gemm_park.py -x -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_p4.log
tail -25 gpurun_out/pytest_p4.log | cut -c1-250
timeout 600 python tools/kernel_bench.py ${KB_ARGS} > gpurun_out/kernel_bench.log 2>&1; grep -E "^gemm|qkv|attn|layernorm|Error|error" gpurun_out/kernel_bench.log | cut -c1-200
