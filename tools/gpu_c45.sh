#!/bin/bash
# GPU-box visit: API tests + C4 / C5 training-step bench lines with per-GEMM-shape detail
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_api.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 | cut -c1-300 > gpurun_out/pytest_api.log
tail -12 gpurun_out/pytest_api.log
timeout 900 python bench.py --workload c4 --steps 3 --warmup 1 ${C4_ARGS} --detail gpurun_out/bench_c4_detail.json > gpurun_out/bench_c4.log 2>&1
tail -3 gpurun_out/bench_c4.log | cut -c1-2500
timeout 900 python bench.py --workload c5 --steps 3 --warmup 1 ${C5_ARGS} --detail gpurun_out/bench_c5_detail.json > gpurun_out/bench_c5.log 2>&1
tail -3 gpurun_out/bench_c5.log | cut -c1-2500
