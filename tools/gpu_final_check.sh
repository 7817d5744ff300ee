#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
PYTEST_K="gemm or qkv" bash tools/gpu_tests.sh | tail -3
python tools/kernel_bench.py --quick 2>&1 | grep -E "^gemm.*cfg-1|^cfg-1" > gpurun_out/kernel_bench_final.log; cat gpurun_out/kernel_bench_final.log
rm -rf gpurun_out/pmc_WRITE_SIZE
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_WRITE_SIZE -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_WRITE_SIZE.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE "gemm_nt_persist2_kernel<256, 256, 2, 4, 0>" gpurun_out/hbm_traffic.json 65792,4096,1024 | cut -c1-700
tail -1 gpurun_out/pmc_WRITE_SIZE.log | cut -c1-200
