#!/bin/bash
# One GPU-box visit: parity tests, micro-benchmarks, bench line, rocprof summary.  Logs -> gpurun_out/
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" > gpurun_out/env.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/env.log
echo "== pytest gpu ==" 
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log | cut -c1-300
fi
echo "== kernel bench =="
if [ -z "$SKIP_KB" ]; then timeout 600 python tools/kernel_bench.py ${KB_ARGS:---quick} > gpurun_out/kernel_bench.log 2>&1; tail -40 gpurun_out/kernel_bench.log; fi
echo "== bench =="
timeout 600 python bench.py --steps 5 --warmup 2 --detail gpurun_out/bench_detail.json > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
echo "== bench c3 =="
timeout 900 python bench.py --workload c3 --batch ${C3_BATCH:-1024} --steps 3 --warmup 1 --detail gpurun_out/bench_c3_detail.json > gpurun_out/bench_c3.log 2>&1
tail -3 gpurun_out/bench_c3.log | cut -c1-1800
echo "== rocprof =="
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; 
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -30 $f; done
echo "== hbm traffic (PMC, separate passes) =="
[ -n "$SKIP_PMC" ] && exit 0
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE "gemm_nt_persist2_kernel<256, 256, 2, 4, 0>" gpurun_out/hbm_traffic.json 65792,4096,1024
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete
# keep only the small summaries
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
