#!/bin/bash
# Round-3 evidence in one GPU-box visit (final tree): the driver's bench line (C3 train step, with its CPU-baseline leg),
# rocprofv3 kernel stats of the same command, HBM traffic of the dominant GEMM from two separate --pmc passes, and the C4 / C5
# lines.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench (driver default = C3) =="
timeout 900 python bench.py --detail gpurun_out/r03_bench_c3_detail.json > gpurun_out/r03_bench_c3.log 2>&1
tail -1 gpurun_out/r03_bench_c3.log | cut -c1-2500
echo "== rocprof stats of the same command =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_c3 -o r03 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r03_prof_c3 -name "*kernel_trace*" -delete
f=$(find gpurun_out/r03_prof_c3 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r03_bench_c3_kernel_stats.csv && head -14 $f | cut -c1-200
if [ "$1" != "quick" ]; then
echo "== hbm traffic (PMC, separate passes) =="
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r03_hbm_traffic_c3.json \
  "gemm_nt_pk_kernel<3, 0,|65792,1024,4096|hi|3,0" "gemm_nt_pk_kernel<0, 1,|65792,4096,1024|hi|0,1" \
  "gemm_nt_pk_kernel<0, 4,|65792,4096,1024|all|0,4" "gemm_nt_pk_kernel<6, 4,|65792,4096,1024|all|6,4"
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +4M -delete
for WL in c4 c5; do
  echo "== bench $WL =="
  timeout 1200 python bench.py --workload $WL --steps 5 --warmup 2 > gpurun_out/r03_bench_$WL.log 2>&1
  tail -1 gpurun_out/r03_bench_$WL.log | cut -c1-1500
done
fi
