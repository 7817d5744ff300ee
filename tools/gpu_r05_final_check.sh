#!/bin/bash
# Round 5, the last visit: the whole suite + smoke + the bench line (with detail and CPU baseline) + kernel statistics on HEAD.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05_pytest_gpu_final_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r05_smoke.log
timeout 900 python bench.py --detail gpurun_out/r05_bench_c3_detail.json > gpurun_out/r05_bench_c3.log 2>&1
tail -1 gpurun_out/r05_bench_c3.log > gpurun_out/r05_bench_c3_train_step.json
cut -c1-2800 gpurun_out/r05_bench_c3_train_step.json
rm -rf gpurun_out/r05_prof_c3
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_c3 -o r05 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r05_prof_c3 -name "*kernel_trace*" -delete
f=$(find gpurun_out/r05_prof_c3 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r05_bench_c3_kernel_stats.csv && head -4 $f | cut -c1-160
