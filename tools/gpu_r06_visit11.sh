#!/bin/bash
# Round 6, visit 11: text-tower backward + module-API two-stream forward (new tests first, then the whole suite), the drop-in
# loop body against the fused step at b = 256, kernel statistics of C4 / C5.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_api.py -q -x -p no:cacheprovider -k "text_tower or second_stream or unlocked" 2>&1 | tail -15 | tee gpurun_out/r06_v11_pytest_new.log
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r06_v11_pytest_gpu_full.log
for v in api step; do
  timeout 600 python bench.py --batch 256 --via $v --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('b=256 via $v:', j['ms_per_step'], 'ms/step', j['value'], 'triplets/s')" | tee -a gpurun_out/r06_api_vs_step.log
done
for WL in c4 c5; do
  rm -rf gpurun_out/r06_prof_$WL
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_$WL -o r06 -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/r06_rocprof_$WL.log 2>&1
  cd $R
  find gpurun_out/r06_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r06_prof_$WL -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r06_bench_${WL}_kernel_stats.csv && head -8 $f | cut -c1-160
done
