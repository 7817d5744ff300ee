#!/bin/bash
# Round 6, visit 25: the leftover-row GEMM with 16 k-slices requested per batch and the leftover rows' statistics in one pass
# over registers: GEMM / LayerNorm-fold / training tests, the C3 and C4 steps under the previous library and the new one, kernel
# statistics of the new one (one stream).
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests/test_hip_gemm.py tests/test_hip_gemm_park.py tests/test_hip_lnfold.py tests/test_hip_ops.py tests/test_hip_train.py tests/test_hip_fullsize_steps.py tests/test_hip_towers.py -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_v25_pytest.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', j['roofline']['step_frac'], 'loss', j.get('final_loss'))"; }
for w in c3 c4; do
  bash tools/lib_ab.sh 2 "tail_before product" -- bash -c "$(declare -f line); line $w" 2>&1 | tee -a gpurun_out/r06_v25_step_ab.log
done
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_v25_prof_c3 -o r06 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/r06_v25_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r06_v25_prof_c3 -name "*kernel_trace*" -delete
f=$(find gpurun_out/r06_v25_prof_c3 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r06_v25_bench_c3_kernel_stats.csv && grep -i "tail\|row_stats\|64, 64" $f | cut -c1-160
