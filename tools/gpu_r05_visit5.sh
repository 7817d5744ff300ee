#!/bin/bash
# Round 5, after the evidence visit: the leftover rows' LayerNorm moved into the row-statistics launch (one tiny launch less per
# folded LayerNorm) -> whole suite again, step A/B against the separate launch (flipped from outside the product), then the
# bench line + kernel statistics of the final tree again; the tests added since the evidence visit (full-size C3 / C4 backward
# given the upstream gradient, the C client of the ABI) are part of the suite; rocprofv3 kernel statistics of C4 / C5.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }
stamp "suite (sequential) + smoke"
VL_RECORD_ERRS=gpurun_out/r05_errs bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05_pytest_gpu_final_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r05_smoke.log
stamp "the new tests, verbosely"
VL_RECORD_ERRS=gpurun_out/r05_errs timeout 900 python -m pytest tests/test_hip_abi_c_client.py tests/test_hip_fullsize_steps.py -q -s -p no:cacheprovider -k "c_client or given_upstream" 2>&1 | grep -v amdgpu.ids | grep -E "given_upstream|vs host|status|ALL OK|passed|failed|^E  |Error" | cut -c1-1500 | tee gpurun_out/r05v5_new_tests.log
stamp "step A/B: leftover-row LayerNorm inside the row-statistics launch (base) vs its own launch"
timeout 900 python tools/step_ab.py ln_left_separate 2>&1 | grep -v amdgpu.ids | grep "^rep" | tee gpurun_out/r05v5_step_ab_ln_left.log
stamp "bench (driver default = C3)"
timeout 900 python bench.py --detail gpurun_out/r05_bench_c3_detail.json > gpurun_out/r05_bench_c3.log 2>&1
tail -1 gpurun_out/r05_bench_c3.log > gpurun_out/r05_bench_c3_train_step.json
cut -c1-2600 gpurun_out/r05_bench_c3_train_step.json
stamp "rocprof stats of the same command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_c3 -o r05 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r05_prof_c3 -name "*kernel_trace*" -delete
f=$(ls -t $(find gpurun_out/r05_prof_c3 -name "*kernel_stats*.csv") | head -1)
[ -n "$f" ] && cp $f gpurun_out/r05_bench_c3_kernel_stats.csv && head -6 $f | cut -c1-160
for WL in c4 c5; do
  stamp "rocprof stats $WL"
  rm -rf gpurun_out/r05_prof_$WL
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_$WL -o r05 -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05_rocprof_$WL.log 2>&1
  cd $R
  find gpurun_out/r05_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r05_prof_$WL -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r05_bench_${WL}_kernel_stats.csv && head -5 $f | cut -c1-160
done
stamp "done"
