#!/bin/bash
# Round 5, after the evidence visit: the tests added since (full-size C3 / C4 backward given the upstream gradient - the oracle's
# autograd at ViT-L runs on the host -, the C client of the ABI) and rocprofv3 kernel statistics of the C4 / C5 bench commands.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
VL_RECORD_ERRS=gpurun_out/r05_errs timeout 1500 python -m pytest tests/test_hip_abi_c_client.py tests/test_hip_fullsize_steps.py -q -s -p no:cacheprovider -k "c_client or given_upstream" 2>&1 | grep -v amdgpu.ids | grep -E "given_upstream|vs host|status|ALL OK|passed|failed|^E  |Error" | cut -c1-1200 | tee gpurun_out/r05v5_tests.log
for WL in c4 c5; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_$WL -o r05 -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05_rocprof_$WL.log 2>&1
  cd $R
  find gpurun_out/r05_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r05_prof_$WL -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r05_bench_${WL}_kernel_stats.csv && head -8 $f | cut -c1-160
done
