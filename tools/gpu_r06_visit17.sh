#!/bin/bash
# Round 6, visit 17: kernel statistics of C4 / C5 on the tree with the token-major weight gradients everywhere (one stream, so
# that a duration is a kernel's own), 3 timed steps + 1 warm-up each.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for WL in c4 c5; do
  rm -rf gpurun_out/r06_prof_$WL
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_$WL -o r06 -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/r06_rocprof_$WL.log 2>&1
  cd $R
  find gpurun_out/r06_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r06_prof_$WL -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r06_v17_bench_${WL}_kernel_stats.csv && head -8 $f | cut -c1-160
done
