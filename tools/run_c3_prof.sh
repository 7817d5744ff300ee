set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -o r02a -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/rocprof_c3.log | cut -c1-1500
find gpurun_out/prof_c3 -name "*kernel_trace*" -delete
for f in $(find gpurun_out/prof_c3 -name "*kernel_stats*.csv" | head -1); do head -45 $f | cut -c1-260; done
