#!/bin/bash
# Round-3 visit: attention parity + timing after the lone-key change; C4 / C5 bench lines with their CPU-baseline legs and
# rocprofv3 kernel stats of the same commands.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== attention parity =="
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_train.py -x -q -p no:cacheprovider -k "attn or attention" > gpurun_out/r03_attn_pytest.log 2>&1; tail -3 gpurun_out/r03_attn_pytest.log
echo "== attention timing =="
(timeout 120 python tools/attn_probe.py; L=256 timeout 120 python tools/attn_probe.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_attn_probe.log
for WL in c4 c5; do
  echo "== bench $WL =="
  timeout 1200 python bench.py --workload $WL --steps 5 --warmup 2 --detail gpurun_out/r03_bench_${WL}_detail.json > gpurun_out/r03_bench_$WL.log 2>&1
  tail -1 gpurun_out/r03_bench_$WL.log | cut -c1-3000
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_$WL -o p -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03_prof_$WL.log 2>&1
  cd $R
  find gpurun_out/r03_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r03_prof_$WL -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r03_bench_${WL}_kernel_stats.csv && head -12 $f | cut -c1-160
done
