#!/bin/bash
# GPU-box visit for an iteration: selected tests + short bench lines of every workload
set +e
mkdir -p gpurun_out
if [ -n "$PYTEST_K" ]; then      # tests only on request (the full suite takes 6-7 GPU-minutes: use tools/gpu_tests.sh for it)
timeout 900 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider -k "$PYTEST_K" 2>&1 | tail -60 | cut -c1-400 > gpurun_out/pytest_iter.log
tail -25 gpurun_out/pytest_iter.log
fi
for WL in ${WLS:-c2 c3 c4 c5}; do
  timeout 900 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --detail gpurun_out/bench_${WL}_detail.json > gpurun_out/bench_$WL.log 2>&1
  tail -1 gpurun_out/bench_$WL.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$WL', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['roofline'].get('step_tflops'), 'TF/s step', d['roofline']['achieved'], 'TF/s dom')
except Exception as e:
    print('$WL failed', e)
"
done
