#!/bin/bash
# Round 6, visit 12: token-major weight gradients for the Perceiver's fp32-stream operands and the patch convolutions:
# training parity tests, then C3 / C4 / C5 lines.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_fullsize_steps.py tests/test_hip_api.py tests/test_hip_openshape.py tests/test_hip_pnsa.py -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06_v12_pytest.log
for w in c3 c4 c5; do
  timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$w', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', r['step_frac'], 'loss', j.get('final_loss'))" | tee -a gpurun_out/r06_v12_bench.log
done
