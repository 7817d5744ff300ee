#!/bin/bash
# Round 6, visit 20: 16-byte fp32 column sums and LayerNorm parameter gradients (bf16 dy, fp32 x): their tests + the training
# tests, then C4 / C5 with the library of the commit before (tools/bin/variants/libbwd_before.so) beside the new one.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_gemm_tn.py tests/test_hip_train.py tests/test_hip_fullsize_steps.py tests/test_hip_api.py -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06_v20_pytest.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', j['roofline']['step_frac'], 'loss', j.get('final_loss'))"; }
for w in c4 c5; do
  bash tools/lib_ab.sh 2 "bwd_before product" -- bash -c "$(declare -f line); line $w" 2>&1 | tee -a gpurun_out/r06_v20_colsum_ab.log
done
