#!/bin/bash
# Round 6, visit 21: attention backward (fused) with ONE memory round trip per item: branch-free staging loads, kernel arguments
# re-read through scalar loads, the next item's q / k rows fetched into registers under the tile loop; attention forward with
# the lone row's load off the critical path.  Attention tests, the probe under the old and the new library, training tests,
# then the C3 / C4 steps under both libraries.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -k "attn or attention" 2>&1 | tail -5 | tee gpurun_out/r06_v21_pytest_attn.log
for L in 257 256 197; do
  L=$L N=20 bash tools/lib_ab.sh 2 "attnbwd_before product" -- python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/L=$L  /" | tee -a gpurun_out/r06_v21_attn_probe_ab.log
done
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_fullsize_steps.py tests/test_hip_api.py -q -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r06_v21_pytest_train.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', j['roofline']['step_frac'], 'loss', j.get('final_loss'))"; }
for w in c3 c4; do
  bash tools/lib_ab.sh 2 "attnbwd_before product" -- bash -c "$(declare -f line); line $w" 2>&1 | tee -a gpurun_out/r06_v21_step_ab.log
done
