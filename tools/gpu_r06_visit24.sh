#!/bin/bash
# Round 6, visit 24: the fused attention backward as kept (scalar-loaded arguments, branch-free one-batch staging, K fragments
# out of the staged image; no register prefetch) and the forward with the lone row's load off the critical path: attention and
# training tests, the probe and the C3 step under the library of before visit 21 and the new one.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -k "attn or attention" 2>&1 | tail -3 | tee gpurun_out/r06_v24_pytest_attn.log
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_fullsize_steps.py -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06_v24_pytest_train.log
for SEQ in 257 256; do
  export L=$SEQ N=20
  bash tools/lib_ab.sh 2 "attnbwd_before product" -- python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/L=$SEQ  /" | tee -a gpurun_out/r06_v24_attn_probe_ab.log
done
unset L N
(./tools/bin/attn_phase_prof 257; ./tools/bin/attn_phase_prof 256) 2>&1 | tee gpurun_out/r06_v24_attn_phase_timeline.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', j['roofline']['step_frac'], 'loss', j.get('final_loss'))"; }
bash tools/lib_ab.sh 2 "attnbwd_before product" -- bash -c "$(declare -f line); line c3" 2>&1 | tee -a gpurun_out/r06_v24_step_ab.log
