#!/bin/bash
# Round 6, visit 18: the Perceiver trainer keeps every LayerNorm / GEGLU output of the forward for the backward's weight
# gradients instead of recomputing them: training tests, then C4 / C5 with the previous trainer (tools/bin/train_before_v18.py)
# beside the new one on this box.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_fullsize_steps.py tests/test_hip_api.py tests/test_hip_openshape.py tests/test_hip_pnsa.py tests/test_hip_points.py -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06_v18_pytest.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'step_frac', j['roofline']['step_frac'], 'loss', j.get('final_loss'))"; }
P=vit-lens_amd/vitlens_hip/train.py
cp $P /tmp/train_new.py
for w in c4 c5; do
for r in 1 2; do
  cp tools/bin/train_before_v18.py $P; echo "== $w round $r recomputing" | tee -a gpurun_out/r06_v18_keep_ab.log; line $w | tee -a gpurun_out/r06_v18_keep_ab.log
  cp /tmp/train_new.py $P;             echo "== $w round $r keeping" | tee -a gpurun_out/r06_v18_keep_ab.log; line $w | tee -a gpurun_out/r06_v18_keep_ab.log
done
done
cp /tmp/train_new.py $P
