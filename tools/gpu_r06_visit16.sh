#!/bin/bash
# Round 6, visit 16: the whole suite on the tree with the packed GELU forms, the GradScaler loop-body test and the point
# tokenizer's token-major weight gradients; C5 with the tokenizer's previous dW path (tools/bin/points_before_v16.py) beside it.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r06_v16_pytest.log
line() { python bench.py --workload $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'loss', j.get('final_loss'))"; }
P=vit-lens_amd/vitlens_hip/points.py
cp $P /tmp/points_new.py
for r in 1 2; do
  cp tools/bin/points_before_v16.py $P; echo "== round $r transposing dW" | tee -a gpurun_out/r06_v16_c5_ab.log; line c5 | tee -a gpurun_out/r06_v16_c5_ab.log
  cp /tmp/points_new.py $P;             echo "== round $r token-major dW" | tee -a gpurun_out/r06_v16_c5_ab.log; line c5 | tee -a gpurun_out/r06_v16_c5_ab.log
done
cp /tmp/points_new.py $P
