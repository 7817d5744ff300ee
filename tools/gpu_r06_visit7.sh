#!/bin/bash
# Round 6, visit 7: two-stream backward (tests, then the C3 step default / --no-overlap-backward / --no-overlap-frozen interleaved
# on one box), attention parity on the DMA forward, C4 / C5 with one and two micro-batches.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_ops.py tests/test_hip_fullsize_steps.py tests/test_hip_api.py -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r06_v7_pytest.log
one() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('   ms/step', j['ms_per_step'], ' value', j['value'], ' dominant TF/s', r['achieved'], ' all-GEMM', r['all_gemm_tflops'], ' step_frac', r['step_frac'], ' loss', j.get('final_loss'))"; }
for r in 1 2; do
  echo "== c3 default"; one
  echo "== c3 --no-overlap-backward"; one --no-overlap-backward
  echo "== c3 --no-overlap-frozen"; one --no-overlap-frozen
done 2>&1 | tee gpurun_out/r06_v7_step_ab_overlap_backward.log
for w in c4 c5; do
  echo "== $w default"; one --workload $w
  echo "== $w --no-overlap-frozen"; one --workload $w --no-overlap-frozen
  echo "== $w two micro-batches"; one --workload $w --micro-batch $([ $w = c4 ] && echo 128 || echo 64)
done 2>&1 | tee gpurun_out/r06_v7_c4_c5.log
