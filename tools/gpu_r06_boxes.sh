#!/bin/bash
# Round 6: the C3 / C4 / C5 lines of the final tree on one more box (the pool's boxes differ by up to 4 % on this step:
# the numbers in DESIGN.md are quoted as ranges over the boxes visited).
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in c3 c4 c5; do
  python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$w', j['ms_per_step'], 'ms/step', j['value'], 'frac', r['frac'], 'all_gemm', r['all_gemm_frac'], 'step_frac', r['step_frac'])" | tee -a gpurun_out/r06_boxes_$1.log
done
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power" | head -4 >> gpurun_out/r06_boxes_$1.log
