#!/bin/bash
# Round 4: C3 step with the frozen blocks' LayerNorms folded into the GEMMs (on) vs as their own passes (off), same box, interleaved
mkdir -p gpurun_out; L=gpurun_out/lnfold_ab.log; : > $L
timeout 600 python -m pytest tests/test_hip_lnfold.py -q -x 2>&1 | tail -3 >> $L
for rep in 1 2; do for v in on off; do
  echo "== ln-fold $v rep $rep" >> $L
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ln-fold $v 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline'].get('step_frac'))" >> $L
done; done
cat $L
