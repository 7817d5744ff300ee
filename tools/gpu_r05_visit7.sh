#!/bin/bash
# Round 5 experiment: the frozen towers' forwards on a second HIP stream beside the trainable tower's forward (power smoothing:
# the step averages 1 335 W against the 1 400 W the GEMM phases are capped at).  Interleaved A/B, same box; the loss must agree.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r05v7_overlap_ab.log
for rep in 1 2; do for v in "" "--overlap-frozen"; do
  echo "rep $rep [${v:-serial}]: $(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'triplets/s  loss', d['final_loss'], ' dominant', d['roofline']['achieved'], 'all-GEMM', d['roofline']['all_gemm_tflops'])")" | tee -a gpurun_out/r05v7_overlap_ab.log
done; done
