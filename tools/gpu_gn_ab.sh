#!/bin/bash
# Round 4: in-step A/B of the persistent GEMM's tile-order group width (VL_GEMM_GN = N-tiles per group; default 4)
# on one box: C3 bench line per setting, twice, interleaved.
mkdir -p gpurun_out
: > gpurun_out/gn_ab.log
for rep in 1 2; do
  for gn in 0 8 16 2; do
    echo "== VL_GEMM_GN=$gn rep $rep" >> gpurun_out/gn_ab.log
    VL_GEMM_GN=$gn timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'])" >> gpurun_out/gn_ab.log
  done
done
cat gpurun_out/gn_ab.log
