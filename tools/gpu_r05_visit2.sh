#!/bin/bash
# Round 5, second GPU visit: the tree after the GEMM clean-up (one main loop, no environment switches), the fp16 text tower and
# the LayerNorm folding as default.  (1) the new fp16 tests + the text-tower accuracy numbers, (2) the WHOLE suite, sequential,
# (3) C3 step A/B: text operands f16 / bf16x2 / bf16, folding on / off, interleaved on this box.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }
stamp "1. fp16 entries + text tower accuracy"
timeout 600 python -m pytest tests/test_hip_f16.py tests/test_hip_towers.py -q -x -s -p no:cacheprovider -k "f16 or text" 2>&1 | grep -v amdgpu.ids | grep -E "text tower cos|passed|failed|Error|error|assert" | tee gpurun_out/r05v2_f16_tests.log
stamp "2. whole suite (sequential)"
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05v2_pytest_gpu_full.log
stamp "3. C3 step A/B (6 steps each, interleaved, two rounds)"
: > gpurun_out/r05v2_step_ab.log
for rep in 1 2; do
  for v in "--text-arith f16" "--text-arith bf16x2" "--text-arith bf16" "--text-arith f16 --ln-fold off"; do
    echo "rep $rep [$v]: $(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"achieved\"], d[\"roofline\"][\"all_gemm_tflops\"], d[\"roofline\"][\"step_tflops\"])")" | tee -a gpurun_out/r05v2_step_ab.log
  done
done
stamp "4. text tower time per 1024 captions"
timeout 120 python tools/text_tower_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05v2_text_tower_time.log
stamp "5. smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
stamp "done"
