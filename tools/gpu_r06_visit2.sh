#!/bin/bash
# Round 6, visit 2: k-rotation probe; the new / changed tests; GEMM parity tests on the rotated kernel.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_kstagger_probe.sh
timeout 1500 python -m pytest tests/test_hip_gemm_park.py tests/test_hip_lnfold.py tests/test_hip_f16.py "tests/test_hip_train.py::test_steps_with_the_frozen_towers_on_a_second_stream" "tests/test_hip_train.py::test_trained_blocks_carry_no_stale_layernorm_folds" "tests/test_hip_fullsize_steps.py::test_c3_at_its_literal_configuration_b1024_as_4x256_with_adamw" -q -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r06_v2_pytest.log
for i in 1 2; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06_v2_bench_$i.json
done
for f in gpurun_out/r06_v2_bench_*.json; do echo "$f: $(python -c "
import json,sys
j=json.load(open('$f')); r=j['roofline']
print(j['ms_per_step'], j['value'], r['achieved'], r['frac'], r['all_gemm_tflops'], r['step_frac'], j.get('final_loss'))")"; done
