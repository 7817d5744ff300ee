#!/bin/bash
# Round 6, visit 3: same-box A/B of the persistent GEMM's two round-6 changes (k-rotation, aux prefetch in the last k-step) as
# separately built library variants, in the C3 step; parity tests on the product library; C4 / C5 with the padded token-major
# weight gradients.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
one() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('   ms/step', j['ms_per_step'], ' dominant TF/s', r['achieved'], ' all-GEMM', r['all_gemm_tflops'], ' loss', j.get('final_loss'))"; }
export -f one
bash tools/lib_ab.sh 2 "base kstag0 auxpf0 product kstag_slot2" -- bash -c one 2>&1 | tee gpurun_out/r06_v3_lib_ab_c3.log
timeout 2400 python -m pytest tests/test_hip_gemm_park.py tests/test_hip_lnfold.py tests/test_hip_f16.py tests/test_hip_ops.py tests/test_hip_train.py tests/test_hip_fullsize_steps.py -q -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/r06_v3_pytest.log
for w in c4 c5; do
  timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06_v3_bench_$w.json
  python -c "
import json
j=json.load(open('gpurun_out/r06_v3_bench_$w.json')); r=j['roofline']
print('$w', j['ms_per_step'], j['value'], r['all_gemm_tflops'], r['step_frac'], j.get('final_loss'))"
done
