#!/bin/bash
# Round 6, visit 5: the one-wave-per-SIMD long-K kernel (vl_gemm_w4.hip): parity, probe against the 8-wave kernel and the vendor
# GEMM, then the C3 step with and without it (library variants, one box); attention parity tests on the DMA forward.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_gemm_park.py tests/test_hip_lnfold.py -q -x -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r06_v5_pytest_gemm.log
timeout 600 python tools/longk_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_longk_probe.log
one() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('   ms/step', j['ms_per_step'], ' dominant TF/s', r['achieved'], ' all-GEMM', r['all_gemm_tflops'], ' loss', j.get('final_loss'))"; }
export -f one
bash tools/lib_ab.sh 2 "now4 product" -- bash -c one 2>&1 | tee gpurun_out/r06_v5_lib_ab_w4.log
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_train.py -q -p no:cacheprovider -k "attention or attn or step or tower" 2>&1 | tail -8 | tee gpurun_out/r06_v5_pytest_attn.log
