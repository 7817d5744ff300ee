#!/bin/bash
# Round 6, visit 13: GELU epilogues on packed-fp32 pairs (bit-identical arithmetic): every test that touches a GEMM epilogue or
# a training step, the GradScaler loop body, the padded token-major dW; then the same-box A/B against the scalar-GELU library
# (tools/bin/variants/libgelu_scalar.so = the library of the commit before) on C3 / C4 / C5.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06_v13_pytest.log
line() { python bench.py --workload $1 --steps $2 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['ms_per_step'], 'ms/step', j['value'], 'loss', j.get('final_loss'))"; }
bash tools/lib_ab.sh 1 "gelu_scalar product" -- python tools/epilogue_checksum.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_v13_checksums.log
for w in c3 c4 c5; do
  bash tools/lib_ab.sh 2 "gelu_scalar product" -- bash -c "$(declare -f line); line $w 8" 2>&1 | tee -a gpurun_out/r06_v13_gelu_ab.log
done
