#!/bin/bash
# Round 6, visit 10: whole GPU suite + smoke on the tree with the ABI collectives and the two-stream backward; bench line.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r06_v10_pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06_v10_bench_c3.json
python -c "
import json
j=json.load(open('gpurun_out/r06_v10_bench_c3.json')); r=j['roofline']
print(j['ms_per_step'], j['value'], r['achieved'], r['frac'], r['all_gemm_tflops'], r['step_frac'], j['config']['backward'])"
