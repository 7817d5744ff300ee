#!/bin/bash
# Round 6, visit 1: (a) the whole GPU suite on the restructured steps (overlap default ON, per-block LayerNorm folding);
# (b) the bench line on the default path and with --no-overlap-frozen, interleaved; (c) the vendor yardstick
# (tools/vendor_gemm_yardstick.py) + the vendor kernels' names from a rocprofv3 kernel trace of a short run of it.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r06_v1_pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r06_v1_smoke.log
for i in 1 2; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r06_v1_bench_overlap_$i.json
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-overlap-frozen 2>&1 | tail -1 > gpurun_out/r06_v1_bench_serial_$i.json
done
for f in gpurun_out/r06_v1_bench_*.json; do echo "$f: $(python -c "
import json,sys
j=json.load(open('$f')); r=j['roofline']
print(j['ms_per_step'], j['value'], r['achieved'], r['frac'], r['all_gemm_tflops'], r['step_frac'], r['measured_on'][:30], j.get('final_loss'))")"; done
YS_SUSTAIN=1 timeout 900 python tools/vendor_gemm_yardstick.py 2>&1 | tee gpurun_out/r06_vendor_gemm_yardstick.log
rm -rf gpurun_out/r06_prof_ys
cd /tmp && YS_ROUNDS=2 YS_REPS=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_ys -o ys -- python $R/tools/vendor_gemm_yardstick.py > $R/gpurun_out/r06_rocprof_ys.log 2>&1
cd $R
find gpurun_out/r06_prof_ys -name "*kernel_trace*" -delete
f=$(find gpurun_out/r06_prof_ys -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r06_vendor_gemm_yardstick_kernel_stats.csv && head -14 $f | cut -c1-260
