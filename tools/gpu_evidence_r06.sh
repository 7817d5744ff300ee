#!/bin/bash
# Round-6 evidence in ONE GPU-box visit on the FINAL tree (everything the bench line, DESIGN.md section 7 and profiles/README.md
# cite as r06_*): the whole GPU suite (sequential) + smoke, the driver's bench line (C3 train step, with its CPU-baseline leg) and
# its per-shape detail, rocprofv3 kernel statistics of the same command, the GEMM probe + two SQ counter passes, the k-loop /
# epilogue cycle split (measurement build), attention timings + phase timeline, HBM traffic of the bench command (FETCH_SIZE /
# WRITE_SIZE in separate passes), watts and MHz, and the C2 / C4 / C5 lines.  Everything lands in gpurun_out/r06_*; the
# summaries are copied to profiles/ afterwards.  Per-kernel statistics and counter passes run the bench with --no-overlap-frozen: with the
# default two-stream schedule kernels of both streams share the chip and a dispatch's duration / counters are not its own.   usage: bash tools/gpu_evidence_r06.sh [quick]   (quick: no traffic / C2,4,5)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }

stamp "suite (sequential) + smoke"
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r06_pytest_gpu_final_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r06_smoke.log

stamp "bench (driver default = C3)"
timeout 900 python bench.py --detail gpurun_out/r06_bench_c3_detail.json > gpurun_out/r06_bench_c3.log 2>&1
tail -1 gpurun_out/r06_bench_c3.log > gpurun_out/r06_bench_c3_train_step.json
cut -c1-2600 gpurun_out/r06_bench_c3_train_step.json

stamp "rocprof stats of the same command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_c3 -o r06 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/r06_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r06_prof_c3 -name "*kernel_trace*" -delete
f=$(find gpurun_out/r06_prof_c3 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r06_bench_c3_kernel_stats.csv && head -14 $f | cut -c1-200

stamp "GEMM probe + SQ counters"
CASES=proj:res_bf16,proj:res_stats,fc:gelu,fc:ln_gelu,fc:gelu+dsave,fc:ln_dsave,dproj:dgelu_saved,qkv:bf16,qkv:ln_bf16,out:res_bf16,out:res_stats,dfc:bf16,sq8k:bf16
KB_TAG=r06 KB_CFGS=8 KB_ROUNDS=9 KB_CASES=$CASES timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm_probe.log
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/r06_pmc_$tag
  KB_CFGS=8 KB_ROUNDS=2 KB_CASES=proj:res_bf16,proj:res_stats,fc:gelu,fc:ln_gelu,fc:gelu+dsave,fc:ln_dsave,dproj:dgelu_saved,qkv:bf16,qkv:ln_bf16 timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r06_pmc_$tag -o g -- python $R/tools/gemm_probe.py > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py "gpurun_out/r06_pmc_*" gpurun_out/r06_gemm_pmc.json 2>&1 | tee gpurun_out/r06_gemm_pmc_summary.txt
for d in gpurun_out/r06_pmc_*; do for f in $(find $d -name "*counter_collection.csv"); do cp $f gpurun_out/$(basename $d)_counters.csv; done; done
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete

stamp "k-loop / epilogue cycles (measurement build)"
if [ -f tools/bin/variants/libprof.so ]; then
  L=vit-lens_amd/vitlens_hip/libvitlens_hip.so
  cp $L /tmp/lib_orig.so && cp tools/bin/variants/libprof.so $L
  timeout 200 python tools/gemm_phase_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm_phase_prof.log
  cp /tmp/lib_orig.so $L
fi

stamp "attention"
(timeout 120 python tools/attn_probe.py; L=256 timeout 120 python tools/attn_probe.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_attn_probe.log
[ -x tools/bin/attn_phase_prof ] && (./tools/bin/attn_phase_prof 257; ./tools/bin/attn_phase_prof 256) 2>&1 | tee gpurun_out/r06_attn_phase_timeline.log

stamp "watts and MHz"
bash tools/gpu_power_trace.sh 2>&1 | tail -6 | tee gpurun_out/r06_power_trace_final_summary.txt
cp gpurun_out/power_trace.log gpurun_out/r06_power_trace_final.log

if [ "$1" != "quick" ]; then
stamp "hbm traffic of the bench command (PMC, separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r06_hbm_traffic_c3.json \
  "gemm_nt_pk_kernel<3, 20, false>|65792,1024,4096|hi|3,0" "gemm_nt_pk_kernel<3, 0, false>|65792,1024,4096|hi|3,0" "gemm_nt_pk_kernel<0, 11, false>|65792,4096,1024|all|0,1" \
  "gemm_nt_pk_kernel<0, 10, false>|65792,3072,1024|all|0,0" "gemm_nt_pk_kernel<0, 14, false>|65792,4096,1024|all|0,4" "gemm_nt_pk_kernel<6, 4, false>|65792,4096,1024|all|6,4" \
  "attn_bwd_fused_kernel|256,16,257,64|all" "attn_fwd_kernel<64, true, false, false, true>|256,16,257,64|all"
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +4M -delete
for WL in c2 c4 c5; do
  stamp "bench $WL"
  timeout 1200 python bench.py --workload $WL --steps 5 --warmup 2 > gpurun_out/r06_bench_$WL.log 2>&1
  tail -1 gpurun_out/r06_bench_$WL.log > gpurun_out/r06_bench_${WL}_train_step.json
  cut -c1-1200 gpurun_out/r06_bench_${WL}_train_step.json
done
for WL in c4 c5; do
  stamp "rocprof stats $WL (one stream)"
  rm -rf gpurun_out/r06_prof_$WL
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_$WL -o r06 -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-frozen > $R/gpurun_out/r06_rocprof_$WL.log 2>&1
  cd $R
  find gpurun_out/r06_prof_$WL -name "*kernel_trace*" -delete
  f=$(find gpurun_out/r06_prof_$WL -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r06_bench_${WL}_kernel_stats.csv
done
fi
stamp "micro-batch size on the two-stream schedule (128 / 512 against the default 256)"
for MB in 128 512; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --micro-batch $MB 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('micro-batch $MB:', j['ms_per_step'], 'ms/step', j['value'], 'triplets/s')" | tee -a gpurun_out/r06_microbatch_sweep.log
done
stamp "done"
