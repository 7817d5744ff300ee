#!/bin/bash
# Round 5, first GPU visit (decisions, not evidence): (1) the one-wave-per-SIMD GEMM as a TEST, then as a measurement (isolated +
# in the step, interleaved); (2) HBM traffic of the GEMM epilogue family and the fused attention backward on the HEAD tree
# (PMC passes over the isolated probes: seconds, not bench runs); (3) watts and MHz; (4) the whole suite with the LayerNorm
# folding on, sequentially.  Everything lands in gpurun_out/r05v1_*.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }

stamp "1. pk4 parity tests"
timeout 420 python -m pytest tests/test_hip_gemm_pk4.py -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/r05v1_pk4_pytest.log
stamp "1b. pk4 probe (vs the 8-wave kernel, timing)"
timeout 300 python tools/pk4_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05v1_pk4_probe.log
if grep -q "ALL OK" gpurun_out/r05v1_pk4_probe.log; then
  stamp "1c. C3 step with VL_GEMM_PK4=1 against the default, interleaved"
  : > gpurun_out/r05v1_pk4_step_ab.log
  for rep in 1 2; do for v in 0 1; do
    echo "VL_GEMM_PK4=$v rep $rep: $(VL_GEMM_PK4=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"achieved\"], d[\"roofline\"][\"all_gemm_tflops\"])")" | tee -a gpurun_out/r05v1_pk4_step_ab.log
  done; done
  stamp "1d. SQ counters: LDS instructions per MFMA, cfg 8 vs cfg 14"
  cd /tmp
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
    tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
    rm -rf $R/gpurun_out/r05v1_pmc_$tag
    KB_CFGS=8,14 KB_ROUNDS=2 KB_CASES=proj:res_bf16,fc:gelu+dsave,qkv:bf16 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r05v1_pmc_$tag -o g -- python $R/tools/gemm_probe.py > /dev/null 2>&1
  done
  cd $R
  python tools/pmc_summary.py "gpurun_out/r05v1_pmc_*" gpurun_out/r05v1_gemm_pmc.json 2>&1 | tee gpurun_out/r05v1_gemm_pmc_summary.txt
  find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
fi

stamp "2. HBM traffic on HEAD: GEMM family + attention backward (isolated probes)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r05v1_traffic_$c $R/gpurun_out/r05v1_traffic_attn_$c
  KB_CFGS=8 KB_ROUNDS=2 KB_CASES=proj:res_bf16,fc:gelu,fc:gelu+dsave,dproj:dgelu_saved,qkv:bf16,out:res_bf16,dfc:bf16 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r05v1_traffic_$c -o t -- python $R/tools/gemm_probe.py > /dev/null 2>&1
  N=3 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r05v1_traffic_attn_$c -o t -- python $R/tools/attn_probe.py > /dev/null 2>&1
done
cd $R
python tools/traffic_probe_summary.py gpurun_out/r05v1_traffic_FETCH_SIZE gpurun_out/r05v1_traffic_WRITE_SIZE gpurun_out/r05v1_traffic_attn_FETCH_SIZE gpurun_out/r05v1_traffic_attn_WRITE_SIZE 2>&1 | tee gpurun_out/r05v1_hbm_traffic_probes.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete

stamp "3. watts and MHz"
bash tools/gpu_power_trace.sh 2>&1 | tail -8 | tee gpurun_out/r05v1_power_trace_summary.txt
cp gpurun_out/power_trace.log gpurun_out/r05v1_power_trace.log

stamp "4. whole suite with the LayerNorm folding on (sequential)"
VL_LN_FOLD=1 bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05v1_pytest_gpu_lnfold.log
stamp "done"
