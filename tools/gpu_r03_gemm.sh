#!/bin/bash
# Round-3 GEMM visit 1: parity of the ping-pong kernel (cfg 10) and of the L2 prefetch, A/B timings, SQ counters.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== parity (cfg 8 + cfg 10) =="
timeout 900 python -m pytest tests/test_hip_gemm_park.py -x -q -p no:cacheprovider > gpurun_out/r03_gemm_pytest.log 2>&1; tail -5 gpurun_out/r03_gemm_pytest.log
echo "== parity with the L2 prefetch on =="
VL_GEMM_PF=3 timeout 900 python -m pytest tests/test_hip_gemm_park.py -x -q -p no:cacheprovider -k "not identity" > gpurun_out/r03_gemm_pytest_pf.log 2>&1; tail -3 gpurun_out/r03_gemm_pytest_pf.log
echo "== split-K partials on the ping-pong kernel =="
VL_GEMM_PP=1 timeout 600 python -m pytest tests/test_hip_ops.py -x -q -p no:cacheprovider -k "gemm" > gpurun_out/r03_gemm_pytest_pp_ops.log 2>&1; tail -3 gpurun_out/r03_gemm_pytest_pp_ops.log
echo "== A/B =="
(KB_TAG=base timeout 300 python tools/gemm_probe.py
 for pf in 2 3 5 8; do VL_GEMM_PF=$pf KB_TAG=pf$pf KB_CFGS=8 KB_CASES=fc:bf16,fc:gelu+save,proj:res_bf16,sq8k:bf16 timeout 300 python tools/gemm_probe.py; done
 for d in 0 2 8; do VL_PP_DELAY=$d KB_TAG=delay$d KB_CFGS=10 KB_CASES=fc:bf16,fc:gelu+save,proj:res_bf16 timeout 300 python tools/gemm_probe.py; done
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm_probe.log
echo "== SQ counters (fc gelu+save, cfg 8 vs cfg 10) =="
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
  KB_ROUNDS=2 KB_CASES=fc:gelu+save,fc:bf16 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r03_gemm_pmc_$tag -o g -- python $R/tools/gemm_probe.py > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r03_gemm_pmc_summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_gemm_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt_p" not in k: continue
        k = k[k.index("gemm_nt_p"):][:26] + " grid=" + r.get("Grid_Size", "?") 
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in sorted(agg.items()):
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
