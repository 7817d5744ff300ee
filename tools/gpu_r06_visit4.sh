#!/bin/bash
# Round 6, visit 4: attention forward with LDS-DMA staging (parity tests + probe, old library variant beside it on one box);
# the C5 bench-geometry test that failed with the padded token-major dW; counters of the vendor GEMM next to ours.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider -k "attn" 2>&1 | tail -8 | tee gpurun_out/r06_v4_pytest_attn.log
bash tools/lib_ab.sh 2 "attn_old product" -- bash -c 'N=20 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | grep fwd; L=256 N=20 python tools/attn_probe.py 2>&1 | grep fwd' 2>&1 | tee gpurun_out/r06_v4_attn_ab.log
timeout 900 python -m pytest "tests/test_hip_fullsize_steps.py::test_c5_step_at_bench_geometry_is_finite_and_split_invariant" tests/test_hip_towers.py -q -x -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/r06_v4_pytest_c5.log
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/r06_ys_pmc_$tag
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r06_ys_pmc_$tag -o g -- python $R/tools/yardstick_pmc_target.py > $R/gpurun_out/r06_ys_pmc_$tag.log 2>&1
done
cd $R
python tools/yardstick_pmc_summary.py "gpurun_out/r06_ys_pmc_*" gpurun_out/r06_vendor_vs_ours_pmc.json 2>&1 | tee gpurun_out/r06_vendor_vs_ours_pmc.txt
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
