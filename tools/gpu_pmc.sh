#!/bin/bash
set +e
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD)_[A-Z0-9_a-z]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc2/counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o a -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o b -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o c -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2/c.log 2>&1
cd $GRAFT_REPO_ROOT; ls -la gpurun_out/pmc2; tail -3 gpurun_out/pmc2/a.log gpurun_out/pmc2/b.log gpurun_out/pmc2/c.log | cut -c1-200
rm -f gpurun_out/pmc2/*kernel_trace.csv
