#!/bin/bash
# on the GPU box: k-loop / epilogue cycles of the persistent GEMM (measurement build) -> gpurun_out/r04b_gemm_phase_prof.log
L=vit-lens_amd/vitlens_hip/libvitlens_hip.so
cp $L /tmp/lib_orig.so && cp tools/bin/variants/libprof.so $L
python tools/gemm_phase_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_gemm_phase_prof.log
cp /tmp/lib_orig.so $L
