L=vit-lens_amd/vitlens_hip/libvitlens_hip.so
cp $L /tmp/lib_orig.so && cp tools/bin/variants/libprof.so $L
LNFOLD=0 KSTEP=1 timeout 200 python tools/gemm_phase_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_kstep_probe.log
cp /tmp/lib_orig.so $L
