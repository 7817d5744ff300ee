#!/bin/bash
# Round 6: k-rotation probe (tools/kstagger_probe.py) on the probe build, then the step A/B of the product default.
set +e
mkdir -p gpurun_out
L=vit-lens_amd/vitlens_hip/libvitlens_hip.so
cp $L /tmp/lib_orig.so && cp tools/bin/variants/libprobe.so $L
timeout 900 python tools/kstagger_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_kstagger_probe.log
cp /tmp/lib_orig.so $L
