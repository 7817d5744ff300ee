#!/bin/bash
# Probe build of the library (-DVL_GEMM_PROBE: the persistent GEMM's k-rotation mode / step become a run-time knob,
# `vl_gemm_probe_kstag[2]`) -> tools/bin/variants/libprobe.so; tools/gpu_kstagger_probe.sh puts it in the in-tree library's
# place for ONE process and restores it (the product has no library override and no knob).
set -e
cd "$(dirname "$0")/../vit-lens_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/bin/variants build_probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -DVL_GEMM_PROBE -x hip -c vl_gemm_park.hip -o build_probe/vl_gemm_park.hip.o
objs=$(ls build/*.o | grep -v vl_gemm_park)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/variants/libprobe.so $objs build_probe/vl_gemm_park.hip.o
ls -la ../../tools/bin/variants/libprobe.so
