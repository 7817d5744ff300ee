#!/bin/bash
# Measurement builds of the persistent GEMM (VL_PK_VARIANT bits) as complete libraries under tools/bin/variants/: on the GPU box a
# variant is copied over vit-lens_amd/vitlens_hip/libvitlens_hip.so between bench runs of ONE visit (same box: +-0.1 % run to run).
set -e
cd "$(dirname "$0")/../vit-lens_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/bin/variants
for v in "$@"; do
  mkdir -p build_v$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -DVL_PK_VARIANT=$v -x hip -c vl_gemm_park.hip -o build_v$v/vl_gemm_park.hip.o
  objs=$(ls build/*.o | grep -v vl_gemm_park)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/variants/libv$v.so $objs build_v$v/vl_gemm_park.hip.o
done
cp ../vitlens_hip/libvitlens_hip.so ../../tools/bin/variants/libv0.so
ls -la ../../tools/bin/variants/
