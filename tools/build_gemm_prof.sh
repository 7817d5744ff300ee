#!/bin/bash
# Measurement build of the library with the persistent GEMM's phase counters (-DVL_GEMM_PROF) -> tools/bin/variants/libprof.so;
# tools/gemm_phase_prof.py copies it over the in-tree library for ONE process and restores it.
set -e
cd "$(dirname "$0")/../vit-lens_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/bin/variants build_prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -DVL_GEMM_PROF -x hip -c vl_gemm_park.hip -o build_prof/vl_gemm_park.hip.o
objs=$(ls build/*.o | grep -v vl_gemm_park)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/variants/libprof.so $objs build_prof/vl_gemm_park.hip.o
ls -la ../../tools/bin/variants/libprof.so
