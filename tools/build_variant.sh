#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2=.." [source.hip ...]   (default source: vl_gemm_park.hip)
# A/B variant of the library: the named sources compiled with extra -D flags, everything else from the product build
# -> tools/bin/variants/libNAME.so.  tools/lib_ab.sh runs a command once per variant with that library in the in-tree
# library's place (the product has no library override and no run-time knobs).
set -e
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-vl_gemm_park.hip}
cd "$(dirname "$0")/../vit-lens_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/bin/variants build_var_$NAME
objs=$(ls build/*.o)
for src in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result $FLAGS -x hip -c $src -o build_var_$NAME/$src.o
  objs=$(echo "$objs" | grep -v "build/$src.o"); objs="$objs build_var_$NAME/$src.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/variants/lib$NAME.so $objs
rm -rf build_var_$NAME
ls -la ../../tools/bin/variants/lib$NAME.so
