#!/bin/bash
# usage: tools/lib_ab.sh REPS "variantA variantB ..." -- command ...
# Same-box A/B of library variants (tools/build_variant.sh; "product" = the in-tree library itself): REPS rounds, in every
# round the command runs once per variant with that variant in the in-tree library's place; the original is restored at the end.
REPS=$1; VARS=$2; shift 3
LIBSO=vit-lens_amd/vitlens_hip/libvitlens_hip.so
cp $LIBSO /tmp/lib_product.so
for r in $(seq 1 $REPS); do
  for v in $VARS; do
    if [ "$v" = product ]; then cp /tmp/lib_product.so $LIBSO; else cp tools/bin/variants/lib$v.so $LIBSO; fi
    echo "== round $r variant $v"
    "$@"
  done
done
cp /tmp/lib_product.so $LIBSO
