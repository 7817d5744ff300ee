#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/attn_fwd_ab.log; : > $L
for rep in 1 2; do for v in old new; do
  cp tools/bin/variants/lib_attn_$v.so vit-lens_amd/vitlens_hip/libvitlens_hip.so
  echo "== $v rep $rep" >> $L
  N=30 python tools/attn_probe.py 2>&1 | grep "fwd" >> $L
  L=256 N=30 python tools/attn_probe.py 2>&1 | grep "fwd" >> $L
done; done
cp tools/bin/variants/lib_attn_new.so vit-lens_amd/vitlens_hip/libvitlens_hip.so
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "attention" 2>&1 | tail -3 >> $L
cat $L
