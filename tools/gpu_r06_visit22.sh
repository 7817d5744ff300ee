#!/bin/bash
# Round 6, visit 22: the attention probe under the old and the new library (visit 21's run of it died on an environment
# variable clash), and the phase timeline of the new kernels.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
for SEQ in 257 256 197; do
  export L=$SEQ N=20
  bash tools/lib_ab.sh 2 "attnbwd_before product" -- python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/L=$SEQ  /" | tee -a gpurun_out/r06_v22_attn_probe_ab.log
done
unset L N
[ -x tools/bin/attn_phase_prof ] && (./tools/bin/attn_phase_prof 257; ./tools/bin/attn_phase_prof 256) 2>&1 | tee gpurun_out/r06_v22_attn_phase_timeline.log
