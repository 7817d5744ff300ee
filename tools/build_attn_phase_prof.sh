# builds tools/bin/attn_phase_prof (gfx950): the attention kernels with their phase stamps enabled + the driver
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -Wno-unused-result -DVL_ATTN_PROF -Iinclude -Ivit-lens_amd/csrc \
  tools/attn_phase_prof.hip vit-lens_amd/csrc/vl_attn.hip vit-lens_amd/csrc/vl_attn_bwd.hip vit-lens_amd/csrc/vl_attn_bwd_fused.hip -o tools/bin/attn_phase_prof
