#!/bin/bash
# stand-alone measurement binaries (not part of the product): tools/bin/ travels to the GPU box with the snapshot
set -e
cd "$(dirname "$0")"
mkdir -p bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-implicit-const-int-float-conversion mfma_power_probe.hip -o bin/mfma_power_probe
