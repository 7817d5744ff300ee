#!/bin/bash
# Round 6, visit 8: ablation of the 8-wave kernel's k-loop (timing only) + the two tests fixed after visit 7.
set +e
mkdir -p gpurun_out
bash tools/lib_ab.sh 2 "product pknodma pknobar pknoread pknothing" -- python tools/pk_ablation.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_pk_ablation.log
timeout 900 python -m pytest "tests/test_hip_fullsize_steps.py::test_c4_audio_step_vitl_vs_oracle_autograd" "tests/test_hip_fullsize_steps.py::test_c5_pc_step_vitl_vs_oracle_autograd" "tests/test_hip_train.py" -q -p no:cacheprovider 2>&1 | tail -4
