#!/bin/bash
# Round 6, visit 14: why the packed-pair GELU changed results - the failing API test with its traceback, and checksums of every
# GELU-carrying epilogue under the scalar and the packed library on the same operands.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest "tests/test_hip_api.py::test_reference_training_sequence_through_api" -q -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/r06_v14_api_fail.log
bash tools/lib_ab.sh 1 "gelu_scalar product" -- python tools/epilogue_checksum.py 2>&1 | tee gpurun_out/r06_v14_checksums.log
