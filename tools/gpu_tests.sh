#!/bin/bash
# GPU-box visit: parity tests only (logs -> gpurun_out/)
set +e
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider ${PYTEST_ARGS} ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -120 gpurun_out/pytest_gpu.log | cut -c1-400
