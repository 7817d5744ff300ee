# usage: bash tools/run_tests.sh [pytest -k expr]   -> gpurun_out/pytest_gpu.log
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider ${1:+-k "$1"} 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
grep -v "^E  " gpurun_out/pytest_gpu.log | tail -45 | cut -c1-220
