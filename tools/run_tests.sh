# usage: bash tools/run_tests.sh [pytest -k expr]   -> gpurun_out/pytest_gpu.log (+ _full.log)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -n ${NPROC:-4} -p no:cacheprovider ${1:+-k "$1"} > gpurun_out/pytest_gpu_full.log 2>&1
tail -60 gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log
grep -E "^E   .*(Error|assert)|^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_full.log | cut -c1-700 | tail -40
