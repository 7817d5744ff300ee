# usage: bash tools/run_tests.sh [pytest -k expr]   -> gpurun_out/pytest_gpu.log (+ _full.log)
# SEQUENTIAL by default, as the driver runs it: 327 tests took 228 s that way at the end of round 3 (GPUTEST_r03.json).  With
# pytest-xdist the same suite is an order of magnitude SLOWER on one GPU (every worker builds its own full-size weights and
# activation buffers, the workers' kernels serialise on the device and their allocations evict each other): -n 4 took ~25
# minutes, -n 6 did not finish in 38 (round 4 lost its last GPU minutes to that).  NPROC=k opts back in for CPU-heavy subsets.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
XD=""
[ -n "$NPROC" ] && [ "$NPROC" != "1" ] && XD="-n $NPROC"
timeout 2400 python -m pytest tests -m gpu -q $XD -p no:cacheprovider ${1:+-k "$1"} > gpurun_out/pytest_gpu_full.log 2>&1
tail -60 gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log
grep -E "^E   .*(Error|assert)|^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_full.log | cut -c1-700 | tail -40
