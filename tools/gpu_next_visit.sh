#!/bin/bash
# The GPU visit that round 4 could not make (its last minutes went into an xdist run of the suite): in order of what decides most.
#   bash tools/gpu_next_visit.sh            everything below, ~15 minutes
#   bash tools/gpu_next_visit.sh quick      steps 1-3 only, ~6 minutes
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== 1. the experimental one-wave-per-SIMD GEMM (cfg 14): correct? faster? =="
timeout 300 python tools/pk4_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_pk4_probe.log
if grep -q "ALL OK" gpurun_out/r05_pk4_probe.log; then
  echo "== 1b. the C3 step with it (VL_GEMM_PK4=1) against the default, interleaved =="
  for rep in 1 2; do for v in 0 1; do
    echo "VL_GEMM_PK4=$v rep $rep: $(VL_GEMM_PK4=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"achieved\"], d[\"roofline\"][\"all_gemm_tflops\"])")" | tee -a gpurun_out/r05_pk4_step_ab.log
  done; done
fi
echo "== 2. whole suite, sequential, default switches =="
bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05_pytest_gpu_default.log
echo "== 3. whole suite with the LayerNorm folding on (flip the default if this is green) =="
VL_LN_FOLD=1 bash tools/run_tests.sh
cp gpurun_out/pytest_gpu_full.log gpurun_out/r05_pytest_gpu_lnfold.log
[ "$1" = "quick" ] && exit 0
echo "== 4. smoke =="
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== 5. watts and MHz =="
bash tools/gpu_power_trace.sh 2>&1 | tail -8
echo "== 6. evidence on the final tree (bench line, rocprofv3, SQ counters incl. the folding kernels, HBM traffic, C4 / C5) =="
bash tools/gpu_evidence_r04.sh
