#!/bin/bash
# Round-2 evidence in one GPU-box visit (final tree): environment, micro-benchmarks, the driver's bench line (C3 train step),
# rocprofv3 kernel stats of the same command, HBM traffic of the dominant GEMM from two separate --pmc passes, the attention
# kernels' phase timeline and SQ counters.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocminfo ==" > gpurun_out/env.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/env.log
echo "== kernel bench =="
timeout 600 python tools/kernel_bench.py --quick > gpurun_out/kernel_bench.log 2>&1; tail -30 gpurun_out/kernel_bench.log
echo "== attention =="
(timeout 120 python tools/attn_probe.py; L=256 timeout 120 python tools/attn_probe.py; ./tools/bin/attn_phase_prof 257; ./tools/bin/attn_phase_prof 256) > gpurun_out/attn.log 2>&1; grep -v amdgpu.ids gpurun_out/attn.log
echo "== bench (driver default = C3) =="
timeout 900 python bench.py --detail gpurun_out/bench_c3_detail.json > gpurun_out/bench_c3.log 2>&1
tail -1 gpurun_out/bench_c3.log | cut -c1-2500
echo "== rocprof stats of the same command =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o r02 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_c3.log 2>&1
cd $R
find gpurun_out/prof_c3 -name "*kernel_trace*" -delete
for f in $(find gpurun_out/prof_c3 -name "*kernel_stats*.csv" | head -1); do head -12 $f | cut -c1-200; done
echo "== hbm traffic (PMC, separate passes) =="
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE "gemm_nt_pk_kernel<0, 3>" gpurun_out/hbm_traffic.json 65792,4096,1024
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +4M -delete
