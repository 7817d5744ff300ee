#!/bin/bash
# Round-4 evidence in one GPU-box visit (final tree): the driver's bench line (C3 train step, with its CPU-baseline leg),
# rocprofv3 kernel stats of the same command, SQ counter passes over the GEMMs that carry the step, HBM traffic of the dominant
# GEMMs from two separate --pmc passes, the k-loop / epilogue cycle split (measurement build), attention timings + phase
# timeline, and the C4 / C5 lines.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench (driver default = C3) =="
timeout 900 python bench.py --detail gpurun_out/r04b_bench_c3_detail.json > gpurun_out/r04b_bench_c3.log 2>&1
tail -1 gpurun_out/r04b_bench_c3.log | cut -c1-2500
echo "== rocprof stats of the same command =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04b_prof_c3 -o r04b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r04b_rocprof_c3.log 2>&1
cd $R
find gpurun_out/r04b_prof_c3 -name "*kernel_trace*" -delete
f=$(find gpurun_out/r04b_prof_c3 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r04b_bench_c3_kernel_stats.csv && head -14 $f | cut -c1-200
echo "== GEMM probe + SQ counters =="
CASES=proj:res_bf16,proj:res_stats,fc:gelu,fc:ln_gelu,fc:gelu+dsave,fc:ln_dsave,dproj:dgelu_saved,qkv:bf16,qkv:ln_bf16,out:res_bf16,out:res_stats,dfc:bf16,sq8k:bf16
KB_TAG=r04b KB_CFGS=8 KB_ROUNDS=9 KB_CASES=$CASES timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_gemm_probe.log
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/r04b_pmc_$tag
  KB_CFGS=8 KB_ROUNDS=2 KB_CASES=proj:res_bf16,proj:res_stats,fc:gelu,fc:ln_gelu,fc:gelu+dsave,fc:ln_dsave,dproj:dgelu_saved,qkv:bf16,qkv:ln_bf16 timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r04b_pmc_$tag -o g -- python $R/tools/gemm_probe.py > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r04b_gemm_pmc_summary.txt
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r04b_pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt_pk" not in k: continue
        k = k[k.index("gemm_nt_pk"):]
        k = k[:k.index(">") + 1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    waves = m.get("SQ_WAVES", 2048.0)
    cyc = 4.0 * m["SQ_WAVE_CYCLES"] / waves
    out[k] = {"waves": round(waves), "kernel_cycles": round(cyc), "mfma_busy_frac": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
              "wait_any_frac": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4), "valu_insts": round(m["SQ_INSTS_VALU"]),
              "lds_insts": round(m["SQ_INSTS_LDS"]), "vmem_insts": round(m["SQ_INSTS_VMEM"]), "launches": len(d["SQ_WAVE_CYCLES"])}
    print(k, out[k])
json.dump({"kernels": out, "source": "tools/gpu_evidence_r04.sh: two rocprofv3 --pmc passes over tools/gemm_probe.py (cfg 8, M = 65536)"}, open("gpurun_out/r04b_gemm_pmc.json", "w"), indent=1)
PY
for d in gpurun_out/r04b_pmc_*; do for f in $(find $d -name "*counter_collection.csv"); do cp $f gpurun_out/$(basename $d)_counters.csv; done; done
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
echo "== k-loop / epilogue cycles (measurement build) =="
bash tools/gpu_gemm_phase_prof.sh
echo "== attention =="
(timeout 120 python tools/attn_probe.py; L=256 timeout 120 python tools/attn_probe.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_attn_probe.log
(./tools/bin/attn_phase_prof 257; ./tools/bin/attn_phase_prof 256) 2>&1 | tee gpurun_out/r04b_attn_phase_timeline.log
if [ "$1" != "quick" ]; then
echo "== hbm traffic (PMC, separate passes) =="
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python tools/traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r04b_hbm_traffic_c3.json \
  "gemm_nt_pk_kernel<3, 20,|65792,1024,4096|hi|3,0" "gemm_nt_pk_kernel<0, 11,|65792,4096,1024|all|0,1" "gemm_nt_pk_kernel<0, 10,|65792,3072,1024|all|0,0" \
  "gemm_nt_pk_kernel<0, 14,|65792,4096,1024|all|0,4" "gemm_nt_pk_kernel<6, 4,|65792,4096,1024|all|6,4" "attn_bwd_fused_kernel|256,16,257,64|all"
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +4M -delete
for WL in c4 c5; do
  echo "== bench $WL =="
  timeout 1200 python bench.py --workload $WL --steps 5 --warmup 2 > gpurun_out/r04b_bench_$WL.log 2>&1
  tail -1 gpurun_out/r04b_bench_$WL.log | cut -c1-1500
done
fi
