#!/bin/bash
# Round-3 GEMM visit 3 (r03c: 32x32x16 main loop; r03d: 16x16x32 main loop = the shipped default): SQ counters of the persistent kernel on the shapes / epilogues that carry the C3 step after the
# gelu'-saving forward: c_proj + residual (K = 4096, the dominant launch), c_fc + GELU (+ gelu' save), dX through GELU with the
# saved derivative.  Two --pmc passes over tools/gemm_probe.py; -> gpurun_out/r03d_gemm_pmc.json
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASES=proj:res_bf16,fc:gelu,fc:gelu+dsave,dproj:dgelu_saved,qkv:bf16
echo "== timings =="
KB_CFGS=8 KB_CASES=$CASES timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d_gemm_probe.log
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  tag=$(echo $grp | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/r03d_pmc_$tag
  KB_CFGS=8 KB_ROUNDS=2 KB_CASES=$CASES timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r03d_pmc_$tag -o g -- python $R/tools/gemm_probe.py > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r03d_gemm_pmc_summary.txt
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r03d_pmc_*/**/*counter_collection.csv", recursive=True)):
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
json.dump(out, open("gpurun_out/r03d_gemm_pmc.json", "w"), indent=1)
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
