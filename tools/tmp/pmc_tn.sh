set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in tn nt; do
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU"; do
    tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
    rm -rf $R/gpurun_out/tnpmc_${v}_$tag
    DW_ONLY=$v timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/tnpmc_${v}_$tag -o g -- python $R/tools/dw_probe.py > /dev/null 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in ("tn", "nt"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"gpurun_out/tnpmc_{v}_*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_tn_kernel" in k or "gemm_nt_pk" in k:
                agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(v, k, {c: round(sum(x) / len(x)) for c, x in d.items()})
PY
find gpurun_out -name "*counter_collection.csv" -size +1M -delete; find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
