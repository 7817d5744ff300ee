set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attn_probe.py 2>&1 | tail -2
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_SALU"; do
  tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
  N=2 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/attn_pmc_$tag -o a -- python $GRAFT_REPO_ROOT/tools/attn_probe.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/attn_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
