#!/bin/bash
# Round 4: rocprofv3 kernel statistics of the C3 step with the LayerNorm folding on and off (3 steps each, same box)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in on off; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lnf_$v -o lnf -- \
    python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --ln-fold $v > $GRAFT_REPO_ROOT/gpurun_out/rocprof_lnf_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  tail -1 gpurun_out/rocprof_lnf_$v.log | cut -c1-300
  find gpurun_out/prof_lnf_$v -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof_lnf_$v -name "*kernel_stats*.csv" | head -1)
  cp $f gpurun_out/lnf_${v}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.reader(open("$f")))[1:]
tot=sum(float(r[2]) for r in rows)
print("fold $v: total kernel time %.1f ms over %d kernels" % (tot/1e6, len(rows)))
for r in rows[:14]: print("  %-70s %6s %9.1f us %6s%%" % (r[0][:70], r[1], float(r[3])/1000, r[4]))
PY
done
