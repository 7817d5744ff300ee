#!/bin/bash
# rocprofv3 kernel-trace --stats of one bench workload: WL=c3|c4|c5 ARGS="..."
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
for WL in ${WLS:-c5}; do
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline ${ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof_$WL.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% n={r["Calls"]:>6} avg={float(r["AverageNs"])/1e3:9.1f}us  {r["Name"][:110]}')
PY
  find gpurun_out/prof_$WL -name "*kernel_trace*" -delete
done
