#!/bin/bash
# Round 6, the LAST visit - run from the build container (not on the GPU box):
#   1. the CPU suite HERE, before anything is uploaded (round 5 ended with three red CPU tests nobody had re-run); its log and count
#      go to profiles/r06_pytest_cpu_final.log;
#   2. only if that is green: the GPU evidence visit (tools/gpu_evidence_r06.sh) through gpurun.
# usage: bash tools/r06_final_check.sh [quick]
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
timeout 2400 python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4 | tee profiles/r06_pytest_cpu_final.log
grep -q " passed" profiles/r06_pytest_cpu_final.log && ! grep -q "failed\|error" profiles/r06_pytest_cpu_final.log || { echo "CPU suite not green: nothing uploaded"; exit 1; }
/usr/local/graft/bin/gpurun --timeout 4200 -- "bash tools/gpu_evidence_r06.sh $1"
