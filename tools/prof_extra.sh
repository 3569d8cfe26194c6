#!/bin/bash
# kernel trace of one tools/bench_extra.py section: bash tools/prof_extra.sh <section> <tag>
set -o pipefail
SEC=${1:-ms}; TAG=${2:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${SEC}_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_extra.py $SEC > $OUT/run.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
