#!/bin/bash
# one rocprofv3 --pmc pass per argument group over tools/bench_extra.py grad: bash tools/pmc_any.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_any_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/bench_extra.py grad > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,json
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ck_overlapg" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:sum(v)/len(v) for k,v in acc.items()}
json.dump(out,open("$OUT/summary.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
