#!/bin/bash
# rocprofv3 --pmc passes over any command, counters of one kernel averaged per launch:
#   bash tools/pmc_cmd.sh <tag> <kernel substring> "<python args>" "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
TAG=$1; KERN=$2; CMD=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cmd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $CMD > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,json
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{"per_launch":sum(v)/len(v),"launches":len(v)} for k,v in acc.items()}
json.dump(out,open("$OUT/summary.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
