#!/bin/bash
# two SQ counter passes of bench.py's C2 forward model: bash tools/pmc_quick.sh <tag> [bench args]
set -o pipefail
TAG=${1:-q}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-jacobian --no-extras $@"
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,json
from collections import defaultdict
out={}
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    acc=defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "k_ck_overlap" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in acc.items(): out.setdefault(k,{}).update({c: sum(v)/len(v) for c,v in cs.items()})
json.dump(out, open("$OUT/summary.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
