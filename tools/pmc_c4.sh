#!/bin/bash
# wave-state counters of the scattering chain kernel on the full-size C4 call: bash tools/pmc_c4.sh <tag> [W]
set -o pipefail
TAG=${1:-x}; W=${2:-10000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c4_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 tools/c4_run.py $W > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 tools/c4_run.py $W > $OUT/b.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,json
from collections import defaultdict
out={}
for sub in ("a","b"):
    f=glob.glob("$OUT/"+sub+"/**/*counter_collection.csv",recursive=True)[0]
    acc=defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_ms_chain16" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    out.update(acc)
json.dump(out,open("$OUT/summary.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
