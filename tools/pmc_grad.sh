#!/bin/bash
# HBM traffic of the gradient merge kernel (FETCH_SIZE / WRITE_SIZE in separate passes): bash tools/pmc_grad.sh <tag>
set -o pipefail
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_grad_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/bench_extra.py grad > $OUT/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/bench_extra.py grad > $OUT/w.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,json
from collections import defaultdict
out={}
for sub,name in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    f=glob.glob("$OUT/"+sub+"/**/*counter_collection.csv",recursive=True)[0]
    vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_ck_overlapg" in r["Kernel_Name"] and r["Counter_Name"]==name]
    out[name+"_raw_per_launch"]=sum(vals)/len(vals); out[name+"_launches"]=len(vals)
# units as in the guide's HBM section: KiB? -> the project's convention (profiles/pmc_traffic.json): raw counter = bytes/ (see note)
json.dump(out,open("$OUT/summary.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
