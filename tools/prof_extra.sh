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
# optional MFMA counters of the same section (own pass, counters only): bash tools/prof_extra.sh ms <tag> pmc
if [ "$3" = "pmc" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/bench_extra.py $SEC > $OUT/pmc.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$OUT/pmc/**/*counter_collection.csv",recursive=True)[0]
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if "ansfm" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items(): print(k, {c: sum(v)/len(v) for c,v in cs.items()})
PY
fi
# optional VALU / LDS issue counters of the same section (own pass): bash tools/prof_extra.sh lbl_c5 <tag> valu
if [ "$3" = "valu" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -- python3 tools/bench_extra.py $SEC > $OUT/pmc_valu.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob,json
from collections import defaultdict
f=glob.glob("$OUT/pmc_valu/**/*counter_collection.csv",recursive=True)[0]
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if "ansfm" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k: {c: sum(v)/len(v) for c,v in cs.items()} for k,cs in acc.items()}
json.dump(out, open("$OUT/pmc_valu_summary.json","w"), indent=1)
for k,v in out.items(): print(k, v)
PY
fi
# wave-state counters (own pass): bash tools/prof_extra.sh grad <tag> wait
if [ "$3" = "wait" ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_wait -- python3 tools/bench_extra.py $SEC > $OUT/pmc_wait.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob,json
from collections import defaultdict
f=glob.glob("$OUT/pmc_wait/**/*counter_collection.csv",recursive=True)[0]
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if "ansfm" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k: {c: sum(v)/len(v) for c,v in cs.items()} for k,cs in acc.items()}
json.dump(out, open("$OUT/pmc_wait_summary.json","w"), indent=1)
for k,v in out.items(): print(k, v)
PY
fi
