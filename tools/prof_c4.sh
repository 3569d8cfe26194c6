#!/bin/bash
# kernel trace + MFMA counters (separate passes) of one full-size C4 call: bash tools/prof_c4.sh <tag> [W]
set -o pipefail
TAG=${1:-x}; W=${2:-10000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c4_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/c4_run.py $W > $OUT/run.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/c4_run.py $W > $OUT/pmc.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,json
from collections import defaultdict
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
st={r["Name"].split("(")[0].replace("void ","").split("<")[0]: r for r in csv.DictReader(open(f))}
for k,r in list(st.items())[:8]: print(k[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
f=glob.glob("$OUT/pmc/**/*counter_collection.csv",recursive=True)[0]
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for r in csv.DictReader(open(f)):
    if "k_ms_chain16" in r["Kernel_Name"]: acc["k_ms_chain16"][r["Counter_Name"]]+=float(r["Counter_Value"])
c=dict(acc["k_ms_chain16"])
# the chain launches of consecutive g-ordinates run on two streams and overlap: time = union of their intervals
f=glob.glob("$OUT/trace/**/*kernel_trace.csv",recursive=True)[0]
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "k_ms_chain16" in r["Kernel_Name"])
tot_ns=0.0; cs,ce=iv[0]
for a,b in iv[1:]:
    if a>ce: tot_ns+=ce-cs; cs,ce=a,b
    else: ce=max(ce,b)
tot_ns+=ce-cs
flops=c["SQ_INSTS_VALU_MFMA_MOPS_F64"]*512.0        # MOPS counts 512-flop units: one 16x16x4 f64 MFMA = 4
out={"counters_summed_over_launches":c,"chain_kernels_union_ms_trace_pass":tot_ns/1e6,"chain_kernels_sum_ms":float(st["ansfm::k_ms_chain16"]["TotalDurationNs"])/1e6,"mfma_f64_flops":flops,
     "TFLOPs":flops/(tot_ns*1e-9)/1e12,"frac_of_78.6":flops/(tot_ns*1e-9)/78.6e12,
     "mfma_busy_over_simd_cycles":c["SQ_VALU_MFMA_BUSY_CYCLES"]/(4.0*c["SQ_BUSY_CYCLES"]) if c.get("SQ_BUSY_CYCLES") else None}
json.dump(out,open("$OUT/summary.json","w"),indent=1); print(json.dumps(out,indent=1))
PY
