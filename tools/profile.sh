#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes.
# usage: bash tools_profile.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-jacobian --no-extras"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -40
