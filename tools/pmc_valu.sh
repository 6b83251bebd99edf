#!/bin/bash
# VALU / LDS / wait accounting per kernel of one bench configuration.  gpurun --timeout 600 -- bash tools/pmc_valu.sh <cfg> [tag]
# (own rocprofv3 --pmc pass with kernel trace only; SQ_* ACTIVE / WAIT / WAVE counters tick in quad-cycles, MI355X_MICROARCH.md)
CFG=${1:-train_bf16}; TAG=${2:-$CFG}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${PMC_OUT:-$ROOT/gpurun_out}/pmc_valu_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o p -- python $ROOT/bench.py --only $CFG --steps 3 --warmup 2 > /dev/null 2> $OUT/a.err; echo rc=$?
python3 $ROOT/tools/summarize_valu_pmc.py $(find $OUT/a -name "*counter_collection.csv" | head -1) $(find $OUT/a -name "*kernel_trace.csv" | head -1) | tee $OUT/summary.txt | head -40
find $OUT/a -name "*.csv" -size +8M -delete
