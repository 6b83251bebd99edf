#!/bin/bash
# re-run only the two PMC passes + summary of a tools/gpu_round.sh visit: gpurun -- bash tools/gpu_pmc_only.sh <tag>
TAG=${1:-r01}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 --train-steps 0 --extra-steps 0 > /dev/null 2> $OUT/pmc_fetch.err; echo "rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 --train-steps 0 --extra-steps 0 > /dev/null 2> $OUT/pmc_write.err; echo "rc=$?"
find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv" | head
