#!/bin/bash
# round 3, visit C: the whole gpu test tier + smoke + the default bench line.  gpurun --timeout 2400 -- bash tools/gpu_r3c.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3c; mkdir -p $OUT; cd $ROOT
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json
