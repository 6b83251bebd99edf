#!/bin/bash
# the GPU test tier + smoke on one box: gpurun --timeout 1500 -- bash tools/gpu_tests.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/tests; mkdir -p $OUT; cd $ROOT
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout 1400 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log
