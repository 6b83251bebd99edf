#!/bin/bash
# GPU-box visit: only the gpu-marked tests (optionally a -k filter).  gpurun --timeout 1800 -- bash tools/gpu_tests.sh [tag] [-k expr]
TAG=${1:-tests}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 "$@" > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest_gpu.log
