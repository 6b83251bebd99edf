#!/bin/bash
# quick perf/parity iteration on the GPU box: gpurun --timeout 600 -- bash tools/gpu_quick.sh [tag] [extra layer_times args]
TAG=${1:-quick}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/layer_times.py "$@" 2>&1 | tee $OUT/layer_times.txt | grep -v amdgpu.ids
