#!/bin/bash
# GPU-box visit: inference parity tests + per-layer times (fp32, fp16) + short bench.  gpurun -- bash tools/gpu_infer_round.sh [tag]
TAG=${1:-infer}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_checkpoints.py -m gpu -q --timeout 600 > $OUT/pytest_parity.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_parity.log
timeout 300 python tools/layer_times.py 2>&1 | grep -v amdgpu.ids | tee $OUT/layer_times_f32.txt | cut -c1-150
timeout 300 python tools/layer_times.py --dtype f16 2>&1 | grep -v amdgpu.ids | tee $OUT/layer_times_f16.txt | cut -c1-120 | head -12
