#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/sk
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "^E|passed|failed" | head -20
echo "== stream-K"; timeout 300 python tools/layer_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sk/layer_times_sk.txt | grep -E "untimed|pw_gemm|sum of" | cut -c1-230
echo "== plain"; timeout 300 python tools/layer_times.py --plan-flags 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sk/layer_times_plain.txt | grep -E "untimed|sum of"
