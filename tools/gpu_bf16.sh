#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/bf16
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "layer_local" 2>&1 | tail -15
timeout 600 python scratch/bf16_e2e.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bf16/e2e.txt
