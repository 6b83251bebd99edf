#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/train
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -25
