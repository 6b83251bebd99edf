#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
for dt in f32 bf16; do echo "== $dt"; bash tools/gpu_train_prof.sh $dt | head -${1:-16}; done
