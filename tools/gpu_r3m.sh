#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3m; mkdir -p $OUT; cd $ROOT
for f in 0 33554432; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family|step" $OUT/bf16_$f.txt | head -8; done
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -4
