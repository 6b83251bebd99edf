#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3n; mkdir -p $OUT; cd $ROOT
for d in bf16 f32; do timeout 200 python tools/train_layer_times.py --dtype $d --plan-flags 0 > $OUT/$d.txt 2>&1; grep -E "plan flags|family|step" $OUT/$d.txt | head -9; done
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -n 4
