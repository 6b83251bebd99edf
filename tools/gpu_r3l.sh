#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3l; mkdir -p $OUT; cd $ROOT
for f in 0 33554432; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -5; done
paste <(grep -E "fd_dwconv_train<" $OUT/bf16_0.txt | awk '{print $2, $3}') <(grep -E "fd_dwconv_train<" $OUT/bf16_33554432.txt | awk '{print $3}')
