#!/bin/bash
# quick A/B of the bf16 (and fp32) train step: gpurun --timeout 400 -- bash tools/gpu_tr.sh [flags...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/tr; mkdir -p $OUT; cd $ROOT
for f in ${@:-0}; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -16; done
