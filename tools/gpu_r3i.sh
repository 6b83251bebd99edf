#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3i; mkdir -p $OUT; cd $ROOT
timeout 120 scratch/dwtrain/dwtrain | cut -c1-330
for f in 8388608 0; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -4; done
