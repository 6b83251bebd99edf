#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3o; mkdir -p $OUT; cd $ROOT
for d in bf16 f32; do timeout 200 python tools/train_layer_times.py --dtype $d --plan-flags 0 > $OUT/$d.txt 2>&1; grep -E "plan flags|family|step" $OUT/$d.txt | head -9; done
timeout 200 python tools/layer_times.py --dtype f16 > $OUT/lt_f16.txt 2>&1; grep -E "untimed|sum of" $OUT/lt_f16.txt
timeout 200 python tools/layer_times.py --dtype f16 --pruned --batch 64 > $OUT/lt_f16p.txt 2>&1; grep -E "untimed|sum of" $OUT/lt_f16p.txt
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -n 4
