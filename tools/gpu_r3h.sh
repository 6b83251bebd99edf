#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3h; mkdir -p $OUT; cd $ROOT
for f in 0 8388608; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -5; done
timeout 200 python tools/train_layer_times.py --dtype f32 > $OUT/f32.txt 2>&1; grep -E "plan flags|family" $OUT/f32.txt | head -4
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -8
