#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3g; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -8
timeout 200 python tools/train_layer_times.py --dtype bf16 > $OUT/bf16.txt 2>&1; grep -E "plan flags|family" $OUT/bf16.txt | head -8
timeout 200 python tools/train_layer_times.py --dtype f32 > $OUT/f32.txt 2>&1; grep -E "plan flags|family" $OUT/f32.txt | head -6
