#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3j; mkdir -p $OUT; cd $ROOT
for f in 0 16777216; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -4; done
grep -E "dwconv_train" $OUT/bf16_0.txt | grep -v family | awk '{print $2, $3, $4}' | head -20
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 600 -k "layer_local or sibling" > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -8
