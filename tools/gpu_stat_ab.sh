#!/bin/bash
# Round 5: the statistics-row train step.  [tests] + alternating A/B of the in-consumer finalisations against one launch per finalisation
# (NO_CONSUMER_FINALIZE) and with the depthwise backward kernels finalising their own unit (DW_BWD_FINALIZE); family sums, launch counts, step time.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/stat_ab; mkdir -p $OUT
if [ "$1" = "tests" ]; then timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 900 > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_train.log | grep -E "passed|failed|Error|assert" ; fi
for rep in 1 2; do
for dt in bf16 f32; do
  for fl in 0 NO_CONSUMER_FINALIZE DW_BWD_FINALIZE; do
    timeout 200 python tools/train_layer_times.py --summary --dtype $dt --plan-flags $fl 2>&1 | grep -E "plan flags|fd_bn_|total|Error|error" | tr '\n' ' '; echo
  done
done
done
timeout 200 python tools/train_layer_times.py --dtype bf16 > $OUT/lt_train_bf16.txt 2>&1
timeout 200 python tools/train_layer_times.py --dtype f32 > $OUT/lt_train_f32.txt 2>&1
timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags NO_CONSUMER_FINALIZE > $OUT/lt_train_bf16_sep.txt 2>&1
grep -E "family" $OUT/lt_train_bf16.txt
