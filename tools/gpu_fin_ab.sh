#!/bin/bash
# A/B of the in-kernel BatchNorm finalisations (FD_TUNE_NO_CONSUMER_FINALIZE = every finalisation its own launch), both train dtypes, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x > gpurun_out/fin_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/fin_pytest.log; fi
for rep in 1 2; do
for dt in bf16 f32; do
  for fl in 0 NO_CONSUMER_FINALIZE; do
    timeout 200 python tools/train_layer_times.py --summary --dtype $dt --plan-flags $fl | grep -E "plan flags|fd_bn_|fd_dwconv_train|fd_dw_bwd|total"
  done
done
done
timeout 200 python tools/train_layer_times.py --dtype bf16 > gpurun_out/lt_train_bf16.txt 2>&1
timeout 200 python tools/train_layer_times.py --dtype f32 > gpurun_out/lt_train_f32.txt 2>&1
