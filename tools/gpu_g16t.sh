#!/bin/bash
# fp32 train step: forward pointwise GEMMs on fd_pw_gemm16_f32<..., TRAIN> (default) vs the 32x32x2 kernel (NO_GEMM16), + the parity tests that cover it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
mkdir -p gpurun_out
for rep in 1 2; do for fl in 0 NO_GEMM16; do
  timeout 200 python tools/train_layer_times.py --summary --dtype f32 --plan-flags $fl | grep -E "plan flags|fd_pw_gemm|fd_bn_finalize|fd_dwconv_train|total"
done; done
timeout 200 python tools/train_layer_times.py --dtype f32 > gpurun_out/lt_train_f32.txt 2>&1
if [ "$1" = "tests" ]; then
  timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "in_kernel or (batch32 and dtype0) or (full_size and dtype0)" > gpurun_out/g16t_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/g16t_pytest.log | tail -5
fi
