#!/bin/bash
# A/B of library variants on the train step: gpurun -- bash tools/gpu_ab_libs.sh <dtype> <reps> <lib|product>...   (family sums + step time per variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; DT=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do for v in "$@"; do
  lib=; [ $v != product ] && lib="--lib $v"
  echo "$v | $(timeout 200 python tools/train_layer_times.py --summary --dtype $DT $lib 2>&1 | grep -E "plan flags|fd_pw_gemm_train|total" | tr '\n' ' ')"
done; done
