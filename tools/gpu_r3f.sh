#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT; cd $ROOT
for fl in 0 1048576 2097152 3145728 4194304; do
  timeout 200 python tools/layer_times.py --dtype f16 --plan-flags $fl > $OUT/lt_$fl.txt 2>&1; echo "== abl flags $fl"; grep -E "conv(6|7|11)\.3" $OUT/lt_$fl.txt | cut -c1-75
done
