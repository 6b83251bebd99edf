#!/bin/bash
# A/B of alternative library builds: copies each over the in-tree library (on the GPU box's scratch copy only) and times the bf16 train step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3k; mkdir -p $OUT; cd $ROOT
LIB=fast-depth_amd/fastdepth_hip/libfastdepth_hip.so; cp $LIB /tmp/orig.so
for v in orig "$@"; do
  if [ $v = orig ]; then cp /tmp/orig.so $LIB; else cp scratch/$v $LIB; fi
  timeout 200 python tools/train_layer_times.py --dtype bf16 > $OUT/bf16_$v.txt 2>&1; echo "== $v"; grep -E "plan flags|fd_pw_bwd|reduce_weights" $OUT/bf16_$v.txt | head -3
done
cp /tmp/orig.so $LIB
