#!/bin/bash
# per-layer inference times: gpurun --timeout 600 -- bash tools/gpu_lt.sh "<layer_times args>" ["<args>" ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/lt; mkdir -p $OUT; cd $ROOT
i=0; for a in "$@"; do i=$((i+1)); timeout 200 python tools/layer_times.py $a > $OUT/lt_$i.txt 2>&1; echo "== $a"; grep -v amdgpu.ids $OUT/lt_$i.txt; done
