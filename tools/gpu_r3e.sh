#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3e; mkdir -p $OUT; cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16bit_gemm16" > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -8
for cfg in "--dtype f16" "--dtype f16 --pruned --batch 64"; do
  tag=$(echo $cfg | tr -d ' -'); timeout 200 python tools/layer_times.py $cfg > $OUT/lt_$tag.txt 2>&1; echo "== $cfg"; grep -E "untimed|sum of" $OUT/lt_$tag.txt
  grep -E "gemm16" $OUT/lt_$tag.txt | cut -c1-110
done
