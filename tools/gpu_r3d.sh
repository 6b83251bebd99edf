#!/bin/bash
# round 3, visit D: 16-bit inference with fd_pw_gemm16_h16.  gpurun --timeout 900 -- bash tools/gpu_r3d.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3d; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16bit or fp16 or forward_graph" > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -12
for cfg in "--dtype f16" "--dtype bf16" "--dtype f16 --pruned --batch 64" "--dtype f16 --plan-flags 576" "--dtype f16 --pruned --batch 64 --plan-flags 576"; do
  tag=$(echo $cfg | tr -d ' -'); timeout 200 python tools/layer_times.py $cfg > $OUT/lt_$tag.txt 2>&1; echo "== $cfg"; grep -E "untimed|sum of" $OUT/lt_$tag.txt
done
grep -E "gemm16|conv1[23]\.|decode_conv[12]" $OUT/lt_dtypef16.txt | cut -c1-200
