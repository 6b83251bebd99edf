#!/bin/bash
# round 3, visit A: new GPU tests + A/B of the paired backward launches.  gpurun --timeout 1500 -- bash tools/gpu_r3a.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3a; mkdir -p $OUT; cd $ROOT
for f in 0 4096 65536 131072; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -14; done
for f in 0 4096; do timeout 200 python tools/train_layer_times.py --dtype f32 --plan-flags $f > $OUT/f32_$f.txt 2>&1; grep -E "plan flags|family" $OUT/f32_$f.txt | head -12; done
timeout 1100 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q --timeout 900 --durations=12 -k "batch32 or masked or forward_graph or layer_local_parity_full_size or dropin" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest.log
