#!/bin/bash
# Round 5: full GPU test tier + driver-protocol bench + sweep of the statistics-row policy (library variants built with tools/build_variant.py).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/stat_sweep; mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -E "^smoke" $OUT/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | grep -E "passed|failed"; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
for rep in 1 2; do
  for v in product st_adds64 st_adds256 st_all4 st_all1 st_blk16; do
    lib=; [ $v != product ] && lib="--lib scratch/variants/$v.so"
    echo "$v $(timeout 200 python tools/train_layer_times.py --summary --dtype bf16 $lib 2>&1 | grep -E "plan flags|total" | tr '\n' ' ')"
  done
done
bash tools/gpu_driver_protocol.sh stat_sweep 2
