cd $GRAFT_REPO_ROOT; OUT=gpurun_out/stat_ab2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 900 -k "in_kernel_batchnorm or bit_reproducible or fused_step or layer_local_parity_batch32" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_sel.log | grep -E "passed|failed|Error|assert"
for rep in 1 2; do for dt in bf16 f32; do for fl in 0 NO_CONSUMER_FINALIZE; do
    timeout 200 python tools/train_layer_times.py --summary --dtype $dt --plan-flags $fl 2>&1 | grep -E "plan flags|fd_bn_|total|Error|error" | tr '\n' ' '; echo
done; done; done
timeout 200 python tools/train_layer_times.py --dtype bf16 > $OUT/lt_train_bf16.txt 2>&1
timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags NO_CONSUMER_FINALIZE > $OUT/lt_train_bf16_sep.txt 2>&1
grep -E "family" $OUT/lt_train_bf16.txt
