#!/bin/bash
# GPU-box visit: train tests + bench (train part) + per-kernel stats of the bf16 train step.  gpurun -- bash tools/gpu_train_round.sh [tag]
TAG=${1:-train}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q --timeout 900 > $OUT/pytest_train.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_train.log
timeout 600 python bench.py --no-cpu-baseline --extra-steps 0 --steps 30 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -2 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | grep -A14 "^train_step"
cd /tmp && export TMPDIR=/tmp
for cfg in train_bf16 train_f32; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$cfg -o trace -- python $ROOT/bench.py --only $cfg --steps 20 --warmup 3 > $OUT/prof_$cfg.json 2> $OUT/prof_$cfg.err; echo "rc=$?"; cat $OUT/prof_$cfg.json
find $OUT/prof_$cfg -name "*kernel_trace.csv" -size +20M -delete
done
