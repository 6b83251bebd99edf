#!/bin/bash
# One rank: the bf16 / fp32 train step single-GPU vs with the data-parallel machinery forced on (RCCL route of the library), alternating, 50 steps each.
# gpurun --timeout 900 -- bash tools/gpu_dist_ab.sh [out file]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUTF=${1:-$ROOT/gpurun_out/dist_overhead.txt}; cd $ROOT; : > $OUTF
for rep in 1 2; do for mode in single forced forced_without_the_nccl_calls; do for cfg in train_bf16 train_f32; do
  if [ $mode = forced ]; then export FD_BENCH_FORCE_DIST=1; elif [ $mode = single ]; then unset FD_BENCH_FORCE_DIST; else export FD_BENCH_FORCE_DIST=2; fi
  echo "$rep $mode $cfg $(timeout 200 python bench.py --only $cfg --steps 50 --warmup 5 2> /dev/null | tail -1)" >> $OUTF
done; done; done
unset FD_BENCH_FORCE_DIST; cat $OUTF
