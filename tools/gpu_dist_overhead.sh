#!/bin/bash
# What the data-parallel machinery costs on ONE rank: kernel-trace statistics of the bf16 train step, single-GPU vs RCCL path forced on one rank.
# gpurun --timeout 900 -- bash tools/gpu_dist_overhead.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/dist_overhead; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in single forced; do
  if [ $mode = forced ]; then export FD_BENCH_FORCE_DIST=1; else unset FD_BENCH_FORCE_DIST; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o t -- python $ROOT/bench.py --only train_bf16 --steps 20 --warmup 3 > $OUT/$mode.json 2> $OUT/$mode.err
  find $OUT/$mode -name "*trace.csv" -delete
  echo "== $mode: $(tail -1 $OUT/$mode.json)"
  python3 - "$OUT/$mode" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("   kernels total %.3f ms over 23 steps = %.4f ms/step" % (tot / 1e6, tot / 1e6 / 23))
for r in rows:
    if not r["Name"].lstrip("void ").startswith("fd_"): print("   non-fd kernel: %-60s calls %5s avg %8.1f ns total %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6))
PY
done
