#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/train_prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/tools/train_prof.py 32 ${1:-f32} > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("void fd_") or r["Name"].startswith("fd_"))
print("total fd kernel time per step: %.3f ms" % (tot/8/1e6))
for r in rows[:34]:
    print("%-70s calls %5s avg %9.1f us  total/step %8.3f ms  %5.1f%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/8/1e6, float(r["Percentage"])))
PY
