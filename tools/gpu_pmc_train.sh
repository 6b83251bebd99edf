#!/bin/bash
# HBM traffic (PMC) of the bf16 train step, per kernel family.  gpurun --timeout 600 -- bash tools/gpu_pmc_train.sh [cfg]
CFG=${1:-train_bf16}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_$CFG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -o p -- python $ROOT/bench.py --only $CFG --steps 3 --warmup 2 > /dev/null 2> $OUT/$ctr.err; echo "$ctr rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + ctr + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "fd_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
                acc[k][ctr][0] += float(r["Counter_Value"]); acc[k][ctr][1] += 1
steps = 6.0   # bench.py --only train_*: 3 warm-up + 3 timed steps
tot = 0.0
rows = []
for k, d in acc.items():
    b = (2.0 * d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024.0 / steps
    rows.append((b, k, d["FETCH_SIZE"][1] / steps)); tot += b
for b, k, n in sorted(rows, reverse=True): print("%-32s %5.1f launches/step %9.1f MB/step" % (k, n, b / 1e6))
print("total %.3f GB per step" % (tot / 1e9))
PY
