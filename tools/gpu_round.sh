#!/bin/bash
# One GPU-box visit: smoke, gpu tests, bench, rocprofv3 kernel trace.  Usage: gpurun --timeout 1500 -- bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -3 $OUT/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
    print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample"))
    for k in d["kernels"]: print(k)
    print(d["whole_step"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; echo "rc=$?"
find $OUT/prof -name "*stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
echo "== rocprofv3 PMC passes (HBM traffic)"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 --train-steps 0 --extra-steps 0 > /dev/null 2> $OUT/pmc_fetch.err; echo "rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 --train-steps 0 --extra-steps 0 > /dev/null 2> $OUT/pmc_write.err; echo "rc=$?"
cd $ROOT; python tools/summarize_profiles.py $OUT > $OUT/profile_summary.txt 2>&1; tail -40 $OUT/profile_summary.txt
du -sh $OUT
