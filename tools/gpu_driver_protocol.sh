#!/bin/bash
# The driver's own measurement, run by the builder: exactly `python3 bench.py --gpus 1 --steps 20 --warmup 5`, N times back to back
# (fresh process each, like the driver's single run).  gpurun -- bash tools/gpu_driver_protocol.sh [tag] [runs]
TAG=${1:-r05}; RUNS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for i in $(seq 1 $RUNS); do
  t0=$(date +%s.%N)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_protocol_$i.json 2> $OUT/bench_driver_protocol_$i.err; rc=$?
  t1=$(date +%s.%N)
  python3 - $OUT/bench_driver_protocol_$i.json $rc $t0 $t1 <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("driver protocol run: rc=%s wall %.1f s | %.1f frames/s, %.4f ms/step | windows %s | host enqueue %s ms/step | bf16 train %s ms" % (
        sys.argv[2], float(sys.argv[4]) - float(sys.argv[3]), d["value"], d["ms_per_step"], d.get("windows_ms_per_step"), d.get("host_enqueue_ms_per_step"),
        (d.get("train_step_bf16") or {}).get("ms_per_step")))
except Exception as e:
    print("driver protocol run: rc=%s, unreadable line (%s)" % (sys.argv[2], e))
P
done
