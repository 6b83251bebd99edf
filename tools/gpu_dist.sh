#!/bin/bash
# GPU-box visit: the multi-rank code paths of bench.py on whatever the box has.  gpurun --timeout 900 -- bash tools/gpu_dist.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/dist; mkdir -p $OUT; cd $ROOT
python -c "import torch; print('gpus visible:', torch.cuda.device_count())"
echo "== python bench.py --gpus 2 (self-launch)"; timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --train-steps 5 > $OUT/n2.json 2> $OUT/n2.err; echo "rc=$?"; tail -2 $OUT/n2.err; head -c 600 $OUT/n2.json; echo
echo "== FD_BENCH_FORCE_DIST=1 python bench.py (RCCL path on one rank)"; FD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 3 --train-steps 5 --no-cpu-baseline --extra-steps 0 > $OUT/force.json 2> $OUT/force.err; echo "rc=$?"; tail -2 $OUT/force.err
python - <<PY
import json
d = json.loads(open("$OUT/force.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step")}, d["config"]["rccl_ranks"])
for t in ("train_step", "train_step_bf16"):
    print(t, {k: d[t].get(k) for k in ("value", "ms_per_step", "parallelism", "allreduce", "error")})
PY
echo "== torch.distributed.run --nproc-per-node 1 (the driver's launch form)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --train-steps 5 --no-cpu-baseline --extra-steps 0 > $OUT/tr1.json 2> $OUT/tr1.err; echo "rc=$?"; tail -2 $OUT/tr1.err; head -c 300 $OUT/tr1.json; echo
