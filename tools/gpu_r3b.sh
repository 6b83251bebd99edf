#!/bin/bash
# round 3, visit B: B=1 reproducibility diagnostic, hipGraph tests, CB16 A/B, forced-1-rank RCCL step (fp32 / bf16 exchange).  gpurun --timeout 1200 -- bash tools/gpu_r3b.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3b; mkdir -p $OUT; cd $ROOT
timeout 300 python tools/diag_b1.py > $OUT/diag_b1.txt 2>&1; grep -v amdgpu.ids $OUT/diag_b1.txt | tail -20
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "forward_graph" > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "^E|passed|failed" $OUT/pytest.log | head -12
for f in 0 262144; do timeout 200 python tools/train_layer_times.py --dtype bf16 --plan-flags $f > $OUT/bf16_$f.txt 2>&1; grep -E "plan flags|family" $OUT/bf16_$f.txt | head -8; done
for gx in f32 bf16; do
FD_BENCH_FORCE_DIST=1 timeout 400 python bench.py --steps 20 --warmup 3 --train-steps 20 --no-cpu-baseline --extra-steps 0 --profile-steps 0 --grad-exchange $gx > $OUT/force_$gx.json 2> $OUT/force_$gx.err; echo "rc=$?"; tail -2 $OUT/force_$gx.err
python - <<PY
import json
d = json.loads(open("$OUT/force_$gx.json").read().strip().splitlines()[-1])
for t in ("train_step", "train_step_bf16"):
    print("$gx", t, {k: d[t].get(k) for k in ("value", "ms_per_step", "parallelism", "allreduce", "error")})
PY
done
timeout 300 python bench.py --steps 20 --warmup 3 --train-steps 20 --no-cpu-baseline --extra-steps 0 --profile-steps 0 > $OUT/plain.json 2> $OUT/plain.err
python - <<PY
import json
d = json.loads(open("$OUT/plain.json").read().strip().splitlines()[-1])
print("plain", d["value"], d["ms_per_step"])
for t in ("train_step", "train_step_bf16"):
    print("plain", t, {k: d[t].get(k) for k in ("value", "ms_per_step", "error")})
PY
