#!/bin/bash
# One GPU-box visit: smoke, gpu tests, bench, then ONE rocprofv3 invocation per configuration (kernel trace + stats) and the PMC passes.
# Usage: gpurun --timeout 2400 -- bash tools/gpu_round.sh [tag] [skip-tests]
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "rc=$?"; tail -3 $OUT/smoke.log
if [ -z "$2" ]; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -8 $OUT/pytest_gpu.log; fi
echo "== bench: the driver's protocol (bench.py --gpus 1 --steps 20 --warmup 5), three fresh processes; the first is the archived bench line"
bash tools/gpu_driver_protocol.sh $TAG 3
cp $OUT/bench_driver_protocol_1.json $OUT/bench.json
python tools/show_bench.py $OUT/bench.json
echo "== bf16 vs fp32 plan gradients at B = 32 (tools/bf16_grad_probe.py)"; timeout 300 python tools/bf16_grad_probe.py 32 2>&1 | tail -5 | tee $OUT/bf16_grad_probe.txt
cd /tmp && export TMPDIR=/tmp
for cfg in infer train_f32 train_bf16 f16 bf16 pruned_f16; do
  echo "== rocprofv3 kernel trace: $cfg"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$cfg -o trace -- python $ROOT/bench.py --only $cfg --steps 20 --warmup 3 > $OUT/prof_$cfg.json 2> $OUT/prof_$cfg.err; echo "rc=$?"
  find $OUT/prof_$cfg -name "*kernel_trace.csv" -size +20M -delete
done
echo "== rocprofv3 PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; MFMA busy)"
for cfg in infer train_bf16 train_f32 f16 bf16 pruned_f16; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${cfg}_$ctr -o p -- python $ROOT/bench.py --only $cfg --steps 3 --warmup 2 > /dev/null 2> $OUT/pmc_${cfg}_$ctr.err; echo "$cfg $ctr rc=$?"
  done
done
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_infer_SQ -o p -- python $ROOT/bench.py --only infer --steps 3 --warmup 2 > /dev/null 2> $OUT/pmc_infer_SQ.err; echo "infer SQ rc=$?"
echo "== VALU / LDS / wait accounting (SQ counters, own passes)"
for cfg in infer f16 train_bf16 train_f32; do PMC_OUT=$OUT bash $ROOT/tools/pmc_valu.sh $cfg > /dev/null 2>&1; echo "$cfg valu rc=$?"; done
echo "== data-parallel machinery on one rank (single vs forced RCCL path, bf16 and fp32 steps, alternating)"
cd $ROOT
for rep in 1; do for mode in single forced forced_without_the_nccl_calls; do for cfg in train_bf16 train_f32; do
  if [ $mode = forced ]; then export FD_BENCH_FORCE_DIST=1; elif [ $mode = single ]; then unset FD_BENCH_FORCE_DIST; else export FD_BENCH_FORCE_DIST=2; fi
  echo "$rep $mode $cfg $(timeout 200 python bench.py --only $cfg --steps 50 --warmup 5 2> /dev/null | tail -1)" >> $OUT/dist_overhead.txt
done; done; done
unset FD_BENCH_FORCE_DIST; cat $OUT/dist_overhead.txt
cd $ROOT; python tools/summarize_profiles.py $OUT > $OUT/profile_summary.txt 2>&1; head -60 $OUT/profile_summary.txt
du -sh $OUT
