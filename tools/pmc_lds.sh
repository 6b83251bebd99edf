cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/a -o p -- python $GRAFT_REPO_ROOT/bench.py --only ${PMC_CFG:-infer} --steps 3 --warmup 2 > /dev/null 2> $OUT/a.err; echo rc=$?
find $OUT -name "*counter_collection.csv" | head; ls -la $OUT/a/* | head
