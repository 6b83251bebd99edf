#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_gemm; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCP|TCC|TA|TD|GRBM)_[A-Z0-9_]+" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  tag=$(echo $set | cut -c1-20 | tr ' ' '_')
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- $ROOT/scratch/gemm/gemm_bench 0 > /dev/null 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k); print("   ", {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
done
