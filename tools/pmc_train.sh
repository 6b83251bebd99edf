#!/bin/bash
# PMC counters for the train-step kernels: gpurun -- bash tools/pmc_train.sh [f32|bf16] [kernel-name filter]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_train; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
DT=${1:-bf16}; FILT=${2:-fd_dw}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
  tag=$(echo $set | cut -c1-20 | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $ROOT/tools/train_prof.py 32 $DT > /dev/null 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$FILT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k); print("   ", {c: round(max(v)) for c, v in d.items()})
PY
done
