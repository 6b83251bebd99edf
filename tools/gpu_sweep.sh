#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for t in "16,16" "8,16" "8,32" "16,8" "4,32" "8,8" "16,32"; do
  echo "== FD_TUNE_DW_TILE=$t"; FD_TUNE_DW_TILE=$t timeout 120 python tools/layer_times.py --iters 10 2>&1 | grep -E "untimed|dwconv"
done
