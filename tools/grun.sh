#!/bin/bash
# Local helper: rebuild the in-tree libraries if stale, then hand the command to gpurun.  tools/grun.sh <timeout> <log> -- <command>
T=$1; LOG=$2; shift 3
python fast-depth_amd/build.py > /dev/null || exit 1
/usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
