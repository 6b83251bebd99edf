#!/bin/bash
# GPU-box visit: the gemm16 microbenchmark (tools/microbench/gemm16.hip, prebuilt in scratch/gemm16).  gpurun -- bash tools/gpu_gemm16.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/gemm16; mkdir -p $OUT
timeout 300 $ROOT/scratch/gemm16/gemm16 ${GEMM16_ARG:--1} > $OUT/gemm16.txt 2>&1; echo "rc=$?"; tail -5 $OUT/gemm16.txt
