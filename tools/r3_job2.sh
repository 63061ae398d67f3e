#!/bin/bash
# round-3 GPU job 2: where does the k-loop's time go?  trace builds with parts of the loop removed (DV_GEMM_EXP, gemm_tile.h)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j2
mkdir -p $O
cd $R
export DVITS_GEMM_PF=0
for e in trace exp1 exp2 exp3 exp4; do
  DVITS_TRACE_LIB=$R/diff-vits_amd/libdvits_hip_$e.so timeout 300 python tools/gemm_trace.py 2048x1152x384 8192x384x128 4096x256x2048 > $O/trace_$e.txt 2>&1
  echo "== $e"
  grep -E "^M=|k-loop  |k-loop sums" $O/trace_$e.txt
done
