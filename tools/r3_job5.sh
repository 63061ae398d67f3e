#!/bin/bash
# round-3 GPU job 5: BD tile traces (slab look-ahead 7 vs 3 chunks)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j5
mkdir -p $O
cd $R
for v in trace ah3; do
  DVITS_TRACE_LIB=$R/diff-vits_amd/libdvits_hip_$v.so timeout 300 python tools/gemm_trace.py conv:8x256x384x384x3 conv:8x1024x128x128x3 conv:8x128x1024x512x3 conv:8x512x256x256x3 > $O/trace_$v.txt 2>&1
  echo "== $v"; grep -E "^M=|k-loop  |issue prologue|first tile|whole workgroup|epilogue|k-split|sums" $O/trace_$v.txt
done
