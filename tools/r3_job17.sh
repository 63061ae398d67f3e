#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j17
mkdir -p $O
cd $R
timeout 900 python tools/gemm_trace_fwd.py > $O/gemm_trace_fwd.txt 2>&1
tail -100 $O/gemm_trace_fwd.txt | cut -c1-260
