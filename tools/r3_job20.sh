#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j20
mkdir -p $O
cd $R
timeout 300 python tools/gemm_trace.py > $O/gemm_trace.txt 2>&1; echo "gemm_trace rc=$?"
timeout 900 python tools/gemm_trace_fwd.py > $O/gemm_trace_fwd.txt 2>&1; echo "fwd rc=$?"
tail -2 $O/gemm_trace_fwd.txt | cut -c1-200
tools/micro/store_pattern > $O/store_pattern.txt 2>&1
