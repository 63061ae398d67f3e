#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_job5; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "column_split or vs_golden or outside_the_round5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python tools/qkv_trace.py 2>&1 | grep -v amdgpu.ids > $O/qkv_trace_v5.txt; cat $O/qkv_trace_v5.txt
python tools/profile_ops.py > $O/ops_split.txt 2>&1; grep "wg / 64 rows" $O/ops_split.txt | grep -v GEGLU
bash tools/ab_bench.sh r06_qkv3 3 split=- chain=DVITS_QKV_SPLIT=0 2>&1 | tail -3
