#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_job5; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "column_split or vs_golden" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python tools/profile_ops.py > $O/ops_split.txt 2>&1; grep "xattn" $O/ops_split.txt; head -3 $O/ops_split.txt | tail -2
bash tools/ab_bench.sh r06_qkv_xa 3 xa=- noxa=DVITS_QKV_XA=0 2>&1 | tail -3
