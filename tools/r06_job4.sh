#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_job4; mkdir -p $O; cd $R
python tools/profile_ops.py --batch 16 > $O/ops_b16_default.txt 2>&1
DVITS_CONV3=0 python tools/profile_ops.py --batch 16 > $O/ops_b16_ring.txt 2>&1
DVITS_GNX_ROUNDS=0 python tools/profile_ops.py --batch 16 --summary > $O/ops_b16_rounds0.txt 2>&1
DVITS_CONV3=0 DVITS_GNX_ROUNDS=0 python tools/profile_ops.py --batch 16 --summary > $O/ops_b16_ring_rounds0.txt 2>&1
head -8 $O/ops_b16_default.txt $O/ops_b16_ring.txt $O/ops_b16_rounds0.txt $O/ops_b16_ring_rounds0.txt
for v in "X=1" "DVITS_GNX_ROUNDS=0" "DVITS_CONV3=0" "DVITS_CONV3=0 DVITS_GNX_ROUNDS=0"; do
  echo "== $v"; ( export $v CWC_VARIANTS=default; python tools/conv_window_check.py 16,1024,256 2>&1 | grep "ms/forward" )
done
