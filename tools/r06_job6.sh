#!/bin/bash
# round 6 evidence: rocprofv3 kernel stats + PMC passes + per-op table (tools/profile_job.sh), the one-utterance latency with the
# convolution kernels / the column-split block head forced down to its grid, the B = 16 schedule variants
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
bash tools/profile_job.sh r06 $1 > $O/r06_profile_job.log 2>&1; tail -3 $O/r06_profile_job.log
( echo "# default"; python tools/b1_latency.py 300 1024
  echo "# DVITS_CONV3_MIN_TILES=1"; DVITS_CONV3_MIN_TILES=1 python tools/b1_latency.py 300 1024
  echo "# DVITS_CONV3_MIN_TILES=1 DVITS_QKV_SPLIT_MIN_WG=1"; DVITS_CONV3_MIN_TILES=1 DVITS_QKV_SPLIT_MIN_WG=1 python tools/b1_latency.py 300 1024
  echo "# DVITS_QKV_SPLIT_MIN_WG=1"; DVITS_QKV_SPLIT_MIN_WG=1 python tools/b1_latency.py 300 1024 ) 2>&1 | grep -v amdgpu.ids > $O/r06_b1_latency_floors.txt
cat $O/r06_b1_latency_floors.txt
( for v in "X=1" "DVITS_GNX_ROUNDS=0" "DVITS_CONV3=0" "DVITS_CONV3=0 DVITS_GNX_ROUNDS=0 DVITS_QKV_SPLIT=0" "DVITS_QKV_SPLIT=0"; do
  echo "== $v"; ( export $v CWC_VARIANTS=default; python tools/conv_window_check.py 16,1024,256 2>&1 | grep "ms/forward" ); done ) > $O/r06_b16_schedule_variants.txt
cat $O/r06_b16_schedule_variants.txt
