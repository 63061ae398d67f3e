#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_job2; mkdir -p $O; cd $R
CWC_VARIANTS=ring,default,rounds0 timeout 1500 python tools/conv_window_check.py 16,99,60 16,128,60 8,300,150 16,1024,256 4,2048,256 32,512,100 > $O/conv_window2.txt 2>&1; echo "check rc=$?"; grep -v amdgpu.ids $O/conv_window2.txt | tail -40
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "golden or odd or resident or fused_schedule or last_forward or config5 or split_feed" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
