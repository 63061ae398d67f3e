#!/bin/bash
# the round's final evidence from ONE box: the -m gpu suite, smoke(), the default bench line, rocprofv3 / PMC passes, the phase traces
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
bash tools/profile_job.sh r06 $1 > $O/r06_profile_job.log 2>&1; tail -2 $O/r06_profile_job.log
cp $O/r06_pmc_roofline.json $R/profiles/r06_pmc_roofline.json     # (bench.py quotes the newest committed file of THIS build)
bash tools/round_check.sh r06_final
python tools/qkv_trace.py 2>&1 | grep -v amdgpu.ids > $O/r06_qkv_phase_trace_final.txt
python tools/conv3_trace.py 2>&1 | grep -v amdgpu.ids > $O/r06_conv3_phase_trace.txt
