#!/bin/bash
# round 6, first GPU call: the new verification tests + the two-half-batches experiment
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_job1; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_prompt.py -x -q -m gpu -k "timeout or downgraded or loops_issue or handover or survives or replan or exclusive" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
bash tools/ab_bench.sh r06_halves 3 base=- two_keep=DVITS_BENCH_STREAMS=2,DVITS_BENCH_KEEP_HANDOVER=1,DVITS_CU_BUDGET=128 two_off=DVITS_BENCH_STREAMS=2 2>&1 | tail -14
grep -h "timed_out\|downgraded" $O/../r06_halves/*.json | head -5 | cut -c1-400
