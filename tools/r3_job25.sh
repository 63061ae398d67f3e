#!/bin/bash
# pair-barrier k-loop (DV_GEMM_PAIRS=1 build) vs the default build: parity tests on the pairs build, then same-box A/B
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j25
mkdir -p $O
cd $R
DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_p1.so timeout 1800 python -m pytest tests/test_gpu_unet.py tests/test_gpu_ops.py -x -q -m gpu > $O/pytest_p1.log 2>&1; echo "pytest(p1) rc=$?"; tail -2 $O/pytest_p1.log
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3 4; do
  run p0_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_p0.so
  run p1_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_p1.so
done
DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_p0.so timeout 600 python tools/profile_ops.py > $O/ops_p0.txt 2>&1
DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_p1.so timeout 600 python tools/profile_ops.py > $O/ops_p1.txt 2>&1
head -4 $O/ops_p0.txt | tail -3; head -4 $O/ops_p1.txt | tail -3
