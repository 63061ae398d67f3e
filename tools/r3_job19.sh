#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j19
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  run prev_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_prev.so
  run new_$rep DVITS_DUMMY=1
done
timeout 900 python tools/gemm_trace_fwd.py > $O/gemm_trace_fwd.txt 2>&1
tail -1 $O/gemm_trace_fwd.txt | cut -c1-200
for sel in "384 1 0" "256 0 1" "128 0 1" "384 0 0" "256 1 0" "128 1 0"; do
  echo "== chain $sel" >> $O/chain_trace.txt
  timeout 300 python tools/chain_trace.py 8 1024 $sel >> $O/chain_trace.txt 2>&1
done
timeout 300 python tools/chain_trace.py 8 1024 >> $O/chain_trace.txt 2>&1
grep -v amdgpu.ids $O/chain_trace.txt | cut -c1-150
