#!/bin/bash
# roofline leg (live back-to-back GEMM replay) and graph time, DMA spread over all groups (e0) vs behind the first (e1 = tree), same box
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j27
mkdir -p $O
cd $R
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$name value=%.0f ms_per_step=%.2f live_us=%.2f frac=%.4f evpair_us=%.2f gemm_ms=%.3f" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r["avg_op_us_event_pair_per_operation"], r["per_kind_ms_per_forward"]["gemm"]))
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  run e0_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_e0.so
  run e1_$rep DVITS_DUMMY=1
done
