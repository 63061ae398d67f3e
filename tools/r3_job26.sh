#!/bin/bash
# DMA units of the next tile spread over the first 1 / 2 / 3 MFMA groups of a k-tile instead of all six (DV_GEMM_EARLY builds)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j26
mkdir -p $O
cd $R
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
for rep in 1 2 3; do
  for e in 0 1 2; do run e${e}_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_e$e.so; done
done
