#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j14
mkdir -p $O
cd $R
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_$name.json 2> $O/bench_$name.err
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
  run base_$rep DVITS_X=0
  run minwg160_$rep DVITS_GEMM_CFG=144,160,1,1
  run sk128_$rep DVITS_SPLITK=160,1024,2
  run sk512_$rep DVITS_SPLITK=160,512,2
done
DVITS_GEMM_CFG=144,160,1,1 timeout 600 python tools/profile_ops.py > $O/ops_minwg160.txt 2>&1; grep "M=1024" $O/ops_minwg160.txt | head -20
