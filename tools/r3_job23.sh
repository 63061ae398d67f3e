#!/bin/bash
# same-box A/B: committed build (prev2) vs working tree
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j23
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
for rep in 1 2 3 4; do
  run head_$rep DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_prev2.so
  run tree_$rep DVITS_DUMMY=1
done
DVITS_LIB_FILE=$R/diff-vits_amd/libdvits_hip_prev2.so timeout 600 python tools/profile_ops.py > $O/ops_head.txt 2>&1
timeout 600 python tools/profile_ops.py > $O/ops_tree.txt 2>&1
head -7 $O/ops_head.txt | tail -6; head -7 $O/ops_tree.txt | tail -6
