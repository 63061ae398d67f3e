#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j21
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
  run rp0_$rep DVITS_SKIP_PLANES=0
  run rp1_$rep DVITS_DUMMY=1
done
DVITS_SKIP_PLANES=0 timeout 600 python tools/profile_ops.py > $O/ops_rp0.txt 2>&1
timeout 600 python tools/profile_ops.py > $O/ops_rp1.txt 2>&1
head -8 $O/ops_rp0.txt; head -8 $O/ops_rp1.txt; grep gn_apply $O/ops_rp1.txt | head -20
