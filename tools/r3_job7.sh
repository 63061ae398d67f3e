#!/bin/bash
# round-3 GPU job 7: full -m gpu suite (hand-over recovery, BD tile tests) + BD default threshold A/B
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j7
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for rep in 1 2 3; do
  for bd in 0 1; do
    DVITS_GEMM_BD=$bd timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_bd${bd}_$rep.json 2> $O/bench_bd${bd}_$rep.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_bd${bd}_$rep.json").read().strip().splitlines()[-1])
    print("bd=$bd rep=$rep value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("bd=$bd rep=$rep FAILED", e)
PY
  done
done
