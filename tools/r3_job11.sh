#!/bin/bash
# round-3 GPU job 11: in-epilogue GroupNorm exchange through the XCD's L2 (DVITS_GNX_LOCAL=1, default) vs through memory (=0)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j11
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "not slow" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
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
  run local0_$rep DVITS_GNX_LOCAL=0
  run local1_$rep DVITS_GNX_LOCAL=1
done
timeout 600 python tools/flaky_repeat.py > $O/flaky.txt 2>&1; tail -5 $O/flaky.txt
for l in 0 1; do DVITS_GNX_LOCAL=$l timeout 600 python tools/profile_ops.py --summary > $O/ops_local$l.txt 2>&1; head -4 $O/ops_local$l.txt | tail -3; done
