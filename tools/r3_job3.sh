#!/bin/bash
# round-3 GPU job 3: full -m gpu suite on the current build; A/B of the chain kernels' prefetch modes (DV_CHAIN_PFMODE 0/1/2)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  for v in pf0 pf1 cur; do
    lib=$R/diff-vits_amd/libdvits_hip_$v.so
    [ $v = cur ] && lib=$R/diff-vits_amd/libdvits_hip.so
    DVITS_LIB_FILE=$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1])
    print("$v rep=$rep value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$v rep=$rep FAILED", e)
PY
  done
done
for v in pf0 pf1 cur; do
  lib=$R/diff-vits_amd/libdvits_hip_$v.so
  [ $v = cur ] && lib=$R/diff-vits_amd/libdvits_hip.so
  DVITS_LIB_FILE=$lib timeout 600 python tools/profile_ops.py > $O/ops_$v.txt 2>&1
  echo "== $v"; head -4 $O/ops_$v.txt | tail -3; grep "^chain" $O/ops_$v.txt
done
