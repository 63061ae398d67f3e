#!/bin/bash
# round-3 GPU job 15: P of the attention as ONE bf16 plane in P V (two products) - error against the goldens and speed
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j15
mkdir -p $O
cd $R
for v in cur plo0; do
  lib=$R/diff-vits_amd/libdvits_hip_$v.so; [ $v = cur ] && lib=$R/diff-vits_amd/libdvits_hip.so
  echo "== $v"; DVITS_LIB_FILE=$lib timeout 600 python tools/err_probe.py 2>/dev/null | tee $O/err_$v.txt
done
for rep in 1 2 3; do
  for v in cur plo0; do
    lib=$R/diff-vits_amd/libdvits_hip_$v.so; [ $v = cur ] && lib=$R/diff-vits_amd/libdvits_hip.so
    DVITS_LIB_FILE=$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
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
