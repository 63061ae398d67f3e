#!/bin/bash
# round-3 GPU job 8: cost of the per-run hand-over verification; attention wave-count variants
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j8
mkdir -p $O
cd $R
export DVITS_GEMM_BD=0
run() {  # name, env...
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
for rep in 1 2; do
  run verify1_$rep DVITS_HANDOVER_VERIFY=1
  run verify0_$rep DVITS_HANDOVER_VERIFY=0
  run attn_nw4_600_$rep DVITS_HANDOVER_VERIFY=0 DVITS_ATTNF_NW4=600
  run attn_nw8_500_$rep DVITS_HANDOVER_VERIFY=0 DVITS_ATTNF_NW8=500
  run attn_nw8_2000_$rep DVITS_HANDOVER_VERIFY=0 DVITS_ATTNF_NW8=2000
done
