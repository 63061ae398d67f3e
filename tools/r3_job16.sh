#!/bin/bash
# A/B of HIP runtime knobs around kernel-argument placement, graph packet capture and fence scopes (same box, interleaved)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j16
mkdir -p $O
cd $R
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
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
  run base_$rep DVITS_DUMMY=1
  run devkernarg1_$rep HIP_FORCE_DEV_KERNARG=1
  run devkernarg0_$rep HIP_FORCE_DEV_KERNARG=0
  run optflush0_$rep AMD_OPT_FLUSH=0
  run pktcap0_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run pktcap1_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run syssig0_$rep ROC_SYSTEM_SCOPE_SIGNAL=0
  run gbatch_$rep DEBUG_HIP_GRAPH_BATCH_SIZE=1024
done
