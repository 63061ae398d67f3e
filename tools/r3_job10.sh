#!/bin/bash
# round-3 GPU job 10: tile-menu thresholds (GEGLU at 1.5 rounds of 128x128 tiles) + BD test
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j10
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "bd_tile" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
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
for rep in 1 2; do
  run base_$rep DVITS_X=0
  run big400_$rep DVITS_GEMM_CFG=400,128,1,1
  run big260_$rep DVITS_GEMM_CFG=260,128,1,1
  run minwg192_$rep DVITS_GEMM_CFG=144,192,1,1
  run minwg96_$rep DVITS_GEMM_CFG=144,96,1,1
done
DVITS_GEMM_CFG=400,128,1,1 timeout 600 python tools/profile_ops.py > $O/ops_big400.txt 2>&1
grep "epi=2" $O/ops_big400.txt
