#!/bin/bash
# round-3 GPU job 4: BD tile - op-level parity, phase trace (BD vs plain tile), bench A/B
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/pytest_ops.log 2>&1
echo "pytest rc=$?" >> $O/pytest_ops.log
tail -8 $O/pytest_ops.log
for bd in 1 0; do
  DVITS_GEMM_BD=$bd timeout 300 python tools/gemm_trace.py 2048x1152x384 8192x384x128 1024x1536x512 8192x128x128 conv:8x256x384x384x3 conv:8x1024x128x128x3 > $O/trace_bd$bd.txt 2>&1
  echo "== bd=$bd"; grep -E "^M=|k-loop  |issue prologue|first tile|whole workgroup|epilogue|k-split" $O/trace_bd$bd.txt
done
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "golden or layerwise or config2_full_size or split_k or cfg1" > $O/pytest_unet.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unet.log
tail -5 $O/pytest_unet.log
for rep in 1 2; do
  for bd in 0 1; do
    DVITS_GEMM_BD=$bd timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_bd${bd}_$rep.json 2> $O/bench_bd${bd}_$rep.err
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
for bd in 0 1; do
  DVITS_GEMM_BD=$bd timeout 600 python tools/profile_ops.py > $O/ops_bd$bd.txt 2>&1
  head -7 $O/ops_bd$bd.txt
done
