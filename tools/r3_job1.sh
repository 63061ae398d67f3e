#!/bin/bash
# round-3 GPU job 1: parity sanity of the GEMM L2 prefetch + same-box A/B (DVITS_GEMM_PF=0/1): bench, per-op table, GEMM phase trace
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "not slow and not config2 and not config4" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do
  for pf in 0 1; do
    DVITS_GEMM_PF=$pf timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_pf${pf}_$rep.json 2> $O/bench_pf${pf}_$rep.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_pf${pf}_$rep.json").read().strip().splitlines()[-1])
    print("pf=$pf rep=$rep value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("pf=$pf rep=$rep FAILED", e)
PY
  done
done
for pf in 0 1; do
  DVITS_GEMM_PF=$pf timeout 600 python tools/profile_ops.py > $O/ops_pf$pf.txt 2>&1
  head -7 $O/ops_pf$pf.txt
done
for pf in 0 1; do
  DVITS_GEMM_PF=$pf timeout 600 python tools/gemm_trace.py 2048x1152x384 8192x384x128 1024x1536x512 8192x128x128 4096x256x2048 2048x384x3072 > $O/trace_pf$pf.txt 2>&1
done
grep -E "^M=|k-loop  |issue prologue|first tile|whole workgroup" $O/trace_pf0.txt | head -40
echo ---- pf1
grep -E "^M=|k-loop  |issue prologue|first tile|whole workgroup" $O/trace_pf1.txt | head -40
