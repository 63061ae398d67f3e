#!/bin/bash
# round-3 GPU job 9: lean address generation in the plain tile's k-loop - parity, A/B against the previous build, phase trace
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j9
mkdir -p $O
cd $R
export DVITS_GEMM_BD=0
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_prompt.py -x -q -m gpu -k "not slow and not handover and not competing" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2 3; do
  for v in prev cur; do
    lib=$R/diff-vits_amd/libdvits_hip_$v.so
    [ $v = cur ] && lib=$R/diff-vits_amd/libdvits_hip.so
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
timeout 300 python tools/gemm_trace.py 2048x1152x384 8192x384x128 1024x1536x512 8192x128x128 4096x256x2048 > $O/trace.txt 2>&1
grep -E "^M=|k-loop  |issue prologue|whole workgroup|sums" $O/trace.txt
timeout 600 python tools/profile_ops.py --summary > $O/ops.txt 2>&1; head -7 $O/ops.txt
