#!/bin/bash
# round-3 GPU job 6: BD tile end to end - where does it pay?  DVITS_GEMM_BD = 0 (off) | 1 (single-segment k=3, K >= 1152) | 768 | 384 | 2 (all)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "not slow" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2; do
  for bd in 0 1 768 384 2; do
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
for bd in 0 2; do
  DVITS_GEMM_BD=$bd timeout 600 python tools/profile_ops.py > $O/ops_bd$bd.txt 2>&1
done
python - <<PY
import re
def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r"gemm\s+(M=.*?)\s+x(\d+)\s+([\d.]+) us", ln)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load("$O/ops_bd0.txt"), load("$O/ops_bd2.txt")
print("%-100s %5s %8s %8s" % ("gemm", "n", "plain", "BD-all"))
for k in sorted(a, key=lambda k: -a[k][1] * a[k][0]):
    if k in b and abs(a[k][1] - b[k][1]) > 0.3: print("%-100s %5d %8.1f %8.1f" % (k, a[k][0], a[k][1], b[k][1]))
PY
