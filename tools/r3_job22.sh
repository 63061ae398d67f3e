#!/bin/bash
# batch sweep of the final build (bench.py --batch), B=1 latency tool
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j22
mkdir -p $O
cd $R
for b in 1 2 4 8 16; do
  timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err
  python - <<PY
import json
d = json.loads(open("$O/bench_b$b.json").read().strip().splitlines()[-1])
print("B=$b value=%.0f ms_per_step=%.2f ms_per_forward=%.3f" % (d["value"], d["ms_per_step"], d["ms_per_step"] / 50))
PY
done
timeout 600 python tools/b1_latency.py > $O/b1_latency.txt 2>&1; tail -8 $O/b1_latency.txt
