#!/bin/bash
# Same-box A/B of the headline bench line (run on the GPU box through gpurun, from the repo root):
#   bash tools/ab_bench.sh <tag> <rounds> <name>=<ENV=VAL,ENV=VAL,...|-> [<name>=...] ...
# Every variant is one `python bench.py` configuration given as environment settings ("-" = none); the variants are run
# round-robin <rounds> times (boxes drift: interleaving keeps the comparison fair), one line per run is appended to
# gpurun_out/<tag>/ab.txt and the medians are printed at the end.  Examples:
#   bash tools/ab_bench.sh wt 3 base=- wt=DVITS_LIB_FILE=diff-vits_amd/libdvits_hip_wt1.so
#   bash tools/ab_bench.sh gnx 3 on=- off=DVITS_GNX=0
# AB_BENCH_ARGS overrides the bench arguments (default: 6 timed runs, no CPU baseline / roofline / other configurations).
set -u
TAG=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
ARGS=${AB_BENCH_ARGS:---steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs}
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}
    (
      if [ "$envs" != "-" ]; then
        IFS=',' read -ra kv <<< "$envs"
        for e in "${kv[@]}"; do export "$e"; done
      fi
      timeout 600 python bench.py $ARGS 2> $O/$name.err | tail -1 > $O/$name.json
    )
    python3 - "$name" "$O/$name.json" >> $O/ab.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    print("%-14s %9.0f mel-frames/s  %8.3f ms/run" % (sys.argv[1], d["value"], d["ms_per_step"]))
except Exception as e:
    print("%-14s FAILED %s" % (sys.argv[1], e))
PY
    tail -1 $O/ab.txt
  done
done
python3 - $O/ab.txt <<'PY'
import statistics, sys
rows = {}
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) >= 2 and f[1] != "FAILED":
        rows.setdefault(f[0], []).append(float(f[1]))
base = None
for k, v in rows.items():
    med = statistics.median(v)
    base = base or med
    print("median %-14s %9.0f  (%+.2f %% vs %s, n=%d)" % (k, med, 100.0 * (med / base - 1.0), next(iter(rows)), len(v)))
PY
