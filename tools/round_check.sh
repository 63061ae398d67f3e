#!/bin/bash
# what the driver runs at round end - build check is local; here: the -m gpu suite, smoke(), the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-check}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value=%.0f ms_per_step=%.2f frac=%.4f traffic=%s rel=%s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["extra"]["unet_rel_l2"]))
PY
