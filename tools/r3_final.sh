#!/bin/bash
# round-3 evidence job: full -m gpu suite, the default bench line (with cpu_baseline + roofline + parity fields), rocprofv3 kernel
# statistics + PMC passes (stamped with the build identity), per-operation table, GEMM phase trace
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
bash tools/profile_job.sh r03 425603b > $O/profile_job.log 2>&1
tail -3 $O/profile_job.log
cp gpurun_out/r03_pmc_roofline.json profiles/r03_pmc_roofline.json   # (so that the bench line below can quote it: same build)
timeout 1200 python bench.py > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_final.json").read().strip().splitlines()[-1])
print("value=%.0f ms_per_step=%.2f frac=%.4f cpu=%.0f rel_l2=%s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("extra", {}).get("unet_rel_l2")))
print(d["roofline"]["traffic_source"])
PY
timeout 300 python tools/gemm_trace.py > $O/gemm_trace.txt 2>&1
timeout 900 python tools/gemm_trace_fwd.py > $O/gemm_trace_fwd.txt 2>&1
