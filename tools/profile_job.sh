#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on the GPU box (copy what is to be kept into profiles/):
#   kernel-trace statistics + three PMC passes over the bench command (eager launches), the per-family roofline JSON,
#   and the per-operation profile.  Usage (from the repo root, via gpurun):  bash tools/profile_job.sh <tag>
set -u
TAG=${1:-r03}
GIT_HEAD=${2:-$(git -C ${GRAFT_REPO_ROOT:-$PWD} rev-parse --short HEAD 2>/dev/null || echo unknown)}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp DVITS_NO_GRAPH=1
CMD="python3 $R/bench.py --steps 1 --warmup 1 --solver-steps 6 --no-cpu-baseline --no-roofline --no-other-configs"
rm -rf $R/gpurun_out/p_stats $R/gpurun_out/p_fetch $R/gpurun_out/p_write $R/gpurun_out/p_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats -- $CMD > $R/gpurun_out/p_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- $CMD > $R/gpurun_out/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- $CMD > $R/gpurun_out/p_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p_mfma -- $CMD > $R/gpurun_out/p_mfma.log 2>&1
cd $R
python3 tools/pmc_roofline.py gpurun_out/p_stats gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_mfma $GIT_HEAD > gpurun_out/${TAG}_pmc_roofline.json
cp $(find gpurun_out/p_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
unset DVITS_NO_GRAPH
python3 tools/profile_ops.py > gpurun_out/${TAG}_ops_profile.txt 2>&1
# the counter CSVs are large: keep only the summaries
rm -rf gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_mfma gpurun_out/p_stats
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmc_roofline.json"))
print({k: (v["launches"], round(v["avg_us_kernel_trace"], 1), round(v["mfma_util"], 3), round(v["hbm_gbps"])) for k, v in d.items() if isinstance(v, dict) and "launches" in v})
print(d.get("build"))
PY
