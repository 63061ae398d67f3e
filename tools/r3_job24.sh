#!/bin/bash
# repeatability of the final build: 60 forwards per shape, bit-equal, no hand-over time-out; then the gpu suite twice more
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j24
mkdir -p $O
cd $R
for shp in "8,1024,256" "2,2048,300" "1,300,150" "16,512,256"; do
  SHAPE=$shp timeout 600 python tools/flaky_repeat.py fused > $O/flaky_$shp.txt 2>&1
  echo "SHAPE=$shp: $(tail -1 $O/flaky_$shp.txt)"
done
for rep in 1 2; do
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_$rep.log 2>&1; echo "pytest $rep rc=$?"; tail -1 $O/pytest_$rep.log
done
