#!/bin/bash
# round-3 GPU job 12: key split inside the attention workgroup at the short levels (k_attention_frag<..., KSP = 2>)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j12
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_prompt.py -x -q -m gpu -k "not slow and not handover and not competing" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value=%.0f ms_per_step=%.2f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  run ksp0_$rep DVITS_ATTNF_KSP=0
  run ksp160_$rep DVITS_ATTNF_KSP=160
  run ksp300_$rep DVITS_ATTNF_KSP=300
done
for k in 0 160 300; do DVITS_ATTNF_KSP=$k timeout 600 python tools/profile_ops.py > $O/ops_ksp$k.txt 2>&1; echo "== ksp $k"; head -5 $O/ops_ksp$k.txt | tail -4; grep "^attn" $O/ops_ksp$k.txt; done
