#!/bin/bash
# A/B of bench.py variants on ONE box (boxes differ by ~2 %): every line = label, then the environment / flags of that run.
#   usage (through gpurun):  bash tools/ab_bench.sh "ks1 EMMAX_KS=1" "ks0 EMMAX_KS=0" "b8 --batch-per-gpu 8" ...
# prints actions/s, ms/step, decode ms/token and the per-stage launch times of each variant
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for spec in "$@"; do
  set -- $spec
  label=$1; shift
  envs=(); flags=()
  for a in "$@"; do case $a in *=*) envs+=("$a");; *) flags+=("$a");; esac; done
  env "${envs[@]}" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "${flags[@]}" 2>&1 | tail -1 > gpurun_out/ab_$label.json
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{lab}.json").read())
    print(lab, d["value"], d["ms_per_step"], d["decode_ms_per_token"], d.get("stage_us"))
except Exception as e:
    print(lab, "failed", e, open(f"gpurun_out/ab_{lab}.json").read()[:400])
PY
done
