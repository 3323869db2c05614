#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_operating_point_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4
run() { # label, bench args..., env via ENVV
  label=$1; shift
  env $ENVV timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 > gpurun_out/r3_bench_$label.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_$label.json").read())
    print("$label", d["value"], d["ms_per_step"], d["decode_ms_per_token"], d.get("stage_us"))
except Exception as e:
    print("$label failed", e, open("gpurun_out/r3_bench_$label.json").read()[:500])
PY
}
ENVV="A=1" run b8 --batch-per-gpu 8
ENVV="A=1" run f8b8 --batch-per-gpu 8 --fp8
ENVV="EMMAX_FOLD_EMBED=1" run b1_fold1
ENVV="EMMAX_FOLD_EMBED=0" run b1_fold0
