#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r3_bench_$label.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_$label.json").read())
    print("$label", d["value"], d["ms_per_step"], d["decode_ms_per_token"], d.get("stage_us"))
except Exception as e:
    print("$label failed", e, open("gpurun_out/r3_bench_$label.json").read()[:500])
PY
}
run default A=1
run nsplit4 EMMAX_ATTN_NSPLIT=4
run ksoproj EMMAX_KS_OPROJ=1
run ksoproj256 EMMAX_KS_OPROJ=1 EMMAX_KS_OPROJ_GRID=256
run ksoproj256n4 EMMAX_KS_OPROJ=1 EMMAX_KS_OPROJ_GRID=256 EMMAX_ATTN_NSPLIT=4
