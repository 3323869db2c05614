#!/bin/bash
# round 3: op tests of the K-split GEMV, then the headline bench with it on / off (EMMAX_KS)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemv or decode_attention" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_operating_point_gpu.py -x -q -m gpu 2>&1 | tail -5
for ks in 1 0; do
  EMMAX_KS=$ks timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r3_bench_ks$ks.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_ks$ks.json").read())
print("KS=$ks", d["value"], d["ms_per_step"], d.get("stage_us"), d.get("roofline"))
PY
done
