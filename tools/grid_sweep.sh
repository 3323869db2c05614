#!/bin/bash
# Tuning aid (not part of the product): decode stage timings for several persistent-grid overrides (EMMAX_GEMV_GRID hook).
# usage: tools/grid_sweep.sh "0,0,0,0,0" "256,0,0,0,0" ...
mkdir -p gpurun_out
for g in "$@"; do
  EMMAX_GEMV_GRID="$g" python bench.py --steps 1 --warmup 1 --no-cpu-baseline --new-tokens 128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$g', d['decode_ms_per_token'], d['stage_us'])
" | tee -a gpurun_out/grid_sweep.log
done
