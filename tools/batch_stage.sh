#!/bin/bash
# Tuning aid: decode stage timings per batch size.  usage: tools/batch_stage.sh [--fp8] B...
extra=""
if [ "$1" == "--fp8" ]; then extra="--fp8"; shift; fi
for b in "$@"; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --new-tokens 128 --batch-per-gpu $b $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('B', d['config']['batch_per_gpu'], '$extra', d['decode_ms_per_token'], d['stage_us'])
" | tee -a gpurun_out/batch_stage.log
done
