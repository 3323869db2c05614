#!/bin/bash
# decode attention split-count sweep at B=1 (tuning aid)
for NS in 2 4 8 16; do
  EMMAX_ATTN_NSPLIT=$NS timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 2>&1 | tail -1 > /tmp/ns.json
  python -c "import json; d=json.load(open('/tmp/ns.json')); print('nsplit', $NS, 'ms/token', d['decode_ms_per_token'], 'attn us', d['stage_us']['paged_attn'], 'oproj us', d['stage_us']['oproj_gemv'])"
done
