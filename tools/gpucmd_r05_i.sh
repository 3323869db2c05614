#!/bin/bash
# Round 5, ninth pass: the fp8-KV-cache error lines at full depth (pytest -s), slot serving with 32 slots on a longer request stream (384 requests:
# 12 per slot -- the 128-request run is mostly ramp and tail)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_i; mkdir -p $O
timeout 900 python -m pytest tests/test_full_depth_gpu.py -m gpu -x -q -s -k "fp8_kv_cache" 2>&1 | grep "full depth B=\|passed\|failed" | cut -c1-400 | tee $O/full_depth_kv8.txt
timeout 1500 python tools/serve_bench.py --requests 384 --slots 32 2>$O/serve.err | tail -1 > $O/serve_bench_32_384.json; python -c "
import json; d=json.load(open('$O/serve_bench_32_384.json')); print('32 slots, 384 requests', {k: (v.get('actions_per_s'), v.get('latency_p50_s'), v.get('decode_steps')) for k,v in d.items() if isinstance(v, dict) and 'actions_per_s' in v}, d['requests_with_identical_ids'])"
timeout 1500 python tools/serve_bench.py --requests 192 --slots 16 2>$O/serve.err | tail -1 > $O/serve_bench_16_192.json; python -c "
import json; d=json.load(open('$O/serve_bench_16_192.json')); print('16 slots, 192 requests', {k: (v.get('actions_per_s'), v.get('latency_p50_s'), v.get('decode_steps')) for k,v in d.items() if isinstance(v, dict) and 'actions_per_s' in v}, d['requests_with_identical_ids'])"
