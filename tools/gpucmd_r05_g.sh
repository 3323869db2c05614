#!/bin/bash
# Round 5, seventh pass: fp8 KV cache tests (op + model), slot serving with 32 slots over the fp8 KV cache
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_g; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_operating_point_gpu.py -m gpu -x -q -s -k "fp8_kv or decode_attention" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "passed\|failed\|fp8 KV cache\|Error\|^E " $O/pytest.log | tail -12 | cut -c1-400
EMMAX_KV_FP8=1 timeout 1200 python tools/serve_bench.py --requests 128 --slots 32 2>$O/serve32.err | tail -1 > $O/serve_bench_32_kvfp8.json; python -c "
import json; d=json.load(open('$O/serve_bench_32_kvfp8.json')); print('32 slots, fp8 KV', {k: (v.get('actions_per_s'), v.get('latency_p50_s')) for k,v in d.items() if isinstance(v, dict) and 'actions_per_s' in v}, d['requests_with_identical_ids'])"
bash tools/ab_bench.sh "b8_kv8 --batch-per-gpu 8 --kv-fp8" "b16_kv8 --batch-per-gpu 16 --kv-fp8" "b32_kv8 --batch-per-gpu 32 --kv-fp8" "b32 --batch-per-gpu 32" 2>&1 | tee $O/ab.txt
