#!/bin/bash
# Round 5, sixth pass: the opt-in fp8 KV cache (op test, model test, bench at B = 8 / 16 / 32 with and without), slot serving with 16 / 32 slots
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_f; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_operating_point_gpu.py -m gpu -x -q -s -k "fp8_kv or decode_attention" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "passed\|failed\|fp8 KV cache" $O/pytest.log | tail -8
bash tools/ab_bench.sh "b8 --batch-per-gpu 8" "b8_kv8 --batch-per-gpu 8 --kv-fp8" "b16_kv8 --batch-per-gpu 16 --kv-fp8" "b32_kv8 --batch-per-gpu 32 --kv-fp8" "b1_kv8 --kv-fp8" "fp8b16_kv8 --fp8 --batch-per-gpu 16 --kv-fp8" 2>&1 | tee $O/ab.txt
timeout 900 python tools/serve_bench.py --requests 96 --slots 16 2>$O/serve16.err | tail -1 > $O/serve_bench_16.json; python -c "
import json; d=json.load(open('$O/serve_bench_16.json')); print('16 slots', {k: (v.get('actions_per_s'), v.get('latency_p50_s')) for k,v in d.items() if isinstance(v, dict) and 'actions_per_s' in v}, d['requests_with_identical_ids'])"
timeout 1200 python tools/serve_bench.py --requests 128 --slots 32 2>$O/serve32.err | tail -1 > $O/serve_bench_32.json; python -c "
import json; d=json.load(open('$O/serve_bench_32.json')); print('32 slots', {k: (v.get('actions_per_s'), v.get('latency_p50_s')) for k,v in d.items() if isinstance(v, dict) and 'actions_per_s' in v}, d['requests_with_identical_ids'])"
tail -3 $O/serve32.err
