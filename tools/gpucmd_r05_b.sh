#!/bin/bash
# Round 5, second pass: decode batches 9-16 (op tests of decode_km.hip, the 16-row model test), fp32 master + bf16 mirror, sessions without staging rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_b; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_operating_point_gpu.py tests/test_serving_gpu.py tests/test_e2e_gpu.py tests/test_gqa_limits_gpu.py -m gpu -x -q -k "km or nine_to_sixteen or residual_stream or serving or slots or e2e or gqa or decode" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
: > $O/bench.jsonl
for v in "--batch-per-gpu 8" "--batch-per-gpu 16" "--batch-per-gpu 12" "--fp8 --batch-per-gpu 16" "--batch-per-gpu 16 --graph"; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v 2>$O/bench.err | tail -1 >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r05_b/bench.jsonl'):
    try: d=json.loads(l)
    except Exception as e: print('bad line', l[:200]); continue
    print(d['config'].get('batch_per_gpu'), d['dtype'], d['config'].get('hipgraph'), d['value'], d['ms_per_step'], d.get('decode_ms_per_token'), d.get('decode_step_hbm_frac'), d['roofline']['us_per_launch'])
PY
tail -5 $O/bench.err
