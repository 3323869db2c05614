#!/bin/bash
# Round 5, first pass: the GPU suite on the fp32-residual build, then the bench A/B (resid32 = 1 / 0) at B = 1 and B = 8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
: > $O/bench_ab.jsonl
for r in 1 0 1 0; do
  for v in "" "--batch-per-gpu 8"; do
    EMMAX_RESID32=$r timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'resid32': $r, 'v': '$v', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'decode_ms_per_token': d.get('decode_ms_per_token'), 'roofline': d.get('roofline')}))" >> $O/bench_ab.jsonl
  done
done
cat $O/bench_ab.jsonl
