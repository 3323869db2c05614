cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -6
bash tools/profile_round.sh variants 2>&1 | tail -2
cat gpurun_out/prof_round/bench_variants.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:70], '|', d['value'], d['decode_ms_per_token'], d['config'].get('hipgraph'))
"
