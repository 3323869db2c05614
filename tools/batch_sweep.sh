cd $GRAFT_REPO_ROOT
for b in 2 3 4 8; do python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch-per-gpu $b --new-tokens 64 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], d['decode_ms_per_token'], d['stage_us'])"; done
