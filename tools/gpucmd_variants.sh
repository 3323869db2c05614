cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-variants}; mkdir -p $O
: > $O/bench_variants.jsonl
for v in "" "--batch-per-gpu 8" "--fp8" "--fp8 --batch-per-gpu 8" "--fp8 --batch-per-gpu 8 --graph"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v 2>/dev/null | tail -1 >> $O/bench_variants.jsonl
done
python - "${1:-variants}" <<'PY'
import json,sys,os
for l in open(os.environ.get("O","gpurun_out/variants")+"/bench_variants.jsonl") if False else open("gpurun_out/%s/bench_variants.jsonl" % (sys.argv[1] if len(sys.argv)>1 else "variants")):
    d=json.loads(l); print(d["config"]["batch_per_gpu"], d["dtype"][:12], d["config"]["hipgraph"], "value", d["value"], "ms/tok", d["decode_ms_per_token"], "frac", d["decode_step_hbm_frac"], d["stage_us"])
PY
