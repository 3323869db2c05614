# round 5, session 2: decode attention K / V rows with non-temporal loads (build B = -DEMMAX_ATTN_KV_NT) against plain loads (build A), alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
L=emma-x_amd/emmax
for rnd in 1 2; do for v in A B; do
  cp $L/lab_$v.so $L/libemmax_hip.so
  for fl in "" "--batch-per-gpu 8" "--batch-per-gpu 32" "--batch-per-gpu 8 --kv-fp8" "--batch-per-gpu 32 --kv-fp8"; do
    timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', '[$fl]', d['value'], d['ms_per_step'], d['decode_ms_per_token'], d['stage_us']['paged_attn'])" >> $O/attn_nt_ab.txt
  done
done; done
cat $O/attn_nt_ab.txt
