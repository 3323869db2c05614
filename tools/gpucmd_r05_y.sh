# round 5, session 2: cache-policy bits of decode_ks.hip's weight stream (raw_buffer_load aux: 1 = sc0, 2 = nt, 16 = sc1; builds -DKS_W_AUX=n, alternating with the product's nt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=emma-x_amd/emmax; cp $L/libemmax_hip.so $L/lab_aux2.so
for a in 2 0 1 3 16 17 18 19 2; do
  cp $L/lab_aux$a.so $L/libemmax_hip.so
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('aux=$a', d['value'], d['ms_per_step'], d['decode_ms_per_token'], d['stage_us'])"
done 2>&1 | tee gpurun_out/r05y_ks_aux.txt
