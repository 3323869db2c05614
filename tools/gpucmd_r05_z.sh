# round 5, session 2: split-K partial tiles through non-temporal stores / loads (build B = -DSK_NT) against plain (build A), alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=emma-x_amd/emmax
for rnd in 1 2; do for v in A B; do
  cp $L/lab_$v.so $L/libemmax_hip.so
  timeout 300 python tools/stage_bench.py --vision-batches 1 --prefill-batches 1,2 2>/dev/null | tail -1 | sed "s/^/$v /"
done; done 2>&1 | tee gpurun_out/r05z_sk_nt.txt
cp $L/lab_B.so $L/libemmax_hip.so; SK_MS=768,1536 timeout 300 python tools/gemm_sk_sweep.py 2>&1 | grep -v amdgpu | cut -c1-200 | sed "s/^/B /" | tee -a gpurun_out/r05z_sk_nt.txt
cp $L/lab_A.so $L/libemmax_hip.so; SK_MS=768,1536 timeout 300 python tools/gemm_sk_sweep.py 2>&1 | grep -v amdgpu | cut -c1-200 | sed "s/^/A /" | tee -a gpurun_out/r05z_sk_nt.txt
