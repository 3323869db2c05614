# round 5, session 2: causal prefill attention -- two key groups per block (attn_ksplit) and the lazy reference maximum (head_dim 128)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $O/attn_tests.txt 2>&1; tail -5 $O/attn_tests.txt
timeout 300 python tools/attn_ksplit_ab.py > $O/attn_ksplit_ab.txt 2>&1; cat $O/attn_ksplit_ab.txt
for sw in 0 -1 0 -1; do EMMAX_ATTN_KSPLIT=$sw timeout 300 python tools/stage_bench.py --vision-batches 1,8 --prefill-batches 1,8 2>/dev/null | tail -1 | sed "s/^/attn_ksplit=$sw /" >> $O/prefill_ab.txt; done; cat $O/prefill_ab.txt
