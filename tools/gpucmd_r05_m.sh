# round 5, session 2: split-K geometry sweep for the short-prefill o-proj / down (tools/gemm_sk_sweep.py), prefill A/B of the planned big-tile split-K
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_operating_point_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for sw in 0 -1 0 -1; do EMMAX_GEMM_SK_BIG=$sw timeout 300 python tools/stage_bench.py --vision-batches 1,8 --prefill-batches 1,2,3 2>/dev/null | tail -1 | sed 's/"vision".*"prefill"/"prefill"/' | sed "s/^/gemm_sk_big=$sw /" >> $O/prefill_ab.txt; done; cat $O/prefill_ab.txt
