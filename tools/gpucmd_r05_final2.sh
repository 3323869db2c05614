#!/bin/bash
# Round 5, session 2, final check of the last build: the whole GPU suite, the driver's smoke, and the bit-regression probe against the
# session-start build (scratch_prev/ = the tree of commit 5273362 with its library, staged by hand, not committed): with the new forms off
# (EMMAX_ATTN_KSPLIT=0 EMMAX_ATTN_LAZY=0 EMMAX_GEMM_SK_BIG=0) the build must reproduce it bit for bit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
if [ -d scratch_prev ]; then
  (cd scratch_prev && timeout 600 python tools/regress_bits.py prev > ../$O/regress_prev.log 2>&1)
  EMMAX_ATTN_KSPLIT=0 EMMAX_ATTN_LAZY=0 EMMAX_GEMM_SK_BIG=0 timeout 600 python tools/regress_bits.py now_off --against prev 2>&1 | grep " vs " | tee $O/regress_bits.txt
  timeout 600 python tools/regress_bits.py now --against prev 2>&1 | grep " vs " | tee -a $O/regress_bits.txt
fi
for sw in 0 -1 0 -1; do EMMAX_ATTN_KSPLIT=$sw EMMAX_GEMM_SK_BIG=$sw timeout 300 python tools/stage_bench.py --vision-batches 1,8 --prefill-batches 1,8 2>/dev/null | tail -1 | sed "s/^/attn_ksplit=gemm_sk_big=$sw /" >> $O/prefill_ab.txt; done; cat $O/prefill_ab.txt
