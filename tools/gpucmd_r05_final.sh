#!/bin/bash
# Round 5, final check of the last build: the whole GPU suite, the driver's smoke, and the bit-regression probe against the round-4 build
# (scratch_r04/ = the tree of commit 71625fb with its library, staged by hand, not committed): with the fp32 residual stream switched off
# (EMMAX_RESID32=0) the final build must reproduce round 4 bit for bit -- patch embeddings, all 768 prefill logit rows, greedy ids at B = 1 and a ragged B = 8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
if [ -d scratch_r04 ]; then
  (cd scratch_r04 && timeout 600 python tools/regress_bits.py r04 > ../$O/regress_r04.log 2>&1)
  EMMAX_RESID32=0 timeout 600 python tools/regress_bits.py r05_resid0 --against r04 2>&1 | grep " vs " | tee $O/regress_bits.txt
  timeout 600 python tools/regress_bits.py r05 --against r04 2>&1 | grep " vs " | tee -a $O/regress_bits.txt
fi
