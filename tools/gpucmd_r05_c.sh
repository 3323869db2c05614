#!/bin/bash
# Round 5, third pass: the whole GPU suite on the build with the fp32 prefill stream, then A/Bs on one box:
# km_roll / attn_nw at B = 8 and 16, the prefill with and without the fp32 stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "passed\|failed\|slot-served\|512-step" $O/pytest.log | cut -c1-1500 | tail -8
bash tools/ab_bench.sh "b8_base --batch-per-gpu 8" "b8_roll EMMAX_KM_ROLL=1 --batch-per-gpu 8" "b8_nw8 EMMAX_ATTN_NW=8 --batch-per-gpu 8" "b8_both EMMAX_KM_ROLL=1 EMMAX_ATTN_NW=8 --batch-per-gpu 8" \
  "b8_base2 --batch-per-gpu 8" "b16_base --batch-per-gpu 16" "b16_roll EMMAX_KM_ROLL=1 --batch-per-gpu 16" "b16_nw8 EMMAX_ATTN_NW=8 --batch-per-gpu 16" "b1_base" "fp8b8_base --fp8 --batch-per-gpu 8" "fp8b8_roll EMMAX_KM_ROLL=1 --fp8 --batch-per-gpu 8" 2>&1 | tee $O/ab.txt
for r in 1 2 1 2; do EMMAX_RESID32=$r timeout 300 python tools/stage_bench.py --vision-batches 1 --prefill-batches 1,8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resid32=$r prefill', d['prefill'])"; done 2>&1 | tee $O/prefill_ab.txt
