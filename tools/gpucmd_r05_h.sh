#!/bin/bash
# Round 5, eighth pass: two lab variants -- the k32 GEMM tile with its DMA requests spread between the MFMA groups (gemm_dbg 32), and
# four chunks of keys in flight in the bf16 decode attention (attn_deep)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h; mkdir -p $O
EMMAX_GEMM_DBG=32 timeout 600 python tools/gemm_race_screen.py 6 --k32 > $O/race_k32_spread.txt 2>&1; tail -2 $O/race_k32_spread.txt
EMMAX_GEMM_BIG=2 timeout 900 python tools/gemm_bench.py --ab gemm_dbg=0,32 > $O/gemm_k32_spread_ab.txt 2>&1; grep -v "^{" $O/gemm_k32_spread_ab.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "decode_attention" 2>&1 | tail -2
EMMAX_ATTN_DEEP=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "decode_attention" 2>&1 | tail -2
bash tools/ab_bench.sh "b8 --batch-per-gpu 8" "b8_deep EMMAX_ATTN_DEEP=1 --batch-per-gpu 8" "b16 --batch-per-gpu 16" "b16_deep EMMAX_ATTN_DEEP=1 --batch-per-gpu 16" "b32 --batch-per-gpu 32" "b32_deep EMMAX_ATTN_DEEP=1 --batch-per-gpu 32" "b1" "b1_deep EMMAX_ATTN_DEEP=1" 2>&1 | tee $O/ab.txt
