# round 5, session 2: the decode-attention switches re-measured behind the non-temporal K / V loads (tools/ab_bench.sh, one box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "b1 " "b1_ns4 EMMAX_ATTN_NSPLIT=4" "b1_ns16 EMMAX_ATTN_NSPLIT=16" "b1_deep EMMAX_ATTN_DEEP=1" \
  "b8 --batch-per-gpu 8" "b8_deep EMMAX_ATTN_DEEP=1 --batch-per-gpu 8" "b8_nw8 EMMAX_ATTN_NW=8 --batch-per-gpu 8" "b8_ns2 EMMAX_ATTN_NSPLIT=2 EMMAX_KM=1 --batch-per-gpu 8" \
  "b16 --batch-per-gpu 16" "b16_deep EMMAX_ATTN_DEEP=1 --batch-per-gpu 16" "b32 --batch-per-gpu 32" "b32_nodeep EMMAX_ATTN_DEEP=0 --batch-per-gpu 32" \
  "b1 " "b8 --batch-per-gpu 8" 2>&1 | tee gpurun_out/r05q_attn_switches.txt
