#!/bin/bash
# Round 5, eleventh pass: the K-split op tests after the uneven-slice / fp8 extension of decode_kmp.hip, kernel stats of the B = 32 fp8 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_k; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fp8_weight_projection or gemm_small_km" 2>&1 | tail -4 | tee $O/ops.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b32fp8 -- python bench.py --batch-per-gpu 32 --fp8 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220 | tee $O/b32fp8_kernel_stats_head.csv
