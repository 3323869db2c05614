# refresh of the GEMM-dependent profiles after a GEMM change (stage timings, per-shape throughput, in-situ kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; mkdir -p $O
timeout 600 python tools/stage_bench.py 2>/dev/null | tail -1 > $O/stage_bench.json
timeout 600 python tools/gemm_bench.py > $O/gemm_bench.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_stage -o stage -- python tools/stage_bench.py --vision-batches 256 --prefill-batches 8 > $O/rocprof_stage.log 2>&1
rm -f $O/rocprof_stage/*kernel_trace.csv
cat $O/stage_bench.json; head -20 $O/gemm_bench.json
