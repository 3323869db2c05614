cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh bench rocprof variants pmc stage serve 2>&1 | tail -12
O=gpurun_out/prof_round
PROBE_BATCH=8 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex emmax_decode --output-format csv -d $O/pmc_b8_FETCH_SIZE -o pmc -- python tools/pmc_probe.py > $O/pmc_b8_FETCH.log 2>&1
PROBE_BATCH=8 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex emmax_decode --output-format csv -d $O/pmc_b8_WRITE_SIZE -o pmc -- python tools/pmc_probe.py > $O/pmc_b8_WRITE.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_stage -o stage -- python tools/stage_bench.py --vision-batches 256 --prefill-batches 8 > $O/rocprof_stage.log 2>&1
bash tools/gpucmd_attn_pmc.sh prof_round/attn_pmc > $O/attn_pmc.log 2>&1; rm -f $O/attn_pmc/pmc*/*kernel_trace.csv
timeout 600 python tools/gemm_bench.py > $O/gemm_bench.json 2>/dev/null
ls -R $O | head -60
# the kernel trace of the bench is ~50 MB: keep the stats, drop the per-dispatch trace (gpurun_out/ merges back <= 64 MiB)
rm -f $O/rocprof/*kernel_trace.csv $O/rocprof_stage/*kernel_trace.csv
du -sh $O
