cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh bench rocprof variants pmc 2>&1 | tail -8
O=gpurun_out/prof_round
python tools/pmc_summarize.py $O/pmc_FETCH_SIZE/*/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/*/pmc_counter_collection.csv $O/pmc_traffic.json 2>&1 | tail -2 || ls -R $O/pmc_FETCH_SIZE | head
bash tools/chain_trace.sh 1 ks_trace > $O/ks_trace.txt 2>&1
EMMAX_PCHAIN=1 bash tools/chain_trace.sh 1 chain_trace > $O/chain_trace.txt 2>&1   # the chain is opt-in: without the switch stage 6 is refused
EMMAX_PCHAIN=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pchain.json
EMMAX_ATTN_MERGE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_attn_merge.json
EMMAX_KS=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ks0.json
# the kernel trace of the bench is ~50 MB: keep the stats, drop the per-dispatch trace (gpurun_out/ merges back <= 64 MiB)
rm -f $O/rocprof/*/*kernel_trace.csv $O/rocprof/*kernel_trace.csv $O/pmc_*/*/*kernel_trace.csv
find $O -type f | head -40; du -sh $O
