# HBM traffic of the batch-8 decode stages (bf16 and fp8 weights) from the PMC counters: two passes each (FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_b8; mkdir -p $O
for mode in bf16 fp8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $mode = fp8 ]; then export PROBE_FP8=1; else unset PROBE_FP8; fi
    PROBE_BATCH=8 timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex emmax_decode --output-format csv -d $O/${mode}_$c -o pmc -- python tools/pmc_probe.py > $O/${mode}_$c.log 2>&1
  done
  python tools/pmc_summarize.py $(find $O/${mode}_FETCH_SIZE -name "*counter_collection.csv") $(find $O/${mode}_WRITE_SIZE -name "*counter_collection.csv") $O/r03_pmc_traffic_b8_$mode.json 8 2>&1 | tail -2
  cat $O/r03_pmc_traffic_b8_$mode.json | python -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d['stages'].items(): print('$mode',k,v['hbm_bytes_per_launch'])"
done
rm -f $(find $O -name "*kernel_trace.csv")
