#!/bin/bash
# Regenerates everything under profiles/ that depends on the decode kernels (run on the GPU box through gpurun; outputs land
# in gpurun_out/prof_round/, copy the summaries to profiles/ afterwards).  usage: tools/profile_round.sh [part...]
#   parts: bench rocprof variants pmc stage serve   (default: all)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out/prof_round
mkdir -p $O
parts=${*:-bench rocprof variants pmc stage serve}
for part in $parts; do
  case $part in
    bench)    timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ;;
    rocprof)  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.log 2>&1 ;;
    variants) : > $O/bench_variants.jsonl
              for v in "--graph" "--batch-per-gpu 8" "--fp8" "--fp8 --batch-per-gpu 8" "--fp8 --batch-per-gpu 8 --graph"; do
                timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v 2>/dev/null | tail -1 >> $O/bench_variants.jsonl
              done ;;
    pmc)      for c in FETCH_SIZE WRITE_SIZE; do
                timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex emmax_decode --output-format csv -d $O/pmc_$c -o pmc -- python tools/pmc_probe.py > $O/pmc_$c.log 2>&1
              done ;;
    stage)    timeout 600 python tools/stage_bench.py 2>/dev/null | tail -1 > $O/stage_bench.json ;;
    serve)    timeout 900 python tools/serve_bench.py 2>/dev/null | tail -1 > $O/serve_bench.json ;;
  esac
  echo "$part done rc=$?"
done
find $O -name "*.csv" | head -20
