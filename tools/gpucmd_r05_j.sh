#!/bin/bash
# Round 5, tenth pass: fp8 weights on decode_kmp.hip (17-32 rows): op + model tests, the bench at B = 24 / 32 with fp8 weights (and the fp8 KV cache),
# per-kernel durations of the B = 32 fp8 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_j; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fp8_weight_projection or gemm_small_km" 2>&1 | tail -4 | tee $O/ops.txt
timeout 1500 python -m pytest tests/test_operating_point_gpu.py -m gpu -x -q -k "nine_to_thirty_two" 2>&1 | tail -4 | tee $O/oppoint.txt
for cfg in "32 --fp8" "24 --fp8" "32 --fp8 --kv-fp8" "17 --fp8"; do
  set -- $cfg; B=$1; shift
  timeout 600 python bench.py --batch-per-gpu $B --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1 > "$O/bench_b${B}$(echo $@ | tr -d ' ').json"
  python - "$O/bench_b${B}$(echo $@ | tr -d ' ').json" "$cfg" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline", {})
print(sys.argv[2], "actions/s", d["value"], "ms/step", d["ms_per_step"], "gate/up us", r.get("kernel_us"), "frac", r.get("frac"))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b32fp8 -- python bench.py --batch-per-gpu 32 --fp8 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220 | tee $O/b32fp8_kernel_stats_head.csv
