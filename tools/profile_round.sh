#!/bin/bash
# THE profile pass of a round (run through gpurun; the one parameterised runner -- per-round one-off command scripts live in the
# git-ignored tools/bin/): bench line with 50 timed calls, rocprofv3 kernel stats of the same command, bench variants (graph,
# B = 8 / 16 / 32 / 64, fp8 weights, fp8 KV cache, exact numerics), decode-kernel HBM traffic (PMC: B = 1, 8, 16, 32), GEMM MFMA-pipe
# counters, stage / serve benches.
#   usage: ROUND=r06 tools/profile_round.sh [part...]     parts: bench rocprof variants stage serve gemm pmc pmcgemm pmcattn (default: all) + pmcexact (the exact-numerics decode step)
# Outputs: gpurun_out/prof_$ROUND/ (the summaries are copied into profiles/${ROUND}_* by hand)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROUND=${ROUND:-r06}
O=gpurun_out/prof_$ROUND; mkdir -p $O
parts=${*:-bench rocprof variants stage serve gemm pmc pmcgemm pmcattn}
for part in $parts; do
  case $part in
    bench)    timeout 900 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err ;;
    rocprof)  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.log 2>&1
              rm -f $O/rocprof/*/*kernel_trace.csv $O/rocprof/*kernel_trace.csv
              timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_exact -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --exact > $O/rocprof_bench_exact.log 2>&1
              rm -f $O/rocprof_exact/*/*kernel_trace.csv $O/rocprof_exact/*kernel_trace.csv
              timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_b32 -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch-per-gpu 32 > $O/rocprof_bench_b32.log 2>&1
              rm -f $O/rocprof_b32/*/*kernel_trace.csv $O/rocprof_b32/*kernel_trace.csv ;;
    variants) : > $O/bench_variants.jsonl
              for v in "--exact" "--exact-fp32kv" "--exact --graph" "--exact --batch-per-gpu 2" "--exact --batch-per-gpu 8" "--graph" "--batch-per-gpu 8" "--batch-per-gpu 16" "--batch-per-gpu 32" "--fp8" "--fp8 --batch-per-gpu 8" "--fp8 --batch-per-gpu 8 --graph" "--fp8 --batch-per-gpu 16" \
                       "--batch-per-gpu 8 --kv-fp8" "--batch-per-gpu 16 --kv-fp8" "--batch-per-gpu 32 --kv-fp8" "--fp8 --batch-per-gpu 8 --kv-fp8" "--fp8 --batch-per-gpu 16 --kv-fp8" \
                       "--batch-per-gpu 64" "--batch-per-gpu 64 --kv-fp8" "--fp8 --batch-per-gpu 32 --kv-fp8" "--fp8 --batch-per-gpu 64 --kv-fp8"; do
                timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v 2>/dev/null | tail -1 >> $O/bench_variants.jsonl
              done
              EMMAX_DIST_SINGLETON=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 2 --warmup 1 --batch-per-gpu 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rccl_singleton_b8.json ;;
    stage)    timeout 600 python tools/stage_bench.py 2>/dev/null | tail -1 > $O/stage_bench.json
              rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_stage -o stage -- python tools/stage_bench.py --vision-batches 256 --prefill-batches 1,8 > $O/rocprof_stage.log 2>&1
              rm -f $O/rocprof_stage/*/*kernel_trace.csv $O/rocprof_stage/*kernel_trace.csv ;;
    serve)    timeout 900 python tools/serve_bench.py 2>/dev/null | tail -1 > $O/serve_bench.json
              timeout 900 python tools/serve_bench.py --requests 96 --slots 16 2>/dev/null | tail -1 > $O/serve_bench_16.json
              timeout 1200 python tools/serve_bench.py --requests 128 --slots 32 2>/dev/null | tail -1 > $O/serve_bench_32.json
              timeout 1200 python tools/serve_bench.py --requests 384 --slots 32 2>/dev/null | tail -1 > $O/serve_bench_32_384req.json
              timeout 1200 python tools/serve_bench.py --requests 768 --slots 64 2>/dev/null | tail -1 > $O/serve_bench_64_768req.json
              EMMAX_KV_FP8=1 timeout 1200 python tools/serve_bench.py --requests 768 --slots 64 2>/dev/null | tail -1 > $O/serve_bench_64_768req_kvfp8.json ;;
    gemm)     timeout 600 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1 ;;
    pmc)      for B in 1 8 16 32; do
                for c in FETCH_SIZE WRITE_SIZE; do
                  PROBE_BATCH=$B timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex emmax_decode --output-format csv -d $O/pmc_b${B}_$c -o pmc -- python tools/pmc_probe.py > $O/pmc_b${B}_$c.log 2>&1
                  rm -f $O/pmc_b${B}_$c/*/*kernel_trace.csv
                done
                python tools/pmc_summarize.py $(find $O/pmc_b${B}_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_b${B}_WRITE_SIZE -name "*counter_collection.csv") $O/pmc_traffic_b$B.json $B 2>&1 | tail -1
              done ;;
    pmcexact) for c in FETCH_SIZE WRITE_SIZE; do
                PROBE_EXACT=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "emmax_(x_)?decode" --output-format csv -d $O/pmc_exact_$c -o pmc -- python tools/pmc_probe.py > $O/pmc_exact_$c.log 2>&1
                rm -f $O/pmc_exact_$c/*/*kernel_trace.csv
              done
              python tools/pmc_summarize.py $(find $O/pmc_exact_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_exact_WRITE_SIZE -name "*counter_collection.csv") $O/pmc_traffic_exact_b1.json 1 2>&1 | tail -1 ;;
    pmcattn)  A=$O/attn_pmc; mkdir -p $A
              python tools/attn_probe.py > $A/attn_probe.txt 2>&1
              ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $A/pmc1 -o pmc -- python tools/attn_probe.py > $A/pmc1.log 2>&1
              ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $A/pmc2 -o pmc -- python tools/attn_probe.py > $A/pmc2.log 2>&1
              ATTN_REPS=2 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $A/pmc3 -o pmc -- python tools/attn_probe.py > $A/pmc3.log 2>&1
              ATTN_REPS=2 timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $A/pmc4 -o pmc -- python tools/attn_probe.py > $A/pmc4.log 2>&1
              python tools/pmc_attn_summary.py $A | tail -3 ;;
    pmcgemm)  : > $O/pmc_gemm_mfma.jsonl
              for SH in "8192,8192,8192,0" "66816,3072,1024,0" "66816,4096,1024,1" "65536,1152,4352,0" "6144,22016,4096,2" "6144,12288,4096,0" "768,12288,4096,0" "65536,4096,8704,1"; do
                D=$O/gemm_$(echo $SH | tr ',' 'x')
                timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --kernel-include-regex emmax_gemm --output-format csv -d $D -o pmc -- python tools/gemm_bench.py "$SH" > $D.log 2>&1
                rm -f $D/*/*kernel_trace.csv
                python tools/pmc_gemm_summary.py $D "M,N,K,act=$SH" | tr -d '\n' >> $O/pmc_gemm_mfma.jsonl; echo >> $O/pmc_gemm_mfma.jsonl
              done ;;
  esac
  echo "$part done rc=$?"
done
du -sh $O
