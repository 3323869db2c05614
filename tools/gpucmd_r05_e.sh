#!/bin/bash
# Round 5, fifth pass: decode batches 17-32 (decode_kmp.hip): op tests, the 32-row model test, bench at B = 16 / 24 / 32; prefill A/B after the OUT_F32 stagger change
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_e; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_operating_point_gpu.py -m gpu -x -q -k "small_km or thirty_two or f32 or big_tile" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
bash tools/ab_bench.sh "b16 --batch-per-gpu 16" "b24 --batch-per-gpu 24" "b32 --batch-per-gpu 32" "b32g --batch-per-gpu 32 --graph" "b17 --batch-per-gpu 17" 2>&1 | tee $O/ab.txt
for r in 1 2 1 2; do EMMAX_RESID32=$r timeout 300 python tools/stage_bench.py --vision-batches 8 --prefill-batches 1,8 2>$O/stage.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resid32=$r prefill', d['prefill'])"; done 2>&1 | tee $O/prefill_ab.txt
