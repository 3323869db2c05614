#!/bin/bash
# Round 5, fourth pass: the 128 x 256 x 32 GEMM tile (two blocks per CU): parity, race screen, A/B per shape against the planned geometry and the forced big tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_d; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/gemm_race_screen.py 10 --k32 > $O/race_k32.txt 2>&1; tail -3 $O/race_k32.txt
timeout 900 python tools/gemm_bench.py --ab gemm_big=-1,1,2 > $O/gemm_ab.txt 2>&1; grep -v "^{" $O/gemm_ab.txt
timeout 600 python -m pytest tests/test_operating_point_gpu.py -m gpu -x -q -k "reduce_pass or every_logit or slot_served" -s 2>&1 | grep -v "^$" | tail -6 | cut -c1-1500 | tee $O/pytest2.log
for r in 1 2 1 2; do EMMAX_RESID32=$r timeout 300 python tools/stage_bench.py --vision-batches 8 --prefill-batches 1,8 2>$O/stage.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resid32=$r prefill', d['prefill'])"; done 2>&1 | tee $O/prefill_ab.txt
