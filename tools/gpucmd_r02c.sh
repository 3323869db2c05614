cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests -m gpu -q -s -k "large_batch_plans or simpler or divergence or verify_checkpoint" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 tools/bin/l2_retention > $O/l2_retention.txt 2>&1; cat $O/l2_retention.txt
timeout 900 python bench.py --steps 4 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
python tools/attn_probe.py > $O/attn_probe.txt 2>&1; cat $O/attn_probe.txt
ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc_attn1 -o pmc -- python tools/attn_probe.py > $O/pmc_attn1.log 2>&1
ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc_attn2 -o pmc -- python tools/attn_probe.py > $O/pmc_attn2.log 2>&1
find $O -name "*counter_collection.csv" | head; tail -3 $O/pmc_attn1.log
