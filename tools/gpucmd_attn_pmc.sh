cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-attn_pmc}; mkdir -p $O
python tools/attn_probe.py > $O/attn_probe.txt 2>&1; cat $O/attn_probe.txt
ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc1 -o pmc -- python tools/attn_probe.py > $O/pmc1.log 2>&1
ATTN_REPS=2 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc2 -o pmc -- python tools/attn_probe.py > $O/pmc2.log 2>&1
ATTN_REPS=2 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc3 -o pmc -- python tools/attn_probe.py > $O/pmc3.log 2>&1
ATTN_REPS=2 timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex emmax_attention --output-format csv -d $O/pmc4 -o pmc -- python tools/attn_probe.py > $O/pmc4.log 2>&1
python tools/pmc_attn_summary.py $O
