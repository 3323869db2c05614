# round-3 counter passes for the dense kernels: GEMM MFMA-pipe utilisation (one rocprofv3 run per
# shape, so that launches of the same kernel instantiation are attributed to their shape), attention counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r03; mkdir -p $O; : > $O/r03_pmc_gemm_mfma.jsonl
for SH in "8192,8192,8192,0" "65536,3072,1024,0" "65536,1152,4352,0" "6144,22016,4096,2" "6144,12288,4096,0" "768,12288,4096,0" "65536,4096,8704,1"; do
  D=$O/gemm_$(echo $SH | tr ',' 'x')
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --kernel-include-regex emmax_gemm --output-format csv -d $D -o pmc -- python tools/gemm_bench.py "$SH" > $D.log 2>&1
  python tools/pmc_gemm_summary.py $D "M,N,K,act=$SH" | tr -d '\n' >> $O/r03_pmc_gemm_mfma.jsonl; echo >> $O/r03_pmc_gemm_mfma.jsonl
done
cat $O/r03_pmc_gemm_mfma.jsonl
if [ "$1" = "attn" ]; then bash tools/gpucmd_attn_pmc.sh pmc_r03/attn > $O/attn.log 2>&1; tail -5 $O/attn.log; fi
