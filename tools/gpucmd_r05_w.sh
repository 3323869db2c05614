# round 5, session 2: runtime switches earlier rounds had not tried, on the EAGER decode step at batch 1 and fp8 batch 1 (tools/ab_bench.sh, one box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "eager " "qdev HSA_ALLOCATE_QUEUE_DEV_MEM=1" "noint HSA_ENABLE_INTERRUPT=0" "kacopy0 DEBUG_HIP_KERNARG_COPY_OPT=0" "kacopy1 DEBUG_HIP_KERNARG_COPY_OPT=1" \
  "hwq1 GPU_MAX_HW_QUEUES=1" "hwq8 GPU_MAX_HW_QUEUES=8" "dynq DEBUG_HIP_DYNAMIC_QUEUES=1" "batch1 DEBUG_CLR_MAX_BATCH_SIZE=1" "scratchalt HSA_ENABLE_SCRATCH_ALT=1" "mwaitx HSA_ENABLE_MWAITX=1" "eager " \
  "fp8 --fp8" "fp8_qdev HSA_ALLOCATE_QUEUE_DEV_MEM=1 --fp8" 2>&1 | cut -c1-60 | tee gpurun_out/r05w_runtime_switches.txt
