# round 5, session 2: hipGraph replay against two runtime switches earlier rounds had not tried (tools/ab_bench.sh, one box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "eager " "graph --graph" "graph_pc0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 --graph" "graph_pc1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 --graph" \
  "graph_hdp0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 --graph" "graph_cpwait GPU_STREAMOPS_CP_WAIT=1 --graph" "eager " "graph --graph" \
  "fp8b8 --fp8 --batch-per-gpu 8" "fp8b8_graph --fp8 --batch-per-gpu 8 --graph" "fp8b8_graph_pc0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 --fp8 --batch-per-gpu 8 --graph" 2>&1 | tee gpurun_out/r05t_graph_switches.txt
