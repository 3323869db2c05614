# round 5, session 2: gaps between the kernels of a decode step (rocprofv3 --kernel-trace of short bench.py runs, tools/kernel_gaps.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
for v in "b1 " "b8 --batch-per-gpu 8" "b32 --batch-per-gpu 32" "b1graph --graph" "b8fp8 --fp8 --batch-per-gpu 8"; do
  set -- $v; lab=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$lab -o t -- python bench.py --steps 1 --warmup 0 --new-tokens 96 --no-cpu-baseline "$@" > $O/tr_$lab.log 2>&1
  f=$(find $O/tr_$lab -name "*kernel_trace.csv" | head -1)
  python tools/kernel_gaps.py $f $lab 2>&1 | tee -a $O/kernel_gaps.txt
  rm -rf $O/tr_$lab
done
