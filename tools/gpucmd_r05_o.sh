# round 5, session 2: bit-regression of this session's build against the session-start build (scratch_prev/ = the tree of commit 5273362 with its
# library, staged by hand, not committed).  With the new forms switched off (EMMAX_ATTN_KSPLIT=0 EMMAX_ATTN_LAZY=0 EMMAX_GEMM_SK_BIG=0) the build must
# reproduce it bit for bit (the shared-M0 tile DMA, the packed descriptors and the non-temporal K / V loads change no arithmetic); then the default
# build against it (what the new forms change), the decode-side tests and the bench at batch 1 / 8 / 32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
(cd scratch_prev && timeout 600 python tools/regress_bits.py prev > ../$O/regress_prev.log 2>&1)
EMMAX_ATTN_KSPLIT=0 EMMAX_ATTN_LAZY=0 EMMAX_GEMM_SK_BIG=0 timeout 600 python tools/regress_bits.py now_off --against prev 2>&1 | grep " vs " | tee $O/regress_bits.txt
timeout 600 python tools/regress_bits.py now --against prev 2>&1 | grep " vs " | tee -a $O/regress_bits.txt
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_full_depth_gpu.py tests/test_serving_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for fl in "" "--batch-per-gpu 8" "--batch-per-gpu 32"; do timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $fl 2>/dev/null | tail -1 >> $O/bench.jsonl; done
python -c "
import json
for l in open('$O/bench.jsonl'):
    d=json.loads(l); print(d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['decode_ms_per_token'], d['roofline']['frac'], d.get('decode_step_hbm_frac'))"
